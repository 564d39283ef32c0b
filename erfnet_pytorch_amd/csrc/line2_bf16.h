// The fused pair of line_bf16_kernel with TWO waves per SIMD, specialised by convolution.
//
// line_bf16_kernel keeps the weights of both convolutions in every wave's registers, which leaves
// room for one wave per SIMD: the DMA waits, the LDS fragment reads and both epilogues of a tile run
// with nothing to overlap them (measured: 11.8 k cycles per 64-row tile of 128 channels against 3.1 k
// of MFMA issue).  Here a block has eight waves:
//   waves 0-3 ("A") hold w1 and run conv_a of tile k+1 (DMA'd operand tiles, first epilogue into the
//                   intermediate tile MID[(k+1) & 1], optional store / mask pass of the intermediate),
//   waves 4-7 ("B") hold w2 and run conv_b of tile k out of MID[k & 1], the final epilogue and the
//                   output pass,
// so each SIMD hosts one A wave and one B wave whose MFMAs fill each other's waits.  The arithmetic of
// a tile is line_bf16_kernel's, instruction for instruction: results are bit-identical (tested).
//
// Synchronisation: s_barrier joins all eight waves, so both roles execute the SAME barriers: two per
// period, whatever the tile's geometry, and the same number of periods (A: its K tiles + one drain
// period; B: one idle period + its K tiles).  In period p role A works on tile p, role B on tile p - 1:
//            A (tile p)                                       B (tile p - 1)
//   wait     operand tiles of all three taps (+ residual)    residual tile of tile p - 1 landed,
//            landed (requested behind beta of period p - 1)   output stores of tile p - 2 done
//   alpha    ------------------------------------------------------------------------------------
//            products of the three taps out of TAP[0..2],    request the residual tile of tile p;
//            first epilogue into MID[p & 1]                   products out of MID[(p - 1) & 1], final
//                                                             epilogue (+ residual) into RES[(p - 1) & 1]
//   beta     ------------------------------------------------------------------------------------
//            request the operand tiles of tile p + 1;         output pass over RES[(p - 1) & 1]
//            store / mask pass over MID[p & 1]
// One buffer per tap (all three operand tiles of a tile are in flight together), the intermediate and
// B's residual / result tile double-buffered: a buffer is rewritten one barrier after its last reader
// passed.  (Five barriers per period — one per tap buffer hand-over — left each role waiting for the
// other in every interval: 7.4 k ticks per period for 2.7 k of products.)
#pragma once
#include "line_bf16.h"

template <int NC, int BM>
__global__ __launch_bounds__(512, 1) void line2_bf16_kernel(const LineDesc d)
{
    constexpr int GC = NC / 32, GP = 4 / GC, WM = BM / (32 * GP);
    static_assert(WM >= 1 && WM * GP * 32 == BM, "tile shape");
    constexpr int KS = NC / 16;
    constexpr int CPR = NC / 8;
    constexpr int PITCH = NC * 2;
    constexpr int T = BM * PITCH;
    constexpr int PASSES = BM * CPR / 256;
    static_assert(BM * CPR % 256 == 0, "whole copy passes");
    constexpr int MIDSZ = T + PITCH;                        // tile + zero row
    constexpr int OFF_TAP = 0, OFF_MID = 3 * T, OFF_RES = OFF_MID + 2 * MIDSZ, OFF_RESA = OFF_RES + 2 * T;
    constexpr int OFF_CST = OFF_RESA + T;
    static_assert(OFF_CST + 4 * NC * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[OFF_CST + 4 * NC * 4];
    __shared__ unsigned a_done;     // role A's waves that have finished reading the tap buffers (running count)
    float *cst = reinterpret_cast<float *>(lds + OFF_CST);        // sc1[NC] sh1[NC] sc2[NC] sh2[NC]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_a = wave8 < 4;                          // wave-uniform
    const int wave = wave8 & 3, ltid = tid & 255;           // index within the role
    const int cg = wave % GC, wp = wave / GC;
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;

    const int L = d.L;
    const int ntiles = d.nlines;                            // a pair covers whole lines (segs = 1)
    const int nblk = gridDim.x;
    const int b0 = (int)xcd_tile(blockIdx.x, nblk);
    const int s1a = d.s1[0], s1b = d.s1[1], s1c = d.s1[2];
    const bool res_first = d.epi1.res != nullptr;
    const bool res_last = d.epi2.res != nullptr;
    const int K = b0 < ntiles ? (ntiles - b0 + nblk - 1) / nblk : 0;      // tiles of this block

    struct Geo { int q; unsigned base; int len; };
    auto geo = [&](int tile) __attribute__((always_inline)) {
        Geo g;
        const int n = tile / d.lines_per_img;
        g.q = tile - n * d.lines_per_img;
        g.len = L < BM ? L : BM;
        g.base = (unsigned)n * d.img_pixels + (unsigned)g.q * d.qstride;
        return g;
    };
    auto tap_shift = [&](int t) __attribute__((always_inline)) { return t == 0 ? s1a : (t == 1 ? s1b : s1c); };
    auto tap_valid = [&](const Geo &g, int t) __attribute__((always_inline)) {
        return (unsigned)(g.q + tap_shift(t)) < (unsigned)d.lines_per_img;
    };
    // LDS-DMA of one line tile by the 256 threads of a role
    auto issue_tile = [&](const bf16_t *src, unsigned pixel0, int len, int dst_off) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int slot = p * 256 + ltid;
            const int r = slot / CPR, pc = slot % CPR;
            const int c = pc ^ line_swz<NC>(r);
            const int rr = r < len ? r : len - 1;
            const bf16_t *gp = src + ((size_t)(pixel0 + (unsigned)(rr * d.pstride)) * NC + c * 8);
            lds_dma16(gp, lds0 + dst_off + (p * 256 + wave * 64) * 16);
        }
    };

    // common set-up: epilogue constants, zero rows of both MID buffers
    for (int c = tid; c < NC; c += 512) {
        float sc, sh;
        epi_channel_affine(d.epi1, c, sc, sh);
        cst[c] = sc; cst[NC + c] = sh;
        epi_channel_affine(d.epi2, c, sc, sh);
        cst[2 * NC + c] = sc; cst[3 * NC + c] = sh;
    }
    if (tid == 0) a_done = 0u;
    for (int i = tid; i < 2 * (PITCH / 4); i += 512) {
        const int buf = i / (PITCH / 4), e = i % (PITCH / 4);
        reinterpret_cast<unsigned *>(lds + OFF_MID + buf * MIDSZ + T)[e] = 0u;
    }

    f32x16 acc[WM];
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    };
    auto mma_tap = [&](const bf16x8 (&w)[KS], const unsigned char *const (&rowp)[WM], const unsigned (&x4)[WM]) __attribute__((always_inline)) {
        constexpr int D = WM >= 4 ? 2 : 3;
        bf16x8 xf[D][WM];
#pragma unroll
        for (int s = 0; s < D - 1; ++s)
#pragma unroll
            for (int a = 0; a < WM; ++a)
                xf[s][a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + ((unsigned)(s << 5) ^ x4[a]));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + D - 1 < KS) {
#pragma unroll
                for (int a = 0; a < WM; ++a)
                    xf[(s + D - 1) % D][a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + ((unsigned)((s + D - 1) << 5) ^ x4[a]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], xf[s % D][a], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // coalesced pass of a role over a finished LDS tile: [mask by a global tensor or by its bit
    // tensor, kept in the tile] -> [global store] -> [the stored values' "> 0" bits].  A thread holds 8
    // channels = one byte of a pixel's mask words; the four lanes of a quad hold the four bytes of a word.
    // The mask words of a tile are requested early (fetch_bits: a few registers, a load latency the
    // pass would otherwise wait out) and handed in.
    auto fetch_bits = [&](const unsigned *mkb, unsigned (&mw)[PASSES], const Geo &g) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int slot = p * 256 + ltid;
            const int r = slot / CPR, pc = slot % CPR;
            const int c = pc ^ line_swz<NC>(r);
            const int rr = r < g.len ? r : g.len - 1;
            const unsigned pix = g.base + (unsigned)(rr * d.pstride);
            mw[p] = mkb[pix * (unsigned)(NC / 32) + (unsigned)(c >> 2)];
        }
    };
    auto sweep_tile = [&](int lds_off, const bf16_t *mk, const unsigned (&mw)[PASSES], bool bits_in, bf16_t *dst,
                          unsigned *ob, bool keep, const Geo &g) __attribute__((always_inline)) {
        constexpr int B = PASSES % 4 == 0 ? 4 : (PASSES % 2 == 0 ? 2 : 1);
#pragma unroll
        for (int p0 = 0; p0 < PASSES; p0 += B) {
            u32x4 v[B], m[B];
            size_t off[B];
            unsigned wix[B], csh[B];
            bool live[B];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const int slot = (p0 + i) * 256 + ltid;
                const int r = slot / CPR, pc = slot % CPR;
                const int c = pc ^ line_swz<NC>(r);
                live[i] = r < g.len;
                const int rr = live[i] ? r : g.len - 1;
                const unsigned pix = g.base + (unsigned)(rr * d.pstride);
                off[i] = (size_t)pix * NC + c * 8;
                wix[i] = pix * (unsigned)(NC / 32) + (unsigned)(c >> 2);
                csh[i] = (unsigned)(c & 3) * 8u;
                v[i] = *reinterpret_cast<const u32x4 *>(lds + lds_off + slot * 16);
            }
            if (mk) {
#pragma unroll
                for (int i = 0; i < B; ++i) m[i] = *reinterpret_cast<const u32x4 *>(mk + off[i]);
#pragma unroll
                for (int i = 0; i < B; ++i) v[i] = mask_bf16x8(v[i], m[i]);
            } else if (bits_in) {
#pragma unroll
                for (int i = 0; i < B; ++i) v[i] = mask_bits_bf16x8(v[i], mw[p0 + i] >> csh[i]);
            }
            if ((mk || bits_in) && keep) {
#pragma unroll
                for (int i = 0; i < B; ++i) *reinterpret_cast<u32x4 *>(lds + lds_off + ((p0 + i) * 256 + ltid) * 16) = v[i];
            }
            if (dst) {
#pragma unroll
                for (int i = 0; i < B; ++i)
                    if (live[i]) *reinterpret_cast<u32x4 *>(dst + off[i]) = v[i];
            }
            if (ob) {
#pragma unroll
                for (int i = 0; i < B; ++i) {
                    int w = (int)(sign_bits_bf16x8(v[i]) << csh[i]);
                    w |= __builtin_amdgcn_mov_dpp(w, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
                    w |= __builtin_amdgcn_mov_dpp(w, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
                    if (live[i] && (ltid & 3) == 0) ob[wix[i]] = (unsigned)w;
                }
            }
        }
    };
    auto store_acc_t = [&](auto relu_t, auto scale_t, auto res_t, const float *sc_, const float *sh_, bool zero_dead,
                           int res_off, int dst_off, const Geo &g) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_t)::value, SCALE = decltype(scale_t)::value, RES = decltype(res_t)::value;
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const bool ragged = zero_dead && g.len < BM;
        f32x4 sc[4], sh[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = 32 * cg + 8 * gq + 4 * h;
            if (SCALE) sc[gq] = *reinterpret_cast<const f32x4 *>(sc_ + ch);
            sh[gq] = *reinterpret_cast<const f32x4 *>(sh_ + ch);
        }
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            const int px = (wp * WM + a) * 32 + l31;
            const int sw = line_swz<NC>(px);
            u32x2 rr[4];
            if (RES) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    rr[gq] = *reinterpret_cast<const u32x2 *>(lds + res_off + px * PITCH + (((4 * cg + gq) ^ sw) << 4) + 8 * h);
            }
            u32x2 pk[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v = f32x4{acc[a][4 * gq], acc[a][4 * gq + 1], acc[a][4 * gq + 2], acc[a][4 * gq + 3]};
                if (SCALE) v = v * sc[gq] + sh[gq];
                else v = v + sh[gq];
                if (RES) v = v + unpack_bf16x4(rr[gq]);
                pk[gq] = pack_bf16x4(v);
                if (RELU) pk[gq] = __builtin_bit_cast(u32x2, __builtin_elementwise_max(__builtin_bit_cast(s16x4, pk[gq]), s16x4{0, 0, 0, 0}));
            }
            if (ragged) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) pk[gq] = px < g.len ? pk[gq] : u32x2{0u, 0u};
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<u32x2 *>(lds + dst_off + px * PITCH + (((4 * cg + gq) ^ sw) << 4) + 8 * h) = pk[gq];
        }
    };
    auto store_acc = [&](const Epilogue &e, const float *sc_, const float *sh_, bool add_res, bool zero_dead,
                         int res_off, int dst_off, const Geo &g) __attribute__((always_inline)) {
        using T1 = std::true_type;
        using T0 = std::false_type;
        const int key = (e.relu ? 4 : 0) | (e.bn.on ? 2 : 0) | (add_res ? 1 : 0);
        switch (key) {          // wave-uniform
        case 0: store_acc_t(T0{}, T0{}, T0{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        case 1: store_acc_t(T0{}, T0{}, T1{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        case 2: store_acc_t(T0{}, T1{}, T0{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        case 3: store_acc_t(T0{}, T1{}, T1{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        case 4: store_acc_t(T1{}, T0{}, T0{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        case 5: store_acc_t(T1{}, T0{}, T1{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        case 6: store_acc_t(T1{}, T1{}, T0{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        default: store_acc_t(T1{}, T1{}, T1{}, sc_, sh_, zero_dead, res_off, dst_off, g); break;
        }
    };

    if (K == 0) return;             // (block-uniform: no barrier has been executed yet)
#ifdef ERF_LINE_STAMPS
    int stamp_i = 0;                // diagnostic build: [block][role][64] stamps of each role's first wave
#define L2_STAMP()                                                                                 \
    do {                                                                                           \
        if (d.stamps && ltid == 0 && stamp_i < 64)                                                 \
            d.stamps[((size_t)blockIdx.x * 2 + (role_a ? 0 : 1)) * 64 + stamp_i++] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define L2_STAMP() do {} while (0)
#endif
    L2_STAMP();
    clear_acc();
    if (role_a) {
        // ================================================================= role A: conv_a, tile by tile
        bf16x8 w1[3][KS];
        const bf16_t *mk1 = (const bf16_t *)d.epi1.mask;
        const unsigned *mkb1 = mk1 ? nullptr : d.epi1.mask_bits;
        unsigned mw[PASSES] = {};
        Geo g = geo(b0);
        auto request_taps = [&](const Geo &q) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (tap_valid(q, t)) issue_tile(d.in, q.base + (unsigned)(tap_shift(t) * d.qstride), q.len, OFF_TAP + t * T);
        };
        auto request_res = [&](const Geo &q) __attribute__((always_inline)) {
            if (res_first) issue_tile((const bf16_t *)d.epi1.res, q.base, q.len, OFF_RESA);
        };
        request_taps(g);
        request_res(g);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                w1[t][s] = *reinterpret_cast<const bf16x8 *>(d.wp1 + ((size_t)((t * KS + s) * GC + cg) * 64 + lane) * 8);
        __syncthreads();            // set-up (constants, zero rows) visible; B executes the same barrier
        L2_STAMP();                 // A: set-up done
        for (int k = 0; k < K; ++k) {
            const bool more = k + 1 < K;
            Geo gn = g;
            if (more) gn = geo(b0 + (k + 1) * nblk);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of the operand tiles
            L2_STAMP();             // A: operand tiles landed
            lds_barrier();                                          // alpha
            L2_STAMP();             // A: alpha passed
            if (mkb1) fetch_bits(mkb1, mw, g);                      // (land during the products)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (!tap_valid(g, t)) continue;                     // wave-uniform
                const unsigned char *tb = lds + OFF_TAP + t * T;
                const unsigned char *rowp[WM];
                unsigned x4[WM];
#pragma unroll
                for (int a = 0; a < WM; ++a) {
                    const int px = (wp * WM + a) * 32 + l31;
                    rowp[a] = tb + px * PITCH;
                    x4[a] = (unsigned)((line_swz<NC>(px) ^ h) << 4);
                }
                mma_tap(w1[t], rowp, x4);
            }
            L2_STAMP();             // A: products issued
            // The tap buffers are free once the four A waves have their last fragments in registers: a
            // hand-shake among them (a running count in LDS; role B is not involved) lets the next tile's
            // operand tiles be requested a first epilogue earlier than the next barrier would.  The
            // residual tile of the first epilogue is read there and follows behind beta.  (A mask read
            // as a tensor inside the pass is waited for in issue order, i.e. behind these requests: by
            // then they have had the first epilogue's time to land.)
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&a_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned want = 4u * (unsigned)(k + 1);
                while (__hip_atomic_load(&a_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
                request_taps(gn);
            }
            L2_STAMP();             // A: next tiles requested
            const int mid_off = OFF_MID + (k & 1) * MIDSZ;
            store_acc(d.epi1, cst, cst + NC, res_first, true, OFF_RESA, mid_off, g);
            clear_acc();
            L2_STAMP();             // A: intermediate in LDS
            lds_barrier();                                          // beta: tap buffers and residual tile are free
            L2_STAMP();             // A: beta passed
            if (more) request_res(gn);
            if (d.mid || mk1 || mkb1)
                sweep_tile(mid_off, mk1, mw, mkb1 != nullptr, d.mid, d.mid ? d.epi1.bits_out : nullptr, true, g);
            L2_STAMP();             // A: pass over the intermediate done
            g = gn;
        }
        // drain period: B works on the last tile
        lds_barrier();
        lds_barrier();
    } else {
        // ================================================================= role B: conv_b + output, one tile behind
        bf16x8 w2[3][KS];
        const bf16_t *mk2 = (const bf16_t *)d.epi2.mask;
        const unsigned *mkb2 = mk2 ? nullptr : d.epi2.mask_bits;
        unsigned mw[PASSES] = {};
        __syncthreads();            // (pairs with A's set-up barrier)
        // this role idles for a period: its weight fragments are requested behind the set-up barrier, so
        // that the barrier's memory wait covers only role A's half of the CU's weight traffic
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                w2[t][s] = *reinterpret_cast<const bf16x8 *>(d.wp2 + ((size_t)((t * KS + s) * GC + cg) * 64 + lane) * 8);
        Geo g = geo(b0);
        if (res_last) issue_tile((const bf16_t *)d.epi2.res, g.base, g.len, OFF_RES);
        // idle period: A works on the first tile
        lds_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // weights, first residual tile (this wave's share)
        lds_barrier();
        for (int k = 0; k < K; ++k) {
            const bool more = k + 1 < K;
            Geo gn = g;
            if (more) gn = geo(b0 + (k + 1) * nblk);
            const int mid_off = OFF_MID + (k & 1) * MIDSZ, res_off = OFF_RES + (k & 1) * T;
            L2_STAMP();             // B: period starts
            lds_barrier();                                          // alpha
            L2_STAMP();             // B: alpha passed
            // the other result buffer was read for the last time before alpha: it takes the next residual tile
            if (more && res_last) issue_tile((const bf16_t *)d.epi2.res, gn.base, gn.len, OFF_RES + ((k + 1) & 1) * T);
            if (mkb2) fetch_bits(mkb2, mw, g);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int sft = d.s2[t];
                const unsigned char *rowp[WM];
                unsigned x4[WM];
#pragma unroll
                for (int a = 0; a < WM; ++a) {
                    const int rr = (wp * WM + a) * 32 + l31 + sft;
                    const int rz = (unsigned)rr < (unsigned)g.len ? rr : BM;        // beyond the line: the zero row
                    rowp[a] = lds + mid_off + rz * PITCH;
                    x4[a] = (unsigned)((line_swz<NC>(rz) ^ h) << 4);
                }
                mma_tap(w2[t], rowp, x4);
            }
            L2_STAMP();             // B: products issued
            store_acc(d.epi2, cst + 2 * NC, cst + 3 * NC, res_last, false, res_off, res_off, g);
            clear_acc();
            L2_STAMP();             // B: result tile in LDS
            lds_barrier();                                          // beta
            L2_STAMP();             // B: beta passed
            // the next residual tile (requested behind alpha) has had the products' time to land; waiting
            // here, in front of the pass's stores, keeps those stores out of every wait
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sweep_tile(res_off, mk2, mw, mkb2 != nullptr, d.out, d.epi2.bits_out, false, g);
            L2_STAMP();             // B: output pass done
            g = gn;
        }
    }
}
