// 1-D convolution of a non_bottleneck_1d on fp32 activations, computed on the bf16 matrix pipe
// (gfx950, v_mfma_f32_32x32x16_bf16) with every fp32 operand split into three bf16 terms:
//
//   x = xh + xm + xl,   xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)      (exact: 3 x 8 bits)
//   x * w  ~=  xh wh + xh wm + xm wh + xm wm + xh wl + xl wh
//
// Each bf16 product is exact in the fp32 accumulator; the three dropped terms (xm wl, xl wm, xl wl)
// are below 2^-24 of |x w|, i.e. below the rounding of one fp32 multiply-add.  Six bf16 MFMAs per
// k-step against one fp32 MFMA sixteen times slower: 2.7x the fp32 matrix rate at fp32 accuracy.
// Non-finite inputs give NaN (inf - inf in the split); values within 2^-8 of FLT_MAX overflow.
//
// Structure (one wave per SIMD, 256 threads, persistent blocks as line_bf16_kernel):
//   * weights STATIONARY IN REGISTERS as MFMA fragments, three planes (hi / mid / lo), packed by
//     line_pack_kernel: 3 planes x 3 taps x NC/16 k-steps x 4 registers = 288 at NC = 128;
//   * the operand tile of a tap is read from global memory as fp32 one tap AHEAD into registers, split
//     and written to LDS as three bf16 planes [pixel][NC] (swizzled as in line_bf16_kernel) while the
//     MFMAs of the current tap run; the split arithmetic, the LDS fragment reads of the next k-step
//     and the global loads of the tap after next are cut into pieces placed one behind each MFMA;
//   * fp32 epilogue through an fp32 LDS tile (residual by LDS-DMA, coalesced mask / store pass).
#pragma once
#include "line_bf16.h"

// x - (bf16 pair as fp32)
__device__ __forceinline__ f32x2 x3_sub_bf16x2(f32x2 x, unsigned p)
{
    const f32x2 b = f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
    return x - b;           // one packed subtraction
}
__device__ __forceinline__ unsigned x3_cvt2(f32x2 v)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));       // v_cvt_pk_bf16_f32 (RNE)
}

// STATS: the variant that also leaves the column sums of its output (LineDesc::stats).
template <int NC, int BM, bool STATS = false>
__global__ __launch_bounds__(256, (NC == 64 ? 2 : 1)) void line_x3_kernel(const LineDesc d)
{
    constexpr int GC = NC / 32, GP = 4 / GC, WM = BM / (32 * GP);
    static_assert((WM == 1 || WM == 2) && WM * GP * 32 == BM, "tile shape: one or two 32-pixel MFMA tiles per wave");
    constexpr int KS = NC / 16;                 // MFMA k-steps per tap
    constexpr int PITCH = NC * 2;               // bytes per pixel row of a bf16 plane
    constexpr int TP = BM * PITCH, T3 = 3 * TP; // one plane, one operand tile (three planes)
    constexpr int F4R = NC / 4;                 // 16-byte chunks per fp32 pixel row
    constexpr int FPITCH = NC * 4, TF = BM * FPITCH;
    constexpr int RPP = 256 / F4R;              // rows per copy pass of the block
    constexpr int PASSES = BM / RPP;            // float4 per thread and tile
    // one staged float4 per k-step (<128,64>, <64,64>) or per second k-step (<128,32>)
    static_assert(KS % PASSES == 0 && (WM == 1 || PASSES == KS), "conversion schedule");
    constexpr int CSTRIDE = KS / PASSES;
    constexpr int OFF_TAP = 0, OFF_RES = 2 * T3, OFF_CST = OFF_RES + TF;
    // optional output statistics: per-thread partial sums of a tile (8 floats), running sums (doubles)
    constexpr int OFF_SPT = OFF_CST + 2 * NC * 4, OFF_RUN = OFF_SPT + (STATS ? 256 * 32 : 0);
    constexpr int LDS_BYTES = OFF_RUN + (STATS ? 2 * NC * 8 : 0);
    static_assert(LDS_BYTES * (NC == 64 ? 2 : 1) <= 160 * 1024, "LDS budget");
    constexpr size_t PL = (size_t)3 * NC * NC;  // elements per weight plane
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    float *cst = reinterpret_cast<float *>(lds + OFF_CST);          // sc[NC] sh[NC]
    double *run = reinterpret_cast<double *>(lds + OFF_RUN);        // sum[NC] sumsq[NC]
    constexpr bool want_stats = STATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % GC, wp = wave / GC;
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;

    const int L = d.L;
    const int ntiles = d.nlines * d.segs;
    const int nblk = gridDim.x;
    const int b0 = (int)xcd_tile(blockIdx.x, nblk);
    const int s1a = d.s1[0], s1b = d.s1[1], s1c = d.s1[2];
    const bool has_res = d.epi1.res != nullptr;
    const int K = b0 < ntiles ? (ntiles - b0 + nblk - 1) / nblk : 0;        // tiles of this block
    if (K == 0) return;
#ifdef ERF_LINE_STAMPS
    int stamp_i = 0;
#define X3_STAMP()                                                                                 \
    do {                                                                                           \
        if (d.stamps && tid == 0 && stamp_i < 48)                                                  \
            d.stamps[(size_t)blockIdx.x * 48 + stamp_i++] = __builtin_amdgcn_s_memtime();          \
    } while (0)
#else
#define X3_STAMP() do {} while (0)
#endif
    X3_STAMP();     // 0: start

    const float *in = reinterpret_cast<const float *>(d.in);
    float *out = reinterpret_cast<float *>(d.out);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (unsigned)((size_t)d.N * d.img_pixels * NC * 4));

    struct Geo { int q; unsigned base; int len; };
    auto geo = [&](int tile) __attribute__((always_inline)) {
        Geo g;
        const int line = tile / d.segs, seg = tile - line * d.segs;
        const int n = line / d.lines_per_img;
        g.q = line - n * d.lines_per_img;
        const int r0 = seg * BM;
        g.len = L - r0 < BM ? L - r0 : BM;
        g.base = (unsigned)n * d.img_pixels + (unsigned)g.q * d.qstride + (unsigned)r0 * d.pstride;
        return g;
    };
    auto tap_shift = [&](int t) __attribute__((always_inline)) { return t == 0 ? s1a : (t == 1 ? s1b : s1c); };
    auto tap_valid = [&](const Geo &g, int t) __attribute__((always_inline)) {
        return (unsigned)(g.q + tap_shift(t)) < (unsigned)d.lines_per_img;
    };
    // The block consumes a STREAM of operand tiles: the valid taps of its first line tile, then of
    // its second, ...  A cursor names one of them (dead = past the end: loads nothing).  The block's
    // first tile takes its taps in the order 1, 0, 2: the centre tap is valid on every line, so the
    // first tap a block runs is always tap 1 (its weights are the only ones the prologue loads).
    struct Cur { int k, i; unsigned soff; int len; };       // i = position in the tile's tap order
    auto tap_at = [&](int k, int i) __attribute__((always_inline)) { return k == 0 ? (i == 0 ? 1 : (i == 1 ? 0 : 2)) : i; };
    auto cur_at = [&](int k, int i, const Geo &g) __attribute__((always_inline)) {
        Cur c;
        c.k = k; c.i = i; c.len = g.len;
        c.soff = (g.base + (unsigned)(tap_shift(tap_at(k, i)) * d.qstride)) * (unsigned)(NC * 4);
        return c;
    };
    auto cur_first = [&](int k) __attribute__((always_inline)) {
        const Geo g = geo(b0 + k * nblk);
        int i = 0;
        while (!tap_valid(g, tap_at(k, i))) ++i;    // the centre tap is always valid
        return cur_at(k, i, g);
    };
    auto cur_next = [&](const Cur &c) __attribute__((always_inline)) {
        if (c.len == 0) return c;
        const Geo g = geo(b0 + c.k * nblk);
        int ni = c.i + 1;
        while (ni < 3 && !tap_valid(g, tap_at(c.k, ni))) ++ni;
        if (ni < 3) return cur_at(c.k, ni, g);
        if (c.k + 1 < K) return cur_first(c.k + 1);
        Cur e = c;
        e.len = 0; e.soff = 0;
        return e;
    };

    // this thread's share of an operand tile: float4 number c4 of rows r0 + p * RPP
    const int r0 = tid / F4R, c4 = tid % F4R;
    const unsigned vo0 = (unsigned)(r0 * d.pstride * NC + c4 * 4) * 4u;
    const unsigned vstep = (unsigned)(RPP * d.pstride * NC) * 4u;
    const int wbase = r0 * PITCH + (c4 & 1) * 8, cw = c4 >> 1;
    f32x4 stg[PASSES];
    auto load_pass = [&](int p, const Cur &c) __attribute__((always_inline)) {
        const unsigned vo = (r0 + p * RPP < c.len) ? vo0 + (unsigned)p * vstep : ERF_OOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, (int)c.soff, 0);
        stg[p] = __builtin_bit_cast(f32x4, v);
    };
    auto plane_off = [&](int p) __attribute__((always_inline)) {
        const int r = p * RPP + r0;
        return p * RPP * PITCH + wbase + ((cw ^ line_swz<NC>(r)) << 4);
    };

    // LDS-DMA of the residual tile (fp32, 16-byte chunks of a row XOR-swizzled by the row index)
    auto issue_res = [&](const Geo &g) __attribute__((always_inline)) {
        const float *res = reinterpret_cast<const float *>(d.epi1.res);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int r = p * RPP + r0;
            const int c = c4 ^ (r & (F4R - 1));
            const int rr = r < g.len ? r : g.len - 1;
            const float *gp = res + ((size_t)(g.base + (unsigned)(rr * d.pstride)) * NC + c * 4);
            lds_dma16(gp, lds0 + OFF_RES + (p * 256 + wave * 64) * 16);
        }
    };

    // ---- prologue: first operand tile into registers, weights, constants, first tile split into LDS
    Cur c1 = cur_first(0);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) load_pass(p, c1);
    c1 = cur_next(c1);

    // A CU's 288 KB of weight fragments need ~5 k cycles of its vector-memory path: only the weights
    // of the block's first tap are loaded here; those of the other two taps are requested piecewise
    // behind the MFMAs of that first tap (run_tap, FIRST) and have arrived when it ends.
    bf16x8 w[3][3][KS];         // [plane][tap][k-step]
    auto load_w = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int P = 0; P < 3; ++P)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                w[P][t][s] = *reinterpret_cast<const bf16x8 *>(d.wp1 + P * PL + ((size_t)((t * KS + s) * GC + cg) * 64 + lane) * 8);
    };
    load_w(1);

    for (int c = tid; c < NC; c += 256) {
        float sc, sh;
        epi_channel_affine(d.epi1, c, sc, sh);
        cst[c] = sc; cst[NC + c] = sh;
        if (STATS) { run[c] = 0.0; run[NC + c] = 0.0; }
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const f32x4 x = stg[p];
        const f32x2 xa = f32x2{x[0], x[1]}, xb = f32x2{x[2], x[3]};
        const unsigned ha = x3_cvt2(xa), hb = x3_cvt2(xb);
        const f32x2 ra = x3_sub_bf16x2(xa, ha), rb = x3_sub_bf16x2(xb, hb);
        const unsigned ma = x3_cvt2(ra), mb = x3_cvt2(rb);
        const unsigned la = x3_cvt2(x3_sub_bf16x2(ra, ma)), lb = x3_cvt2(x3_sub_bf16x2(rb, mb));
        unsigned char *dst = lds + OFF_TAP + plane_off(p);
        *reinterpret_cast<u32x2 *>(dst) = u32x2{ha, hb};
        *reinterpret_cast<u32x2 *>(dst + TP) = u32x2{ma, mb};
        *reinterpret_cast<u32x2 *>(dst + 2 * TP) = u32x2{la, lb};
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) load_pass(p, c1);     // second tile of the stream (dead: zeros)
    Cur c2 = cur_next(c1);
    X3_STAMP();     // 1: prologue issued
    lds_barrier();              // (orders LDS only: the weight loads stay in flight)
    X3_STAMP();     // 2: prologue done

    f32x16 acc[WM];
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    };

    // One tap: 6 x WM MFMAs per k-step (terms lh hl mm mh hm hh per pixel tile), one piece of side
    // work behind each: the fragment reads of the next k-step, then the split of a float4 of the
    // staged tile into the other TAP buffer and the load of the same float4 of the tile after it.
    auto run_tap = [&](auto tap_c, auto first_c, int seq, const Cur &nx) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;    // the block's first tap (tap 1): fetches the other taps' weights
        static_assert(!FIRST || TAP == 1, "a block starts with the centre tap");
        constexpr int OT1 = 0, OT2 = 2;
        const unsigned char *tb = lds + OFF_TAP + (seq & 1) * T3;
        unsigned char *nb = lds + OFF_TAP + ((seq + 1) & 1) * T3;
        const unsigned char *rowp[WM];
        unsigned x4[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            const int px = (wp * WM + a) * 32 + l31;
            rowp[a] = tb + px * PITCH;
            x4[a] = (unsigned)((line_swz<NC>(px) ^ h) << 4);
        }
        bf16x8 xf[2][3][WM];
#pragma unroll
        for (int P = 0; P < 3; ++P)
#pragma unroll
            for (int a = 0; a < WM; ++a)
                xf[0][P][a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + P * TP + x4[a]);
        f32x2 ra, rb;
        unsigned ha = 0, hb = 0, ma = 0, mb = 0, la = 0, lb = 0;
        // piece q (0..5) of the conversion of staged float4 p
        auto piece = [&](int p, int q) __attribute__((always_inline)) {
            if (q == 0) {
                const f32x2 xa = f32x2{stg[p][0], stg[p][1]};
                ha = x3_cvt2(xa);
                ra = x3_sub_bf16x2(xa, ha);
            } else if (q == 1) {
                ma = x3_cvt2(ra);
                la = x3_cvt2(x3_sub_bf16x2(ra, ma));
            } else if (q == 2) {
                const f32x2 xb = f32x2{stg[p][2], stg[p][3]};
                hb = x3_cvt2(xb);
                rb = x3_sub_bf16x2(xb, hb);
            } else if (q == 3) {
                mb = x3_cvt2(rb);
                lb = x3_cvt2(x3_sub_bf16x2(rb, mb));
            } else if (q == 4) {
                unsigned char *dst = nb + plane_off(p);
                *reinterpret_cast<u32x2 *>(dst) = u32x2{ha, hb};
                *reinterpret_cast<u32x2 *>(dst + TP) = u32x2{ma, mb};
                *reinterpret_cast<u32x2 *>(dst + 2 * TP) = u32x2{la, lb};
            } else {
                load_pass(p, nx);
            }
        };
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < 6 * WM; ++j) {
                const int a = j % WM, term = j / WM;
                // (weight plane, operand plane): small terms first
                const int pw = term == 0 ? 2 : (term == 1 || term == 4 || term == 5 ? 0 : 1);
                const int px = term == 1 ? 2 : (term == 0 || term == 3 || term == 5 ? 0 : 1);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[pw][TAP][s], xf[s & 1][px][a], acc[a], 0, 0, 0);
                if (j < 3 * WM && s + 1 < KS) {
                    const int P = j / WM, fa = j % WM;
                    xf[(s + 1) & 1][P][fa] =
                        *reinterpret_cast<const bf16x8 *>(rowp[fa] + P * TP + ((unsigned)((s + 1) << 5) ^ x4[fa]));
                }
                if (FIRST && j < 6) {
                    const int P = j % 3, ot = j < 3 ? OT1 : OT2;
                    w[P][ot][s] = *reinterpret_cast<const bf16x8 *>(d.wp1 + P * PL + ((size_t)((ot * KS + s) * GC + cg) * 64 + lane) * 8);
                }
                // the six pieces of a conversion: one per slot (WM = 1), or the second half of the slots
                if (WM == 1) {
                    if (s % CSTRIDE == 0) piece(s / CSTRIDE, j);
                } else if (j >= 6) {
                    piece(s, j - 6);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // accumulators -> fp32 tile in LDS: v = acc*sc + sh [+ residual tile, in place] -> [relu]
    auto store_acc_t = [&](auto relu_t, auto scale_t, auto res_t) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_t)::value, SCALE = decltype(scale_t)::value, RES = decltype(res_t)::value;
        f32x4 sc[4], sh[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = 32 * cg + 8 * gq + 4 * h;
            if (SCALE) sc[gq] = *reinterpret_cast<const f32x4 *>(cst + ch);
            sh[gq] = *reinterpret_cast<const f32x4 *>(cst + NC + ch);
        }
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            const int px = (wp * WM + a) * 32 + l31;
            const int sw = px & (F4R - 1);
            f32x4 rr[4];
            if (RES) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    rr[gq] = *reinterpret_cast<const f32x4 *>(lds + OFF_RES + px * FPITCH + (((8 * cg + 2 * gq + h) ^ sw) << 4));
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v = f32x4{acc[a][4 * gq], acc[a][4 * gq + 1], acc[a][4 * gq + 2], acc[a][4 * gq + 3]};
                if (SCALE) v = v * sc[gq] + sh[gq];
                else v = v + sh[gq];
                if (RES) v = v + rr[gq];
                if (RELU) v = __builtin_elementwise_max(v, f32x4{0.0f, 0.0f, 0.0f, 0.0f});
                *reinterpret_cast<f32x4 *>(lds + OFF_RES + px * FPITCH + (((8 * cg + 2 * gq + h) ^ sw) << 4)) = v;
            }
        }
    };
    auto store_acc = [&]() __attribute__((always_inline)) {
        using T1 = std::true_type;
        using T0 = std::false_type;
        const int key = (d.epi1.relu ? 4 : 0) | (d.epi1.bn.on ? 2 : 0) | (has_res ? 1 : 0);
        switch (key) {          // wave-uniform
        case 0: store_acc_t(T0{}, T0{}, T0{}); break;
        case 1: store_acc_t(T0{}, T0{}, T1{}); break;
        case 2: store_acc_t(T0{}, T1{}, T0{}); break;
        case 3: store_acc_t(T0{}, T1{}, T1{}); break;
        case 4: store_acc_t(T1{}, T0{}, T0{}); break;
        case 5: store_acc_t(T1{}, T0{}, T1{}); break;
        case 6: store_acc_t(T1{}, T1{}, T0{}); break;
        default: store_acc_t(T1{}, T1{}, T1{}); break;
        }
    };
    // coalesced pass over the finished fp32 tile: [mask by a global tensor] -> global store.  Buffer
    // accesses with the tile's base as the scalar offset: rows beyond the line present an out-of-range
    // offset (loads return zero, stores are dropped), no 64-bit address arithmetic.
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, (unsigned)((size_t)d.N * d.img_pixels * NC * 4));
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(d.epi1.mask, (unsigned)((size_t)d.N * d.img_pixels * NC * 4));
    const unsigned so0 = (unsigned)(r0 * d.pstride * NC) * 4u;
    auto sweep_tile = [&](const Geo &g) __attribute__((always_inline)) {
        const bool masked = d.epi1.mask != nullptr;
        const unsigned soff = g.base * (unsigned)(NC * 4);
        constexpr int B = 4;
#pragma unroll
        for (int p0 = 0; p0 < PASSES; p0 += B) {
            f32x4 v[B], m[B];
            unsigned vo[B];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const int p = p0 + i;
                const int r = p * RPP + r0;
                const int c = c4 ^ (r & (F4R - 1));
                vo[i] = r < g.len ? so0 + (unsigned)p * vstep + (unsigned)(c * 16) : ERF_OOB;
                v[i] = *reinterpret_cast<const f32x4 *>(lds + OFF_RES + (p * 256 + tid) * 16);
            }
            if (masked) {
#pragma unroll
                for (int i = 0; i < B; ++i)
                    m[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rmk, (int)vo[i], (int)soff, 0));
#pragma unroll
                for (int i = 0; i < B; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][e] = m[i][e] > 0.0f ? v[i][e] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < B; ++i)
                buf_store4_soff(__builtin_bit_cast(u32x4, v[i]), rout, (int)vo[i], (int)soff);
        }
    };
    // Optional statistics of the output (train-mode BatchNorm: the tile holds conv + bias, no mask or
    // residual follows): a thread sums ITS channel quad over the tile's rows (it reads the chunk where
    // the swizzle put that quad), leaves the partial sums in LDS, and after the tile's last barrier
    // the first F4R threads add the RPP partial rows to the block's running sums in double — a fixed
    // order, so the statistics are bitwise reproducible.
    float *spt = reinterpret_cast<float *>(lds + OFF_SPT);
    auto stats_tile = [&](const Geo &g) __attribute__((always_inline)) {
        f32x4 s1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, s2 = s1;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int r = p * RPP + r0;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(lds + OFF_RES + (r * F4R + (c4 ^ (r & (F4R - 1)))) * 16);
            if (r < g.len) { s1 = s1 + v; s2 = s2 + v * v; }
        }
        *reinterpret_cast<f32x4 *>(spt + tid * 8) = s1;
        *reinterpret_cast<f32x4 *>(spt + tid * 8 + 4) = s2;
    };
    auto stats_fold = [&]() __attribute__((always_inline)) {
        if (tid < F4R) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = 0.0, b = 0.0;
                for (int q = 0; q < RPP; ++q) {
                    a += (double)spt[(q * F4R + tid) * 8 + e];
                    b += (double)spt[(q * F4R + tid) * 8 + 4 + e];
                }
                run[4 * tid + e] += a;
                run[NC + 4 * tid + e] += b;
            }
        }
    };

    clear_acc();
    int seq = 0;
    auto tile_body = [&](auto first_tile_c, int k) __attribute__((always_inline)) {
        constexpr bool FT = decltype(first_tile_c)::value;
        using F1 = std::true_type;
        using F0 = std::false_type;
        const Geo g = geo(b0 + k * nblk);
        if (has_res) issue_res(g);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = FT ? (i == 0 ? 1 : (i == 1 ? 0 : 2)) : i;
            if (!tap_valid(g, t)) continue;                 // wave-uniform
            if (FT && i == 0) run_tap(std::integral_constant<int, 1>{}, F1{}, seq, c2);
            else if (t == 0) run_tap(std::integral_constant<int, 0>{}, F0{}, seq, c2);
            else if (t == 1) run_tap(std::integral_constant<int, 1>{}, F0{}, seq, c2);
            else run_tap(std::integral_constant<int, 2>{}, F0{}, seq, c2);
            c2 = cur_next(c2);
            ++seq;
            X3_STAMP();     // tap MFMAs issued
            bool last = true;
#pragma unroll
            for (int u = i + 1; u < 3; ++u) last = last && !tap_valid(g, FT ? (u == 1 ? 0 : 2) : u);
            // the residual tile (this wave's part) has landed before the barrier that opens the epilogue
            if (last && has_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();      // next operand tile complete in LDS; this one may be overwritten
            X3_STAMP();     // tap barrier passed
        }
        store_acc();
        clear_acc();
        X3_STAMP();         // accumulators in the fp32 tile
        lds_barrier();
        X3_STAMP();
        sweep_tile(g);
        if (want_stats) stats_tile(g);          // (block-uniform)
        X3_STAMP();         // output pass issued
        lds_barrier();          // the fp32 tile may be refilled; the stores stay in flight
        if (want_stats) stats_fold();           // (the partial rows are rewritten three barriers from here)
        X3_STAMP();
    };
    tile_body(std::true_type{}, 0);
    for (int k = 1; k < K; ++k) tile_body(std::false_type{}, k);
    if (want_stats && tid < F4R) {              // the threads that kept the running sums
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d.stats[(size_t)blockIdx.x * 2 * NC + 4 * tid + e] = run[4 * tid + e];
            d.stats[(size_t)blockIdx.x * 2 * NC + NC + 4 * tid + e] = run[NC + 4 * tid + e];
        }
    }
}
