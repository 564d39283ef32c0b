// 1-D convolutions of a non_bottleneck_1d on bf16 activations (gfx950, v_mfma_f32_32x32x16_bf16),
// one or two chained convolutions per launch:
//
//   mid = epi1( conv_a(in) )        conv_a runs ACROSS lines  (tap t reads line q + s1[t])
//   out = epi2( conv_b(mid) )       conv_b runs ALONG a line  (tap t reads position p + s2[t])   [PAIR]
//
// At the bf16 MFMA rate (16x the fp32 one) a convolution of NC = 64 / 128 channels is no longer bound
// by arithmetic but by what a CU can pull in per clock, so nothing is fetched twice that can stay:
//   * weights are STATIONARY IN REGISTERS: a wave owns 32 output channels and keeps the MFMA
//     fragments of all 3 x NC contraction steps of both convolutions (96 VGPRs per convolution at
//     NC = 128) for its whole life; a block is persistent and walks many lines;
//   * activations reach LDS by LDS-DMA (global_load_lds_dwordx4, no registers, no ds_write), one
//     line tile per tap, double buffered, the tile of the next tap / next line in flight during
//     the MFMAs of the current one;
//   * the product is formed transposed, D[channel][pixel] = W[channel][k] * X[k][pixel], so a lane
//     ends up with 4 consecutive channels of one pixel per accumulator quad: the intermediate tile
//     (bf16, the second convolution's operand) is written with 8-byte ds_write, and the output
//     goes through an LDS tile to full-row 16-byte global stores (residual and mask tensors are read
//     the same way, coalesced).
// LDS tiles are [pixel][NC] bf16 with the 16-byte chunks of a row XOR-swizzled by the row index, so
// that the fragment reads (lanes = 32 pixels, also tap-shifted), the 8-byte accumulator writes and
// the linear DMA / copy passes are all free of bank conflicts.
// Arithmetic: bf16 operands, fp32 accumulation; bias / BatchNorm affine / residual in fp32, then
// one rounding to bf16 (the intermediate is rounded once, after its ReLU).
#pragma once
#include "common.h"
#include <type_traits>

struct LineDesc {
    const bf16_t *in;
    const bf16_t *wp1, *wp2;    // weight fragments (line_pack_kernel); wp2 unused without PAIR
    bf16_t *mid;                // may be null: the intermediate never leaves the CU
    bf16_t *out;
    Epilogue epi1, epi2;        // nscale unsupported; res / mask are dense NC-channel bf16 tensors
    int N, nlines, lines_per_img, img_pixels;
    int L;                      // pixels per line
    int segs;                   // segments per line (1 with PAIR); a block tile covers BM pixels
    int pstride;                // pixel-index distance between neighbours on a line
    int qstride;                // pixel-index distance between the bases of neighbouring lines
    int s1[3], s2[3];
    double *stats;              // line_x3_kernel only, may be null: [blocks][2 * NC] per-block column sums of the
                                // output and of its squares (train-mode BatchNorm statistics)
#ifdef ERF_LINE_STAMPS
    unsigned long long *stamps;     // diagnostic build only: [block][32] shader-clock stamps of wave 0
#endif
};

// Weight fragments: wp[tap][k16 step s][channel group cg][lane][8] with
//   channel = 32 cg + (lane & 31), k = 16 s + 8 (lane >> 5) + j   — the A operand of the MFMA as a
// lane holds it, so that a wave fetches a fragment as one contiguous KiB.
struct LinePackDesc {
    const float *w;
    bf16_t *wp;
    int NC, wsk, wsn;           // W[tap][k][j] = w[woff[tap] + k*wsk + j*wsn]
    int woff[3];
    int x3;                     // 1 = three planes wp[plane][...]: the value split into bf16 hi + mid + lo (line_x3_kernel)
};
#define ERF_LINE_PACK_BATCH 32
struct LinePackBatch {
    LinePackDesc p[ERF_LINE_PACK_BATCH];
};

__global__ __launch_bounds__(256) void line_pack_kernel(const LinePackBatch b)
{
    const LinePackDesc &d = b.p[blockIdx.y];
    const int NC = d.NC, KS = NC / 16, GC = NC / 32;
    const int total = 3 * KS * GC * 64 * 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int j = idx & 7, lane = (idx >> 3) & 63;
        int r = idx >> 9;
        const int cg = r % GC; r /= GC;
        const int s = r % KS, t = r / KS;
        const int ch = 32 * cg + (lane & 31), k = 16 * s + 8 * (lane >> 5) + j;
        const float v = d.w[d.woff[t] + (long long)k * d.wsk + (long long)ch * d.wsn];
        const bf16_t hi = f32_to_bf16(v);
        d.wp[idx] = hi;
        if (d.x3) {
            const float r1 = v - bf16_to_f32(hi);
            const bf16_t mid = f32_to_bf16(r1);
            d.wp[total + idx] = mid;
            d.wp[2 * total + idx] = f32_to_bf16(r1 - bf16_to_f32(mid));
        }
    }
}

template <int NC>
__device__ __forceinline__ int line_swz(int row)
{
    return NC == 128 ? (row & 15) : ((row >> 1) & 7);
}

__device__ __forceinline__ void lds_dma16(const void *gsrc, unsigned lds_byte_addr)
{
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gsrc,
                                     (void __attribute__((address_space(3))) *)(uintptr_t)lds_byte_addr, 16, 0, 0);
}

// bf16 patterns: zero the elements of v whose mask element is not > 0
// The same with the mask given as 8 bits (bit j = keep element j; higher bits ignored), and the bits of
// a vector of ReLU outputs (no element negative: "> 0" is "not zero" on the bit pattern).  The ReLU
// masks travel as bit tensors, word [pixel][c / 32], bit c % 32.  Packed 16-bit arithmetic: a dozen
// instructions per 8 elements either way.
__device__ __forceinline__ u32x4 mask_bits_bf16x8(u32x4 v, unsigned bits)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    bits &= 0xffu;
    const unsigned s = bits | (bits << 15);         // bit 2e -> bit 2e, bit 2e + 1 -> bit 2e + 16
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const u16x2 one = __builtin_bit_cast(u16x2, (s >> (2 * e)) & 0x00010001u);
        const u16x2 keep = u16x2{0, 0} - one;       // 0xffff where the bit is set
        o[e] = v[e] & __builtin_bit_cast(unsigned, keep);
    }
    return o;
}
__device__ __forceinline__ unsigned sign_bits_bf16x8(u32x4 v)
{
    unsigned b = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned nz;                    // min(x, 1) on both halves (the compiler would expand it into compares)
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(v[e]), "s"(0x00010001u));
        b |= nz << (2 * e);             // bits 2e and 2e + 16
    }
    return (b | (b >> 15)) & 0xffu;
}

__device__ __forceinline__ u32x4 mask_bf16x8(u32x4 v, u32x4 m)
{
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned lo = m[e] & 0xffffu, hi = m[e] >> 16;
        // "> 0" on the bit pattern: sign clear and not zero
        const unsigned klo = (lo != 0u && lo < 0x8000u) ? 0x0000ffffu : 0u;
        const unsigned khi = (hi != 0u && hi < 0x8000u) ? 0xffff0000u : 0u;
        o[e] = v[e] & (klo | khi);
    }
    return o;
}

// grid = persistent blocks (<= number of line tiles); 256 threads; one block per CU.
template <int NC, int BM, bool PAIR>
__global__ __launch_bounds__(256, 1) void line_bf16_kernel(const LineDesc d)
{
    constexpr int GC = NC / 32, GP = 4 / GC, WM = BM / (32 * GP);
    static_assert(WM >= 1 && WM * GP * 32 == BM, "tile shape");
    constexpr int KS = NC / 16;             // MFMA k-steps per tap
    constexpr int CPR = NC / 8;             // 16-byte chunks per pixel row
    constexpr int PITCH = NC * 2;           // bytes per pixel row
    constexpr int T = BM * PITCH;           // bytes per line tile
    constexpr int PASSES = BM * CPR / 256;
    static_assert(BM * CPR % 256 == 0, "whole copy passes");
    constexpr int OFF_TAP = 0, OFF_MID = 2 * T, OFF_RES = PAIR ? 3 * T + PITCH : 2 * T;
    constexpr int OFF_CST = OFF_RES + T;
    __shared__ __attribute__((aligned(16))) unsigned char lds[OFF_CST + 4 * NC * 4];
    float *cst = reinterpret_cast<float *>(lds + OFF_CST);        // sc1[NC] sh1[NC] sc2[NC] sh2[NC]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ERF_LINE_STAMPS
    int stamp_i = 0;
#define LINE_STAMP()                                                                               \
    do {                                                                                           \
        if (d.stamps && tid == 0 && stamp_i < 32)                                                  \
            d.stamps[(size_t)blockIdx.x * 32 + stamp_i++] = __builtin_amdgcn_s_memtime();          \
    } while (0)
#else
#define LINE_STAMP() do {} while (0)
#endif
    LINE_STAMP();
    const int cg = wave % GC, wp = wave / GC;
    const int l31 = lane & 31, h = lane >> 5;
    // LDS byte address of the array (the DMA's destination is an LDS address, not a generic pointer)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;

    const int L = d.L;
    const int ntiles = d.nlines * d.segs;
    const int nblk = gridDim.x;
    const int b0 = (int)xcd_tile(blockIdx.x, nblk);
    const int s1a = d.s1[0], s1b = d.s1[1], s1c = d.s1[2];
    // One residual tensor per launch, added in the epilogue that names it: the final one, or (a pair
    // whose first convolution closes a non_bottleneck_1d) the intermediate's.
    const bool res_first = PAIR && d.epi1.res != nullptr;
    const bool has_res = res_first || (PAIR ? d.epi2.res : d.epi1.res) != nullptr;

    // (every lambda below is forced inline: an outlined one makes the accumulators it captures by
    // reference live in scratch memory, which costs a memory round trip per MFMA)
    // tile index -> geometry (wave-uniform)
    struct Geo { int q; unsigned base; int r0, len; };
    auto geo = [&](int tile) __attribute__((always_inline)) {
        Geo g;
        const int line = tile / d.segs, seg = tile - line * d.segs;
        const int n = line / d.lines_per_img;
        g.q = line - n * d.lines_per_img;
        g.r0 = seg * BM;
        g.len = L - g.r0 < BM ? L - g.r0 : BM;
        g.base = (unsigned)n * d.img_pixels + (unsigned)g.q * d.qstride + (unsigned)g.r0 * d.pstride;
        return g;
    };
    auto tap_shift = [&](int t) __attribute__((always_inline)) { return t == 0 ? s1a : (t == 1 ? s1b : s1c); };
    auto tap_valid = [&](const Geo &g, int t) __attribute__((always_inline)) { return (unsigned)(g.q + tap_shift(t)) < (unsigned)d.lines_per_img; };

    // LDS-DMA of one line tile (rows beyond the line repeat its last row: their products are dropped)
    auto issue_tile = [&](const bf16_t *src, unsigned pixel0, int len, int dst_off) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int slot = p * 256 + tid;
            const int r = slot / CPR, pc = slot % CPR;
            const int c = pc ^ line_swz<NC>(r);
            const int rr = r < len ? r : len - 1;
            const bf16_t *g = src + ((size_t)(pixel0 + (unsigned)(rr * d.pstride)) * NC + c * 8);
            lds_dma16(g, lds0 + dst_off + (p * 256 + wave * 64) * 16);
        }
    };

    int tile = b0;
    if (tile >= ntiles) return;
    Geo g = geo(tile);
    int seq = 0;            // tap tiles consumed so far: tile number seq lives in TAP buffer seq & 1
    // first valid tap of the first tile
    {
        int t = 0;
        while (!tap_valid(g, t)) ++t;        // the centre tap is always valid
        issue_tile(d.in, g.base + (unsigned)(tap_shift(t) * d.qstride), g.len, OFF_TAP);
    }
    // ---- per-block set-up, behind the first tile's DMA: weights into registers, epilogue constants,
    // zero row
    bf16x8 w1[3][KS], w2[PAIR ? 3 : 1][PAIR ? KS : 1];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const size_t f = ((size_t)((t * KS + s) * GC + cg) * 64 + lane) * 8;
            w1[t][s] = *reinterpret_cast<const bf16x8 *>(d.wp1 + f);
            if constexpr (PAIR) w2[t][s] = *reinterpret_cast<const bf16x8 *>(d.wp2 + f);
        }

    for (int c = tid; c < NC; c += 256) {
        float sc, sh;
        epi_channel_affine(d.epi1, c, sc, sh);
        cst[c] = sc; cst[NC + c] = sh;
        if constexpr (PAIR) {
            epi_channel_affine(d.epi2, c, sc, sh);
            cst[2 * NC + c] = sc; cst[3 * NC + c] = sh;
        }
    }
    if constexpr (PAIR)
        for (int i = tid; i < PITCH / 4; i += 256) reinterpret_cast<unsigned *>(lds + OFF_MID + T)[i] = 0u;

    f32x16 acc[WM];
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    };

    // MFMAs of one tap: operand rows rowp[a] (+ swizzle term x4[a]) against the weight fragments w[s].
    // The fragments of k-step s+1 are read while the MFMAs of step s run (the compiler, left alone,
    // waits for every read right in front of its MFMA: with one wave per SIMD nothing else hides the
    // LDS latency).
    auto mma_tap = [&](const bf16x8 (&w)[KS], const unsigned char *const (&rowp)[WM], const unsigned (&x4)[WM]) __attribute__((always_inline)) {
        constexpr int D = WM >= 4 ? 2 : 3;      // fragment sets: reads run D - 1 k-steps ahead of their MFMAs
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

    // Coalesced pass over a finished LDS tile (bf16, swizzled): [mask by a global tensor, written back
    // to the tile when `keep`] -> [global store].  Four 16-byte chunks per thread are in flight at a time.
    auto sweep_tile = [&](int lds_off, const bf16_t *mk, bf16_t *dst, bool keep, const Geo &g) __attribute__((always_inline)) {
        constexpr int B = PASSES % 4 == 0 ? 4 : (PASSES % 2 == 0 ? 2 : 1);
#pragma unroll
        for (int p0 = 0; p0 < PASSES; p0 += B) {
            u32x4 v[B], m[B];
            size_t off[B];
            bool live[B];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const int slot = (p0 + i) * 256 + tid;
                const int r = slot / CPR, pc = slot % CPR;
                const int c = pc ^ line_swz<NC>(r);
                live[i] = r < g.len;
                const int rr = live[i] ? r : g.len - 1;
                off[i] = (size_t)(g.base + (unsigned)(rr * d.pstride)) * NC + c * 8;
                v[i] = *reinterpret_cast<const u32x4 *>(lds + lds_off + slot * 16);
            }
            if (mk) {
#pragma unroll
                for (int i = 0; i < B; ++i) m[i] = *reinterpret_cast<const u32x4 *>(mk + off[i]);
#pragma unroll
                for (int i = 0; i < B; ++i) v[i] = mask_bf16x8(v[i], m[i]);
                if (keep) {
#pragma unroll
                    for (int i = 0; i < B; ++i) *reinterpret_cast<u32x4 *>(lds + lds_off + ((p0 + i) * 256 + tid) * 16) = v[i];
                }
            }
            if (dst) {
#pragma unroll
                for (int i = 0; i < B; ++i)
                    if (live[i]) *reinterpret_cast<u32x4 *>(dst + off[i]) = v[i];
            }
        }
    };

    // Accumulators -> bf16 tile in LDS: v = acc*sc + sh [+ res tile] -> [relu] -> [zero beyond the line].
    // The per-channel constants and a pixel tile's residual quads are read as batches ahead of the
    // arithmetic.
    // (the three switches are compile-time: evaluated per element they cost more instructions than
    // the arithmetic itself, and with one wave per SIMD nothing overlaps an epilogue)
    auto store_acc_t = [&](auto relu_t, auto scale_t, auto res_t, const float *sc_, const float *sh_, bool zero_dead,
                           int dst_off, const Geo &g) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_t)::value, SCALE = decltype(scale_t)::value, RES = decltype(res_t)::value;
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const bool ragged = zero_dead && g.len < BM;          // wave-uniform
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
                    rr[gq] = *reinterpret_cast<const u32x2 *>(lds + OFF_RES + px * PITCH + (((4 * cg + gq) ^ sw) << 4) + 8 * h);
            }
            u32x2 pk[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v = f32x4{acc[a][4 * gq], acc[a][4 * gq + 1], acc[a][4 * gq + 2], acc[a][4 * gq + 3]};
                // packed fp32 arithmetic (two elements per instruction); without BatchNorm the scale is 1
                if (SCALE) v = v * sc[gq] + sh[gq];
                else v = v + sh[gq];
                if (RES) v = v + unpack_bf16x4(rr[gq]);
                pk[gq] = pack_bf16x4(v);
                // ReLU on the bf16 patterns: as signed 16-bit integers every negative value is < 0
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
                         int dst_off, const Geo &g) __attribute__((always_inline)) {
        using T1 = std::true_type;
        using T0 = std::false_type;
        const int key = (e.relu ? 4 : 0) | (e.bn.on ? 2 : 0) | (add_res ? 1 : 0);
        switch (key) {          // wave-uniform
        case 0: store_acc_t(T0{}, T0{}, T0{}, sc_, sh_, zero_dead, dst_off, g); break;
        case 1: store_acc_t(T0{}, T0{}, T1{}, sc_, sh_, zero_dead, dst_off, g); break;
        case 2: store_acc_t(T0{}, T1{}, T0{}, sc_, sh_, zero_dead, dst_off, g); break;
        case 3: store_acc_t(T0{}, T1{}, T1{}, sc_, sh_, zero_dead, dst_off, g); break;
        case 4: store_acc_t(T1{}, T0{}, T0{}, sc_, sh_, zero_dead, dst_off, g); break;
        case 5: store_acc_t(T1{}, T0{}, T1{}, sc_, sh_, zero_dead, dst_off, g); break;
        case 6: store_acc_t(T1{}, T1{}, T0{}, sc_, sh_, zero_dead, dst_off, g); break;
        default: store_acc_t(T1{}, T1{}, T1{}, sc_, sh_, zero_dead, dst_off, g); break;
        }
    };

    // Final epilogue of a tile: accumulators (+ residual) -> RES tile (in place of the residual), then
    // a coalesced pass [mask] -> global.
    auto finish_tile = [&](const Epilogue &e, const float *sc_, const float *sh_, bf16_t *dst, const Geo &g) __attribute__((always_inline)) {
        store_acc(e, sc_, sh_, has_res && !res_first, false, OFF_RES, g);
        lds_barrier();
        sweep_tile(OFF_RES, (const bf16_t *)e.mask, dst, false, g);
        lds_barrier();          // the RES tile may be refilled; the stores stay in flight
    };

    clear_acc();
    LINE_STAMP();       // 1: set-up issued (weights, constants, first tap)
    while (true) {
        const bf16_t *resp = (const bf16_t *)(res_first ? d.epi1.res : (PAIR ? d.epi2.res : d.epi1.res));
        if (has_res) issue_tile(resp, g.base, g.len, OFF_RES);
        const int ntile = tile + nblk;
        const bool more = ntile < ntiles;
        Geo gn = g;
        if (more) gn = geo(ntile);

        // ------------------------------------------------------------ conv_a: operand tiles by DMA
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (!tap_valid(g, t)) continue;                  // wave-uniform
            // successor in the stream of tap tiles: the next valid tap of this tile, else the first
            // valid tap of the block's next tile
            int nt = t + 1;
            while (nt < 3 && !tap_valid(g, nt)) ++nt;
            // the tap's DMA (and everything older on the vector-memory counter) has landed in this
            // wave; after the barrier, in all of them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            LINE_STAMP();   // tap landed
            if (nt < 3) {
                issue_tile(d.in, g.base + (unsigned)(tap_shift(nt) * d.qstride), g.len, OFF_TAP + ((seq + 1) & 1) * T);
            } else if (more) {
                int ft = 0;
                while (!tap_valid(gn, ft)) ++ft;
                issue_tile(d.in, gn.base + (unsigned)(tap_shift(ft) * d.qstride), gn.len, OFF_TAP + ((seq + 1) & 1) * T);
            }
            LINE_STAMP();   // next DMA issued
            const unsigned char *tb = lds + OFF_TAP + (seq & 1) * T;
            const unsigned char *rowp[WM];
            unsigned x4[WM];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                const int px = (wp * WM + a) * 32 + l31;
                rowp[a] = tb + px * PITCH;
                x4[a] = (unsigned)((line_swz<NC>(px) ^ h) << 4);
            }
            mma_tap(w1[t], rowp, x4);
            LINE_STAMP();   // tap MFMAs done
            ++seq;
        }

        if constexpr (PAIR) {
            // -------------------------------------------------------- intermediate tile -> LDS (bf16)
            // (positions beyond the line are conv_b's zero padding)
            store_acc(d.epi1, cst, cst + NC, res_first, true, OFF_MID, g);
            clear_acc();
            lds_barrier();
            LINE_STAMP();   // intermediate in LDS
            const bf16_t *mk1 = (const bf16_t *)d.epi1.mask;
            if (d.mid || mk1) {                              // wave-uniform
                sweep_tile(OFF_MID, mk1, d.mid, true, g);
                if (mk1) lds_barrier();
            }
            // -------------------------------------------------------- conv_b: operand = intermediate tile
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int sft = d.s2[t];
                const unsigned char *rowp[WM];
                unsigned x4[WM];
#pragma unroll
                for (int a = 0; a < WM; ++a) {
                    const int rr = (wp * WM + a) * 32 + l31 + sft;
                    const int rz = (unsigned)rr < (unsigned)g.len ? rr : BM;        // beyond the line: the zero row
                    rowp[a] = lds + OFF_MID + rz * PITCH;
                    x4[a] = (unsigned)((line_swz<NC>(rz) ^ h) << 4);
                }
                mma_tap(w2[t], rowp, x4);
            }
            LINE_STAMP();   // conv_b MFMAs done
            finish_tile(d.epi2, cst + 2 * NC, cst + 3 * NC, d.out, g);
        } else {
            finish_tile(d.epi1, cst, cst + NC, d.out, g);
        }
        clear_acc();
        LINE_STAMP();       // tile stored
        if (!more) break;
        tile = ntile;
        g = gn;
    }
}
