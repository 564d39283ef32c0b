// Two chained 1-D convolutions of a non_bottleneck_1d in one launch (gfx950, fp32 MFMA 32x32x2).
//
//   mid = epi1( conv_a(in) )        conv_a runs ACROSS lines  (tap shift s1[t] lines)
//   out = epi2( conv_b(mid) )       conv_b runs ALONG a line  (tap shift s2[t] pixels of the line)
//
// A block owns one whole line of the image (an image column of H pixels for a "1x3 then 3x1" pair:
// conv_a reaches into the neighbouring columns of `in`, conv_b stays inside the column), so the
// second convolution needs no halo: its A operand is the block's own `mid` tile, kept in LDS.
// `mid` is also written to memory when the caller needs it (training saves it for backward).
// Compared with two igemm launches this removes one launch ramp, the re-read of `mid`, and lets the
// first epilogue of one block overlap the MFMAs of its neighbours on the CU.
//
// Both convolutions are NC -> NC channels (NC = 64 or 128), 3 taps, dense NHWC tensors.  The
// contraction visits taps, then channels in ascending groups of 8 — the order igemm_f32_kernel
// uses — so the results are bit-identical to the two-launch path.
#pragma once
#include "common.h"
#include <type_traits>

struct PairDesc {
    const float *in;
    const float *wp1, *wp2;     // packed weights [tap][NC/16][NC][16]
    float *mid;                 // may be null (inference: the intermediate never leaves the CU)
    float *out;
    Epilogue epi1, epi2;        // nscale is not supported (must be null)
    int N, nlines, lines_per_img, img_pixels;
    int L;                      // pixels per line (<= BM)
    int pstride;                // pixel-index distance between neighbours on a line
    int qstride;                // pixel-index distance between the bases of neighbouring lines
    int s1[3], s2[3];
};

// 16-byte chunk swizzle of the mid tile (row pitch = NC floats, a multiple of the 64 banks): any 16
// consecutive rows map to 16 different chunk slots (conflict-free ds_read_b128 with lanes = rows,
// also for tap-shifted rows), and rows r, r+4 land in different bank halves (the two lane halves of
// an accumulator store).
__device__ __forceinline__ int mid_swz(int row, int chunk)
{
    return chunk ^ ((row & 7) | ((((row >> 3) ^ (row >> 2)) & 1) << 3));
}

// Weight tile in LDS: [channel][16 k] floats, the four 16-byte chunks of a row permuted by the row
// index so that both the staging stores (a wave writes 1 KiB contiguous) and the fragment reads
// (lanes = 16 consecutive rows, one chunk each) touch every bank once.
__device__ __forceinline__ int bs_off(int row, int chunk)
{
    return row * 16 + ((chunk ^ ((row >> 2) & 3)) << 2);
}

// NW = waves per block: 4 for launches that fill the chip (three resident blocks per CU), 8 when
// there are fewer lines than CU slots (small batches), where a block's own latency sets the time.
// NS = 2 (very small launches): two blocks per line; both run the first convolution in full, each
// runs the second for one half of the output channels (blockIdx.y), which puts the idle CUs to work.
template <int BM, int NC, int NW, int NS = 1>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 1 : (BM * NC <= 8192 ? 3 : 2))) void pair_f32_kernel(const PairDesc d)
{
    constexpr int NC2 = NC / NS, GN2 = NC2 / 32, GM2 = NW / GN2, WM2 = BM / (32 * GM2);
    static_assert(GM2 >= 1 && WM2 >= 1 && GM2 * WM2 * 32 == BM, "tile shape of the second convolution");
    constexpr int KC = 16, CPR = KC / 4, BPITCH = KC, NT = NW * 64, KCH = NC / KC;
    constexpr int GN = NC / 32, GM = NW / GN, WM = BM / (32 * GM);
    static_assert(GM >= 1 && WM >= 1 && GM * WM * 32 == BM, "tile shape");
    constexpr int A_SLOTS = BM * CPR, B_SLOTS = NC * CPR;
    constexpr int A_PASSES = (A_SLOTS + NT - 1) / NT, B_PASSES = (B_SLOTS + NT - 1) / NT;
    constexpr bool A_FULL = A_SLOTS % NT == 0, B_FULL = B_SLOTS % NT == 0;
    constexpr int NSTEPS = 3 * KCH;
    // mid tile [BM][NC] + one row of zeros (taps of conv_b beyond the line read it); while conv_a
    // runs, its first 2*BM*KC floats are the double-buffered A staging area
    __shared__ __attribute__((aligned(16))) float Ts[(BM + 1) * NC];
    __shared__ __attribute__((aligned(16))) float Bs[2][NC * BPITCH];
    static_assert(2 * BM * KC <= BM * NC, "A staging fits inside the mid tile");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / GN, wn = wave % GN;
    const int l31 = lane & 31, h = lane >> 5;

    const unsigned line = xcd_tile(blockIdx.x, gridDim.x);
    const int n = (int)(line / (unsigned)d.lines_per_img);
    const int q = (int)(line - (unsigned)n * d.lines_per_img);
    const unsigned base = (unsigned)n * d.img_pixels + (unsigned)q * d.qstride;
    const int L = d.L;
    const int s1a = d.s1[0], s1b = d.s1[1], s1c = d.s1[2], s2a = d.s2[0], s2b = d.s2[1], s2c = d.s2[2];

    for (int i = tid; i < NC; i += NT) Ts[BM * NC + i] = 0.0f;

    // per-column constants of both epilogues (the second convolution has its own wave layout when
    // the output channels are split over two blocks)
    const int col = wn * 32 + l31;
    const int wm2 = wave / GN2, wn2 = wave % GN2;
    const int colb = wn2 * 32 + l31;                         // column within the block's B tile
    const int col2 = (NS > 1 ? (int)blockIdx.y * NC2 : 0) + colb;
    float sc1, sh1, sc2, sh2;
    epi_channel_affine(d.epi1, col, sc1, sh1);
    epi_channel_affine(d.epi2, col2, sc2, sh2);

    static_assert(WM2 <= WM, "accumulators are shared");
    f32x16 acc[WM];
    auto clear_acc = [&]() {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    };
    clear_acc();

    const unsigned P = (unsigned)d.N * d.img_pixels;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, P * NC * 4u);

    struct Stage {
        f32x4 a[A_PASSES], b[B_PASSES];
    };
    // ------------------------------------------------------------------ conv_a: A and B from memory
    auto load1 = [&](int s, Stage &R) {
        const int t = s / KCH, kc = s % KCH;
        const int sh = t == 0 ? s1a : (t == 1 ? s1b : s1c);
        const unsigned tapbad = (unsigned)(q + sh) < (unsigned)d.lines_per_img ? 0u : ERF_OOB;
        const unsigned tbase = base + (unsigned)(sh * d.qstride);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, ac = slot % CPR;
            unsigned off = ((tbase + (unsigned)(r * d.pstride)) * NC + kc * KC + ac * 4) * 4u;
            off |= tapbad | ((unsigned)((L - 1 - r) >> 31) & ERF_OOB);      // (covers slots beyond the tile: r >= BM >= L)
            R.a[i] = buf_load4(r_in, off);
        }
        const float *wt = d.wp1 + (long long)(t * KCH + kc) * NC * KC;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int slot = tid + i * NT;
            R.b[i] = *reinterpret_cast<const f32x4 *>(wt + (B_FULL || slot < B_SLOTS ? slot : 0) * 4);
        }
    };
    auto store_b = [&](int buf, const f32x4 *b) {
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int slot = tid + i * NT;
            if (B_FULL || slot < B_SLOTS)
                *reinterpret_cast<f32x4 *>(&Bs[buf][bs_off(slot / CPR, slot % CPR)]) = b[i];
        }
    };
    auto store1 = [&](int buf, const Stage &R) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, ac = slot % CPR;
            if (A_FULL || slot < A_SLOTS)
                *reinterpret_cast<f32x4 *>(&Ts[buf * BM * KC + r * KC + ((ac ^ ((r >> 2) & 3)) << 2)]) = R.a[i];
        }
        store_b(buf, R.b);
    };
    // Step s multiplies the tiles staged in buffer s & 1.  Its loads were issued two steps earlier
    // (at step s - 3 into R, stored to LDS at step s - 1), so a load has two whole steps to land;
    // the barrier orders LDS traffic only and leaves the loads in flight.
    auto step1 = [&](int s, Stage &R) {
        const int buf = s & 1;
        if (s + 1 < NSTEPS) store1(buf ^ 1, R);
        if (s + 3 < NSTEPS) load1(s + 3, R);
        const float *Ab = Ts + buf * BM * KC, *Bb = Bs[buf];
#pragma unroll
        for (int qq = 0; qq < KC / 8; ++qq) {
            f32x4 af[WM], bf;
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                const int row = (wm * WM + a) * 32 + l31;
                af[a] = *reinterpret_cast<const f32x4 *>(&Ab[row * KC + (((2 * qq + h) ^ ((row >> 2) & 3)) << 2)]);
            }
            bf = *reinterpret_cast<const f32x4 *>(&Bb[bs_off(col, 2 * qq + h)]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int a = 0; a < WM; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][m], bf[m], acc[a], 0, 0, 0);
        }
        lds_barrier();
    };

    static_assert(NSTEPS % 2 == 0, "two register stages alternate");
    {
        Stage stA, stB;             // stA: odd steps' tiles, stB: even steps'
        load1(0, stB);
        load1(1, stA);
        store1(0, stB);
        load1(2, stB);
        lds_barrier();
        for (int s = 0; s < NSTEPS; s += 2) {
            step1(s, stA);
            step1(s + 1, stB);
        }
    }

    // ------------------------------------------------------------------ epilogues
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // All loads of an epilogue (residual, mask) are issued before the first store: the compiler
    // must assume that a store may alias a later load, so interleaving them would expose one
    // memory latency per group of rows.  Absent tensors are skipped with wave-uniform branches.
    auto epilogue = [&](const Epilogue &e, float sc, float sh, float *dst, bool to_lds, auto wtag, int wmx, int col) {
        constexpr int WMX = decltype(wtag)::value;
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(dst, P * NC * 4u);
        const __amdgpu_buffer_rsrc_t r_res = make_rsrc(e.res, P * e.res_C * 4u);
        const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(e.mask, P * e.mask_C * 4u);
        const bool has_mask = e.mask != nullptr, has_res = e.res != nullptr, relu = e.relu != 0;
        // row of accumulator element r: a wave-uniform part (scalar offset operand of the buffer
        // instructions) plus 4 * (lane >> 5) (part of the per-lane offset)
        auto urow = [&](int a, int r) { return (wmx * WMX + a) * 32 + (r & 3) + 8 * (r >> 2); };
        const unsigned px0 = base + (unsigned)(4 * h * d.pstride);
        float rres[WMX][16], rmask[WMX][16];
#pragma unroll
        for (int a = 0; a < WMX; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) { rres[a][r] = 0.0f; rmask[a][r] = 1.0f; }
        // rows beyond the line need no guard on the load side: the value is not used, and an
        // address past the tensor is out of range for the descriptor
        if (has_res) {
            const unsigned voff = (px0 * e.res_C + e.res_coff + col) * 4u;
            const unsigned rowb = (unsigned)(d.pstride * e.res_C) * 4u;
#pragma unroll
            for (int a = 0; a < WMX; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rres[a][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        r_res, (int)voff, (int)__builtin_amdgcn_readfirstlane(urow(a, r) * rowb), 0));
        }
        if (has_mask) {
            const unsigned voff = (px0 * e.mask_C + e.mask_coff + col) * 4u;
            const unsigned rowb = (unsigned)(d.pstride * e.mask_C) * 4u;
#pragma unroll
            for (int a = 0; a < WMX; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rmask[a][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        r_mask, (int)voff, (int)__builtin_amdgcn_readfirstlane(urow(a, r) * rowb), 0));
        }
        const unsigned ovoff = (px0 * NC + col) * 4u;
        const unsigned orowb = (unsigned)(d.pstride * NC) * 4u;
#pragma unroll
        for (int a = 0; a < WMX; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rw = urow(a, r) + 4 * h;
                float v = acc[a][r] * sc + sh;
                v += rres[a][r];
                if (relu) v = fmaxf(v, 0.0f);
                v = rmask[a][r] > 0.0f ? v : 0.0f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_out,
                                                      (int)(ovoff | ((unsigned)((L - 1 - rw) >> 31) & ERF_OOB)),
                                                      (int)__builtin_amdgcn_readfirstlane(urow(a, r) * orowb), 0);
                if (to_lds) Ts[rw * NC + (mid_swz(rw, col >> 2) << 2) + (col & 3)] = v;
            }
    };

    // conv_b's first two weight tiles travel while the first epilogue runs
    constexpr int B_SLOTS2 = NC2 * CPR, B_PASSES2 = (B_SLOTS2 + NT - 1) / NT;
    constexpr bool B_FULL2 = B_SLOTS2 % NT == 0;
    f32x4 wb0[B_PASSES2], wb1[B_PASSES2];
    auto load_w2 = [&](int s, f32x4 *b) {
        const float *wt = d.wp2 + (long long)s * NC * KC + (NS > 1 ? (int)blockIdx.y * NC2 * KC : 0);
#pragma unroll
        for (int i = 0; i < B_PASSES2; ++i) {
            const int slot = tid + i * NT;
            b[i] = *reinterpret_cast<const f32x4 *>(wt + (B_FULL2 || slot < B_SLOTS2 ? slot : 0) * 4);
        }
    };
    auto store_b2 = [&](int buf, const f32x4 *b) {
#pragma unroll
        for (int i = 0; i < B_PASSES2; ++i) {
            const int slot = tid + i * NT;
            if (B_FULL2 || slot < B_SLOTS2)
                *reinterpret_cast<f32x4 *>(&Bs[buf][bs_off(slot / CPR, slot % CPR)]) = b[i];
        }
    };
    load_w2(0, wb0);
    load_w2(1, wb1);
    epilogue(d.epi1, sc1, sh1, (NS > 1 && blockIdx.y != 0) ? nullptr : d.mid, true,
             std::integral_constant<int, WM>{}, wm, col);
    store_b2(0, wb0);
    load_w2(2, wb0);
    clear_acc();
    __syncthreads();

    // ------------------------------------------------------------------ conv_b: A = mid tile in LDS
    // (weight tiles two steps ahead, as above: wb1 feeds the odd steps, wb0 the even ones)
    auto step2 = [&](int s, f32x4 *wreg) {
        const int buf = s & 1;
        if (s + 1 < NSTEPS) store_b2(buf ^ 1, wreg);
        if (s + 3 < NSTEPS) load_w2(s + 3, wreg);
        const int t = s / KCH, kc = s % KCH;
        const int sh = t == 0 ? s2a : (t == 1 ? s2b : s2c);
        const float *Bb = Bs[buf];
        const float *arow[WM2];
        int asw[WM2];
#pragma unroll
        for (int a = 0; a < WM2; ++a) {
            const int rr = (wm2 * WM2 + a) * 32 + l31 + sh;
            const int rz = (unsigned)rr < (unsigned)L ? rr : BM;       // beyond the line: the zero row
            arow[a] = Ts + rz * NC;
            asw[a] = (rz & 7) | ((((rz >> 3) ^ (rz >> 2)) & 1) << 3);
        }
#pragma unroll
        for (int qq = 0; qq < KC / 8; ++qq) {
            f32x4 af[WM2], bf;
#pragma unroll
            for (int a = 0; a < WM2; ++a)
                af[a] = *reinterpret_cast<const f32x4 *>(arow[a] + (((kc * CPR + 2 * qq + h) ^ asw[a]) << 2));
            bf = *reinterpret_cast<const f32x4 *>(&Bb[bs_off(colb, 2 * qq + h)]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int a = 0; a < WM2; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][m], bf[m], acc[a], 0, 0, 0);
        }
        lds_barrier();
    };
    for (int s = 0; s < NSTEPS; s += 2) {
        step2(s, wb1);
        step2(s + 1, wb0);
    }
    epilogue(d.epi2, sc2, sh2, d.out, false, std::integral_constant<int, WM2>{}, wm2, col2);
}
