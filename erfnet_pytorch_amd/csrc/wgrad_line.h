// Weight gradient of the 3-tap 1-D convolutions of a non_bottleneck_1d, all three taps per block
// (gfx950, fp32 MFMA 32x32x2, contraction index = pixel).
//
//   dW[t][k][j] = sum_p X[p + (t-1)*dil along the conv axis][k] * G[p][j]
//
// wgrad_f32_kernel gives every tap a block of its own, so X and G travel from memory three times.
// Here a block walks segments of 64 consecutive positions of an image line (a column for a
// convolution along H, a row for one along W): it stages the G rows of the segment and the X rows
// of the segment extended by `dil` positions on both sides (positions outside the line are zero),
// and feeds the three taps from the same LDS tile by shifting the row index.  The G fragment of a
// pixel pair is read once and used by all three taps.  Partial sums go to the slab format of
// wgrad_f32_kernel ([3][K][N] + bias rows, summed in a fixed order by reduce_partials_kernel).
#pragma once
#include "common.h"

struct WgradLineDesc {
    const void *in, *g;     // storage type T
    float *partial;
    int N, img_pixels, Len, lines_per_img;
    int pstride;            // pixel-index distance between neighbours on a line
    int qstride;            // pixel-index distance between the bases of neighbouring lines
    int dil;                // 1..16
    int segs_per_line, nseg, spb;       // segments per line / in total / per block
};
struct WgradLineBatch {
    WgradLineDesc d[ERF_WGRAD_BATCH];
};

// NC channels on both sides; block tile = all NC input channels x JBLK = GJ*JB*32 output channels,
// GI*IB*32 = NC, 4 or 8 waves.  grid = (slabs, NC / JBLK, convolutions).
template <int NC, int IB, int JB, int GI, int GJ, int MINB, typename T = float>
__global__ __launch_bounds__(GI *GJ * 64, MINB) void wgrad_line_kernel(const WgradLineBatch batch)
{
    const WgradLineDesc &d = batch.d[blockIdx.z];
    constexpr int L = 64, XR = 96, NT = GI * GJ * 64;
    constexpr int IBLK = GI * IB * 32, JBLK = GJ * JB * 32;
    static_assert(IBLK == NC && (GI * GJ == 4 || GI * GJ == 8), "tile shape");
    constexpr int XQ = IBLK / 4, GQ = JBLK / 4;             // float4 slots per row
    constexpr int XP = XR * XQ / NT, GP = L * GQ / NT;
    static_assert(XR * XQ % NT == 0 && L * GQ % NT == 0, "whole passes");
    __shared__ __attribute__((aligned(16))) float Xs[XR * IBLK];
    __shared__ __attribute__((aligned(16))) float Gs[L * JBLK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / GJ, wj = wave % GJ;
    const int l31 = lane & 31, h = lane >> 5;
    const int j0 = blockIdx.y * JBLK;
    const int dil = d.dil;
    const int s0 = blockIdx.x * d.spb;
    const int s1 = s0 + d.spb < d.nseg ? s0 + d.spb : d.nseg;

    f32x16 acc[3][IB][JB];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < IB; ++a)
#pragma unroll
            for (int b = 0; b < JB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.0f;
    float bsum[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) bsum[b] = 0.0f;

    const unsigned P = (unsigned)d.N * d.img_pixels;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, P * NC * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, P * NC * Act<T>::B);
    f32x4 xreg[XP], greg[GP];

    auto load_seg = [&](int s) {
        const int line = s / d.segs_per_line;
        const int q0 = (s - line * d.segs_per_line) * L;
        const int n = line / d.lines_per_img;
        const int ql = line - n * d.lines_per_img;
        const unsigned base = (unsigned)n * d.img_pixels + (unsigned)ql * d.qstride;
        const int xrows = L + 2 * dil;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / XQ, c = (slot % XQ) * 4;
            const int q = q0 - dil + r;
            unsigned off = (base + (unsigned)(q * d.pstride)) * NC + c;
            off |= (unsigned)(((xrows - 1 - r) | q | (d.Len - 1 - q)) >> 31) & ERF_OOB;
            xreg[i] = Act<T>::load4(r_x, off);
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / GQ, c = (slot % GQ) * 4;
            const int q = q0 + r;
            unsigned off = (base + (unsigned)(q * d.pstride)) * NC + j0 + c;
            off |= (unsigned)((d.Len - 1 - q) >> 31) & ERF_OOB;
            greg[i] = Act<T>::load4(r_g, off);
        }
    };
    auto store_seg = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i) *reinterpret_cast<f32x4 *>(&Xs[(tid + i * NT) * 4]) = xreg[i];
#pragma unroll
        for (int i = 0; i < GP; ++i) *reinterpret_cast<f32x4 *>(&Gs[(tid + i * NT) * 4]) = greg[i];
    };

    if (s0 < s1) load_seg(s0);
    for (int s = s0; s < s1; ++s) {
        store_seg();
        __syncthreads();
        if (s + 1 < s1) load_seg(s + 1);
        // pixel pair pp = positions 2pp, 2pp+1 of the segment (the two lane halves); the tile row of
        // tap t is position + t*dil (row 0 of the X tile is position -dil)
        const float *xb = Xs + h * IBLK + wi * IB * 32 + l31;
        const float *gb = Gs + h * JBLK + wj * JB * 32 + l31;
        const int tstep = dil * IBLK;
        float af[2][3][IB], bf[2][JB];
        auto read_pair = [&](int pp, int set) {
#pragma unroll
            for (int b = 0; b < JB; ++b) bf[set][b] = gb[2 * pp * JBLK + b * 32];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < IB; ++a) af[set][t][a] = xb[2 * pp * IBLK + t * tstep + a * 32];
        };
        read_pair(0, 0);
#pragma unroll
        for (int pp = 0; pp < L / 2; ++pp) {
            const int cur = pp & 1;
            if (pp + 1 < L / 2) read_pair(pp + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of this pair's MFMAs
#pragma unroll
            for (int b = 0; b < JB; ++b) bsum[b] += bf[cur][b];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < IB; ++a)
#pragma unroll
                    for (int b = 0; b < JB; ++b)
                        acc[t][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][t][a], bf[cur][b], acc[t][a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // partial slab [3][K][N], then the bias rows ([3][N]; row 0 carries the column sums of G)
    const long long slab = 3LL * NC * NC + 3 * NC;
    float *Pp = d.partial + (long long)blockIdx.x * slab;
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        const int j = j0 + (wj * JB + b) * 32 + l31;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int a = 0; a < IB; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = (wi * IB + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    Pp[((long long)t * NC + k) * NC + j] = acc[t][a][b][r];
                }
        if (wi == 0) {
            const float other = __shfl_xor(bsum[b], 32);
            if (h == 0) Pp[3LL * NC * NC + j] = bsum[b] + other;
        }
    }
}
