// Implicit-GEMM convolution kernels on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
//
// Forward / data-gradient kernel (igemm_f32_kernel)
//   M = output pixels, N = output channels, K = taps x input channels, NHWC activations.
//   One K-step = one tap x 32 input channels.  The A tile (BM pixels x 32 channels, gathered at the
//   tap-shifted source pixels, zero outside the image) and the B tile (32 x BN weights, gathered
//   from torch's weight layout) are prefetched into registers while the previous step's MFMAs run,
//   then written to a single LDS buffer:
//     A: [BM][32] floats, 16-byte chunks XOR-swizzled with (row>>1)&7  -> conflict-free
//        ds_write_b128 (8 lanes cover one 128-B row) and ds_read_b128 (lanes = rows);
//     B: [BN][36] floats (k contiguous, row pitch 144 B)              -> conflict-free ds_read_b128.
//   A wave owns WM x WN blocks of 32x32 outputs.  A lane's ds_read_b128 delivers 4 consecutive k for
//   its row; MFMA number m of the group takes element m, so the 2-deep k of the instruction is
//   {8q+m, 8q+4+m} — A and B use the same permutation, the sum over k is unchanged.
//
// Weight-gradient kernel (wgrad_f32_kernel)
//   D[k][j] += sum_pixels X[pix][k] * G[pix][j] per tap; the MFMA's i index is the input channel,
//   its k index the pixel.  Pixels are split over blockIdx.x; each block writes its partial
//   [tap][k][j] tile to a slab, a second kernel (reduce_partials_kernel) sums the slabs in a fixed
//   order (bitwise reproducible) and scatters into torch's weight layout.
#pragma once
#include "common.h"

#define IG_KC 32          // channels per K-step
#define IG_BPITCH 36      // floats per B-tile row (k) : 144 B keeps 16-B alignment, spreads banks

struct RowInfo {          // per output pixel of the block tile
    int src_base;         // n * in_H * in_W   (pixel index), -1 = row beyond M
    int hs_ws;            // (gi*ss) | (gj*ss) << 16
    int out_pix;          // output pixel index (n*out_H + oh)*out_W + ow
    int n;
};

template <int WM, int WN, int GM, int GN>
__global__ __launch_bounds__(GM *GN * 64, 2) void igemm_f32_kernel(const IgemmDesc d)
{
    constexpr int BM = GM * WM * 32, BN = GN * WN * 32, NT = GM * GN * 64;
    constexpr int A_PASSES = BM / (NT / 8);      // float4 loads per thread per step
    constexpr int B_ELEMS = BN * IG_KC / NT;     // scalar weight loads per thread per step
    __shared__ __attribute__((aligned(16))) float As[BM * IG_KC];
    __shared__ __attribute__((aligned(16))) float Bs[BN * IG_BPITCH];
    __shared__ RowInfo rows[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / GN, wn = wave % GN;
    const int l31 = lane & 31, h = lane >> 5;
    const int grp = blockIdx.z;
    const long long M = (long long)d.N * d.GH * d.GW;
    const long long m0 = (long long)blockIdx.x * BM;
    const int j0 = blockIdx.y * BN;

    for (int r = tid; r < BM; r += NT) {
        const long long m = m0 + r;
        RowInfo ri;
        if (m < M) {
            const int n = (int)(m / (d.GH * d.GW));
            const int rem = (int)(m - (long long)n * d.GH * d.GW);
            const int gi = rem / d.GW, gj = rem - gi * d.GW;
            ri.src_base = n * d.in_H * d.in_W;
            ri.hs_ws = (gi * d.ss) | ((gj * d.ss) << 16);
            ri.out_pix = (n * d.out_H + gi * d.os + d.goa[grp]) * d.out_W + gj * d.os + d.gob[grp];
            ri.n = n;
        } else {
            ri.src_base = -1; ri.hs_ws = 0; ri.out_pix = -1; ri.n = 0;
        }
        rows[r] = ri;
    }
    __syncthreads();

    const int kchunks = (d.K + IG_KC - 1) / IG_KC;
    const int tap0 = d.gtap0[grp];
    const int nsteps = d.gntaps[grp] * kchunks;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    f32x4 areg[A_PASSES];
    float breg[B_ELEMS];
    const int ac = tid & 7;          // 16-byte chunk of the 128-B channel row
    const int ar = tid >> 3;         // first row handled by this thread
    // Weight gather: element i of this thread is (k, j) = (bk0 + i*bdk, bj0 + i*bdj); consecutive
    // lanes walk whichever of k / j has the smaller stride in torch's weight layout.
    static_assert(NT % IG_KC == 0 && NT % BN == 0, "weight gather tiling");
    int bk0, bj0, bdk, bdj;
    if (d.lane_along_j) { bj0 = tid % BN; bk0 = tid / BN; bdk = NT / BN; bdj = 0; }
    else { bk0 = tid % IG_KC; bj0 = tid / IG_KC; bdk = 0; bdj = NT / IG_KC; }
    const long long bstep = (long long)bdk * d.wsk + (long long)bdj * d.wsn;
    const long long bbase = (long long)bk0 * d.wsk + (long long)(j0 + bj0) * d.wsn;
    const int blds0 = bj0 * IG_BPITCH + bk0, bldsd = bdj * IG_BPITCH + bdk;

    auto load_step = [&](int s) {
        const int t = tap0 + s / kchunks;
        const int kc0 = (s % kchunks) * IG_KC;
        const int dh = d.tdh[t], dw = d.tdw[t];
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const RowInfo ri = rows[ar + i * (NT / 8)];
            const int ih = (ri.hs_ws & 0xffff) + dh, iw = (ri.hs_ws >> 16) + dw;
            const int kk = kc0 + ac * 4;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ri.src_base >= 0 && (unsigned)ih < (unsigned)d.in_H && (unsigned)iw < (unsigned)d.in_W &&
                kk < d.K) {
                const float *p = d.in + ((long long)(ri.src_base + ih * d.in_W + iw) * d.in_C + d.in_coff + kk);
                if (d.vec_ok) {
                    v = *reinterpret_cast<const f32x4 *>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kk + e < d.K) v[e] = p[e];
                }
            }
            areg[i] = v;
        }
        const float *wt = d.w + d.woff[t] + (long long)kc0 * d.wsk + bbase;
#pragma unroll
        for (int i = 0; i < B_ELEMS; ++i) {
            float v = 0.0f;
            if (kc0 + bk0 + i * bdk < d.K && j0 + bj0 + i * bdj < d.Nout) v = wt[i * bstep];
            breg[i] = v;
        }
    };

    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int r = ar + i * (NT / 8);
            *reinterpret_cast<f32x4 *>(&As[r * IG_KC + ((ac ^ ((r >> 1) & 7)) << 2)]) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < B_ELEMS; ++i) Bs[blds0 + i * bldsd] = breg[i];
    };

    load_step(0);
    for (int s = 0; s < nsteps; ++s) {
        store_step();
        __syncthreads();
        if (s + 1 < nsteps) load_step(s + 1);
        const int kc0 = (s % kchunks) * IG_KC;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (kc0 + q * 8 >= d.K) break;     // wave-uniform: short last chunk (K = 16)
            f32x4 af[WM], bf[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                const int row = (wm * WM + a) * 32 + l31;
                af[a] = *reinterpret_cast<const f32x4 *>(&As[row * IG_KC + (((2 * q + h) ^ ((row >> 1) & 7)) << 2)]);
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                const int col = (wn * WN + b) * 32 + l31;
                bf[b] = *reinterpret_cast<const f32x4 *>(&Bs[col * IG_BPITCH + q * 8 + h * 4]);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][m], bf[b][m], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
    for (int b = 0; b < WN; ++b) {
        const int c = j0 + (wn * WN + b) * 32 + l31;
        if (c >= d.Nout) continue;
        float sc, sh;
        epi_channel_affine(d.epi, c, sc, sh);
#pragma unroll
        for (int a = 0; a < WM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * WM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const RowInfo ri = rows[row];
                if (ri.out_pix < 0) continue;
                float v = acc[a][b][r] * sc + sh;
                v = epi_finish(d.epi, v, ri.out_pix, ri.n, c, d.Nout);
                d.out[(long long)ri.out_pix * d.out_C + d.out_coff + c] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient.  Block tile: IBLK = GI*IB*32 input channels x JBLK = GJ*JB*32 output channels of
// one tap; blockIdx.x = pixel slab, blockIdx.y = (tap, i-tile, j-tile).
// LDS: X tile [32 pixels][IBLK], G tile [32 pixels][JBLK]; MFMA operands are read with ds_read_b32
// (lanes walk the channel, the two lane halves take pixel 2s and 2s+1).
template <int IB, int JB, int GI, int GJ>
__global__ __launch_bounds__(GI *GJ * 64) void wgrad_f32_kernel(const WgradDesc d)
{
    constexpr int IBLK = GI * IB * 32, JBLK = GJ * JB * 32, NT = GI * GJ * 64;
    constexpr int PM = 32;
    constexpr int XV = PM * IBLK / 4, GV = PM * JBLK / 4;        // float4 slots per tile
    constexpr int XP = (XV + NT - 1) / NT, GP = (GV + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float Xs[PM * IBLK];
    __shared__ __attribute__((aligned(16))) float Gs[PM * JBLK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / GJ, wj = wave % GJ;
    const int l31 = lane & 31, h = lane >> 5;

    int y = blockIdx.y;
    const int jt = y % d.jtiles; y /= d.jtiles;
    const int it = y % d.itiles; y /= d.itiles;
    const int t = y;
    const int i0 = it * IBLK, j0 = jt * JBLK;
    const int dh = d.tdh[t], dw = d.tdw[t], oa = d.toa[t], ob = d.tob[t];

    const long long M = (long long)d.N * d.GH * d.GW;
    const long long mbeg = (long long)blockIdx.x * d.split_pixels;
    long long mend = mbeg + d.split_pixels;
    if (mend > M) mend = M;
    const int nsteps = mbeg < mend ? (int)((mend - mbeg + PM - 1) / PM) : 0;

    f32x16 acc[IB][JB];
#pragma unroll
    for (int a = 0; a < IB; ++a)
#pragma unroll
        for (int b = 0; b < JB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) bsum[b] = 0.0f;

    f32x4 xreg[XP], greg[GP];

    auto load_step = [&](int s) {
        const long long mb = mbeg + (long long)s * PM;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (slot < XV) {
                const int p = slot / (IBLK / 4), c = (slot % (IBLK / 4)) * 4;
                const long long m = mb + p;
                if (m < mend && i0 + c < d.K) {
                    const int n = (int)(m / (d.GH * d.GW));
                    const int rem = (int)(m - (long long)n * d.GH * d.GW);
                    const int gi = rem / d.GW, gj = rem - gi * d.GW;
                    const int ih = gi * d.ss + dh, iw = gj * d.ss + dw;
                    if ((unsigned)ih < (unsigned)d.in_H && (unsigned)iw < (unsigned)d.in_W) {
                        const float *px = d.in + (((long long)n * d.in_H + ih) * d.in_W + iw) * d.in_C + d.in_coff + i0 + c;
                        if (d.vec_ok) v = *reinterpret_cast<const f32x4 *>(px);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (i0 + c + e < d.K) v[e] = px[e];
                        }
                    }
                }
            }
            xreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (slot < GV) {
                const int p = slot / (JBLK / 4), c = (slot % (JBLK / 4)) * 4;
                const long long m = mb + p;
                if (m < mend && j0 + c < d.Nout) {
                    const int n = (int)(m / (d.GH * d.GW));
                    const int rem = (int)(m - (long long)n * d.GH * d.GW);
                    const int gi = rem / d.GW, gj = rem - gi * d.GW;
                    const float *pg = d.g + (((long long)n * d.g_H + gi * d.os + oa) * d.g_W + gj * d.os + ob) * d.g_C + d.g_coff + j0 + c;
                    if (d.vec_ok) v = *reinterpret_cast<const f32x4 *>(pg);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (j0 + c + e < d.Nout) v[e] = pg[e];
                    }
                }
            }
            greg[i] = v;
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            if (slot < XV) *reinterpret_cast<f32x4 *>(&Xs[slot * 4]) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            if (slot < GV) *reinterpret_cast<f32x4 *>(&Gs[slot * 4]) = greg[i];
        }
    };

    if (nsteps > 0) load_step(0);
    for (int s = 0; s < nsteps; ++s) {
        store_step();
        __syncthreads();
        if (s + 1 < nsteps) load_step(s + 1);
#pragma unroll 4
        for (int pp = 0; pp < PM / 2; ++pp) {
            const int p = 2 * pp + h;
            float af[IB], bf[JB];
#pragma unroll
            for (int a = 0; a < IB; ++a) af[a] = Xs[p * IBLK + (wi * IB + a) * 32 + l31];
#pragma unroll
            for (int b = 0; b < JB; ++b) {
                bf[b] = Gs[p * JBLK + (wj * JB + b) * 32 + l31];
                bsum[b] += bf[b];
            }
#pragma unroll
            for (int a = 0; a < IB; ++a)
#pragma unroll
                for (int b = 0; b < JB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // partial slab: [tap][k][j], then one bias row (column sums of G) per tap
    const long long slab = (long long)d.ntaps * d.K * d.Nout + (long long)d.ntaps * d.Nout;
    float *P = d.partial + (long long)blockIdx.x * slab;
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        const int j = j0 + (wj * JB + b) * 32 + l31;
        if (j >= d.Nout) continue;
#pragma unroll
        for (int a = 0; a < IB; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = i0 + (wi * IB + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (k < d.K) P[((long long)t * d.K + k) * d.Nout + j] = acc[a][b][r];
            }
        }
        // column sums of this tap's G pixels, taken once per tap (first i-tile, first wave row)
        if (it == 0 && wi == 0) {
            const float other = __shfl_xor(bsum[b], 32);
            if (h == 0) P[(long long)d.ntaps * d.K * d.Nout + (long long)t * d.Nout + j] = bsum[b] + other;
        }
    }
}

// dst[woff[t] + k*wsk + j*wsn] = sum_s partial[s][t][k][j];
// db[j] = sum_s sum_{t in bias_taps} partial[s][bias row t][j]   (fixed order: reproducible).
struct ReduceDesc {
    const float *partial;
    float *dw, *db;
    int nsplit, ntaps, K, Nout;
    int woff[ERF_MAX_TAPS], wsk, wsn;
    unsigned bias_taps;
};

__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceDesc d)
{
    const long long nw = (long long)d.ntaps * d.K * d.Nout;
    const long long slab = nw + (long long)d.ntaps * d.Nout;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nw + d.Nout) return;
    float s = 0.0f;
    if (idx < nw) {
        for (int i = 0; i < d.nsplit; ++i) s += d.partial[(long long)i * slab + idx];
        const int j = (int)(idx % d.Nout);
        const int k = (int)((idx / d.Nout) % d.K);
        const int t = (int)(idx / ((long long)d.Nout * d.K));
        d.dw[d.woff[t] + (long long)k * d.wsk + (long long)j * d.wsn] = s;
    } else if (d.db) {
        const int j = (int)(idx - nw);
        for (int t = 0; t < d.ntaps; ++t) {
            if (!((d.bias_taps >> t) & 1u)) continue;
            for (int i = 0; i < d.nsplit; ++i) s += d.partial[(long long)i * slab + nw + (long long)t * d.Nout + j];
        }
        d.db[j] = s;
    }
}
