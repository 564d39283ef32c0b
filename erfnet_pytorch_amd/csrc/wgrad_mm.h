// Generic weight gradient (any taps / strides / output-parity offsets, as wgrad_f32_kernel) of bf16
// tensors on the bf16 matrix pipe (gfx950, v_mfma_f32_32x32x16_bf16, contraction index = pixel, fp32
// accumulation and output).  (The split-operand form for fp32 tensors was measured slower than
// wgrad_f32_kernel here — twice the LDS fragment traffic per MFMA at these tile shapes — and is not
// built.)
// Same block decomposition, pixel table and slab format as wgrad_f32_kernel (one (tap, k-tile,
// j-tile) and a slab of pixels per block; 32 pixels per step = two 16-pixel k-steps).  The tensors are
// [pixel][channel], both operands need the pixel as contraction index: fragments come out of LDS
// through the transposing read (tr_frag, wgrad_line_bf16.h); tile rows are 64 bytes longer than the
// data, which spreads the rows a transposed read touches over different banks.
#pragma once
#include "igemm.h"
#include "wgrad_line_bf16.h"

template <int IB, int JB, int GI, int GJ>
__global__ __launch_bounds__(GI *GJ * 64) void wgrad_mm_kernel(const WgradBatch batch)
{
    const WgradDesc &d = batch.d[blockIdx.z];      // up to four convolutions per launch
    constexpr int PM = 32;
    constexpr int IBLK = GI * IB * 32, JBLK = GJ * JB * 32, NT = GI * GJ * 64;
    static_assert(PM <= NT, "one thread per pixel of a step keeps the pixel table");
    typedef bf16_t T;
    constexpr int PLANES = 1;
    constexpr int XV = PM * IBLK / 4, GV = PM * JBLK / 4;        // 4-channel slots per tile
    constexpr int XP = (XV + NT - 1) / NT, GP = (GV + NT - 1) / NT;
    constexpr int XPB = IBLK * 2 + 64, GPB = JBLK * 2 + 64;      // bytes per tile row (one plane)
    constexpr int XSZ = PM * XPB, GSZ = PM * GPB;
    __shared__ __attribute__((aligned(16))) unsigned char Xs[PLANES * XSZ];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[PLANES * GSZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / GJ, wj = wave % GJ;
    const int l31 = lane & 31, h = lane >> 5;

    int y = blockIdx.y;
    const int jt = y % d.jtiles; y /= d.jtiles;
    const int it = y % d.itiles; y /= d.itiles;
    const int t = y;
    const int i0 = it * IBLK, j0 = jt * JBLK;
    const int dh = d.tdh[t], dw = d.tdw[t], oa = d.toa[t], ob = d.tob[t];

    const unsigned M = (unsigned)d.N * d.GH * d.GW, GHW = (unsigned)d.GH * d.GW;
    const unsigned mbeg = blockIdx.x * (unsigned)d.split_pixels;
    unsigned mend = mbeg + d.split_pixels;
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

    // pixel table of the step being loaded (see wgrad_f32_kernel)
    __shared__ int4 ptab[2][PM];
    unsigned tn = 0, tgi = 0, tgj = 0;
    if (tid < PM) {
        const unsigned m = mbeg + tid;
        const unsigned mm = m < M ? m : 0;
        tn = mm / GHW;
        const unsigned rem = mm - tn * GHW;
        tgi = rem / (unsigned)d.GW;
        tgj = rem - tgi * d.GW;
    }
    auto table_step = [&](int s) __attribute__((always_inline)) {
        if (tid < PM) {
            const unsigned m = mbeg + (unsigned)s * PM + tid;
            int4 e;
            e.x = m < mend ? (int)(tn * d.in_H * d.in_W) : -1;
            e.y = (int)tgi * d.ss;
            e.z = (int)tgj * d.ss;
            e.w = (int)((tn * d.g_H + tgi * d.os + oa) * d.g_W + tgj * d.os + ob);
            ptab[s & 1][tid] = e;
            tgj += PM;
            while (tgj >= (unsigned)d.GW) {
                tgj -= d.GW;
                if (++tgi == (unsigned)d.GH) { tgi = 0; ++tn; }
            }
        }
    };

    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, (unsigned)d.N * d.in_H * d.in_W * d.in_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, (unsigned)d.N * d.g_H * d.g_W * d.g_C * Act<T>::B);
    auto load_step = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            const int p = slot / (IBLK / 4), c = (slot % (IBLK / 4)) * 4;
            const int4 e = ptab[s & 1][p < PM ? p : 0];
            const int ih = e.y + dh, iw = e.z + dw;
            const bool ok = slot < XV && e.x >= 0 && i0 + c < d.K && (unsigned)ih < (unsigned)d.in_H &&
                            (unsigned)iw < (unsigned)d.in_W;
            unsigned off = (unsigned)(e.x + ih * d.in_W + iw) * d.in_C + d.in_coff + i0 + c;
            off |= ok ? 0u : ERF_OOB;
            xreg[i] = Act<T>::load4(r_x, off);
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            const int p = slot / (JBLK / 4), c = (slot % (JBLK / 4)) * 4;
            const int4 e = ptab[s & 1][p < PM ? p : 0];
            const bool ok = slot < GV && e.x >= 0 && j0 + c < d.Nout;
            unsigned off = (unsigned)e.w * d.g_C + d.g_coff + j0 + c;
            off |= ok ? 0u : ERF_OOB;
            greg[i] = Act<T>::load4(r_g, off);
        }
    };
    // 4 channels -> 8 bytes per plane
    auto put = [&](unsigned char *dst, int plane_bytes, f32x4 v) __attribute__((always_inline)) {
        const u32x2 hi = pack_bf16x4(v);
        *reinterpret_cast<u32x2 *>(dst) = hi;        // (exact: the values were bf16 in memory)
    };
    auto store_step = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            if (slot < XV) put(Xs + (slot / (IBLK / 4)) * XPB + (slot % (IBLK / 4)) * 8, XSZ, xreg[i]);
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            if (slot < GV) put(Gs + (slot / (JBLK / 4)) * GPB + (slot % (JBLK / 4)) * 8, GSZ, greg[i]);
        }
    };

    if (nsteps > 0) {
        table_step(0);
        __syncthreads();
        load_step(0);
    }
    for (int s = 0; s < nsteps; ++s) {
        store_step();
        if (s + 1 < nsteps) table_step(s + 1);
        __syncthreads();
        if (s + 1 < nsteps) load_step(s + 1);
#pragma unroll
        for (int ks = 0; ks < PM / 16; ++ks) {
            bf16x8 xf[PLANES][IB], gf[PLANES][JB];
#pragma unroll
            for (int p = 0; p < PLANES; ++p) {
#pragma unroll
                for (int a = 0; a < IB; ++a) xf[p][a] = tr_frag(Xs + p * XSZ, XPB, 16 * ks, (wi * IB + a) * 32, lane);
#pragma unroll
                for (int b = 0; b < JB; ++b) gf[p][b] = tr_frag(Gs + p * GSZ, GPB, 16 * ks, (wj * JB + b) * 32, lane);
            }
            if (it == 0 && wi == 0) {       // column sums of G (the bias gradient)
#pragma unroll
                for (int b = 0; b < JB; ++b)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bsum[b] += (float)gf[0][b][e];
                    }
            }
#pragma unroll
            for (int a = 0; a < IB; ++a)
#pragma unroll
                for (int b = 0; b < JB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[0][a], gf[0][b], acc[a][b], 0, 0, 0);
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
        if (it == 0 && wi == 0) {
            const float other = __shfl_xor(bsum[b], 32);
            if (h == 0) P[(long long)d.ntaps * d.K * d.Nout + (long long)t * d.Nout + j] = bsum[b] + other;
        }
    }
}
