// 16-channel 1-D convolutions (the full-resolution non_bottleneck_1d blocks of the decoder) as
// LDS-free streaming kernels on v_mfma_f32_16x16x4_f32.
//
// With 16 input and 16 output channels a 3-tap convolution carries 96 FLOP per 128 bytes of
// activation traffic: the layer is HBM-bound, a 32x32 MFMA tile would be half padding, and a
// shared-memory stage buys nothing because there is no reuse beyond the three taps (which the
// L1/L2 serve).  Operands therefore go from global memory straight into the MFMA's registers.
//
// MFMA 16x16x4 layouts (lane l: i = l & 15, g = l >> 4):
//   A[16][4]: lane holds A[i][g];  B[4][16]: lane holds B[g][i];  D[16][16]: reg r = D[4g + r][i].
#pragma once
#include "common.h"

// Forward / data gradient:  out[pix][j] = epi( sum_t sum_k in[src(pix,t)][k] * W[t][k][j] ).
// Computed transposed, D[j][pix] = sum_k Wt[j][k] * X[k][pix], so that a lane's four D registers
// are four consecutive channels of one pixel: loads, epilogue operands and stores are all one
// 16-byte access per lane and a wave touches 16 pixels x 64 B = 1 KiB contiguous per access.
// A lane loads channels 4g..4g+3 of pixel i; MFMA number m of a tap takes element m, i.e. the
// instruction's k index g stands for channel 4g + m, and the weight operand uses the same map.
// One wave = U tiles of 16 consecutive pixels (independent accumulators hide the MFMA latency).
template <typename T, int NTAPS, int U, int G = 1>
__global__ __launch_bounds__(256) void conv16_kernel(const IgemmDesc d)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const unsigned M = (unsigned)d.N * d.GH * d.GW, GHW = (unsigned)d.GH * d.GW;
    const unsigned base0 = ((unsigned)blockIdx.x * 4 + wave) * (16 * U * G);
    if (base0 >= M) return;
    const unsigned Pin = (unsigned)d.N * d.in_H * d.in_W, Pout = (unsigned)d.N * d.out_H * d.out_W;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, Pin * d.in_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, Pout * d.epi.res_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, Pout * d.epi.mask_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_ns = make_rsrc(d.epi.nscale, (unsigned)d.N * 64u);
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(d.epi.bias, 64u);
    const bool has_ns = d.epi.nscale != nullptr, has_mask = d.epi.mask != nullptr, has_res = d.epi.res != nullptr;

    float wr[NTAPS][4];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) wr[t][m] = d.w[d.woff[t] + (long long)(4 * g + m) * d.wsk + (long long)i * d.wsn];

    // Coordinates: one division pair per wave, on wave-uniform values; the tiles advance by 16
    // pixels with a uniform wrap, a lane adds its column (GW >= 16: at most one row wrap).
    unsigned tn = base0 / GHW;
    unsigned trem = base0 - tn * GHW;
    unsigned tgi = trem / (unsigned)d.GW, tgj = trem - tgi * d.GW;
    // channel affine of the lane's four output channels 4g..4g+3: v * sc + sh
    f32x4 sc = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
    f32x4 sh = buf_load4(r_bias, g * 16u);
    if (d.epi.bn.on) {          // wave-uniform
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(d.epi.bn.gamma + 4 * g);
        const f32x4 be = *reinterpret_cast<const f32x4 *>(d.epi.bn.beta + 4 * g);
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(d.epi.bn.mean + 4 * g);
        const f32x4 va = *reinterpret_cast<const f32x4 *>(d.epi.bn.var + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float inv = d.epi.bn.var_is_invstd ? va[r] : rsqrtf(va[r] + d.epi.bn.eps);
            sc[r] = ga[r] * inv;
            sh[r] = (sh[r] - mu[r]) * sc[r] + be[r];
        }
    }

    for (int grp = 0; grp < G; ++grp) {
    const unsigned base = base0 + grp * (16 * U);
    if (base >= M) break;
    f32x4 xv[U][NTAPS], rres[U], rmask[U], rns[U];
    int opix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned m = base + u * 16 + i;
        const bool valid = m < M;
        unsigned n = tn, gi = tgi, gj = tgj + i;
        if (gj >= (unsigned)d.GW) {
            gj -= d.GW;
            gi += 1;
            if (gi == (unsigned)d.GH) { gi = 0; n += 1; }
        }
        const unsigned px = (n * d.out_H + gi) * d.out_W + gj;
        opix[u] = valid ? (int)px : -1;
        const unsigned vbad = valid ? 0u : ERF_OOB;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int ih = (int)gi + d.tdh[t], iw = (int)gj + d.tdw[t];
            const unsigned bad = (unsigned)((ih | iw | (d.in_H - 1 - ih) | (d.in_W - 1 - iw)) >> 31) & ERF_OOB;
            const unsigned off = ((n * d.in_H + ih) * d.in_W + iw) * d.in_C + d.in_coff + 4 * g;
            xv[u][t] = Act<T>::load4(r_in, off | bad | vbad);
        }
        rres[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        rmask[u] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        rns[u] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        // absent tensors are skipped with wave-uniform branches (no registers held for them)
        if (has_res) rres[u] = Act<T>::load4(r_res, (px * d.epi.res_C + d.epi.res_coff + 4 * g) | vbad);
        if (has_mask) rmask[u] = Act<T>::load4(r_mask, (px * d.epi.mask_C + d.epi.mask_coff + 4 * g) | vbad);
        if (has_ns) rns[u] = buf_load4(r_ns, (n * 16 + 4 * g) * 4u);
        tgj += 16;                                   // uniform advance to the next tile
        while (tgj >= (unsigned)d.GW) {
            tgj -= d.GW;
            if (++tgi == (unsigned)d.GH) { tgi = 0; ++tn; }
        }
    }
    f32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][m], xv[u][t][m], acc[u], 0, 0, 0);

#pragma unroll
    for (int u = 0; u < U; ++u) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float e = acc[u][r] * sc[r] + sh[r];
            if (has_ns) e *= rns[u][r];
            e += rres[u][r];
            if (d.epi.relu) e = fmaxf(e, 0.0f);
            v[r] = rmask[u][r] > 0.0f ? e : 0.0f;
        }
        if (opix[u] >= 0)
            Act<T>::st4(d.out, (long long)((unsigned)opix[u] * d.out_C + d.out_coff + 4 * g), v);
    }
    }
}

// Weight gradient:  dW[t][k][j] = sum_pix X[src(pix,t)][k] * G[pix][j]  (stride 1, GW >= 16).
// MFMA roles: i index = input channel k, contraction = 4 consecutive pixels, column = output
// channel j.  A lane (i, g) reads one float of pixel 4u+g from X (per tap) and from G: a wave's
// access covers 4 pixels x 64 B.  Every wave keeps one accumulator per tap over its pixel range;
// the waves of a block are summed through LDS in a fixed order and written as one slab in the
// layout reduce_partials_kernel expects ([tap][k][j], then one bias row per tap).
template <typename T, int NTAPS, int NW>
__global__ __launch_bounds__(NW * 64) void wgrad16_kernel(const WgradDesc d)
{
    constexpr int SLOTS = NTAPS * 256 + 16;
    __shared__ float red[NW][SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const unsigned M = (unsigned)d.N * d.GH * d.GW;
    const unsigned H = (unsigned)d.GH, W = (unsigned)d.GW;
    const unsigned wsp = (unsigned)d.split_pixels / NW;              // multiple of 16 (host)
    const unsigned wbeg = blockIdx.x * (unsigned)d.split_pixels + wave * wsp;
    unsigned wend = wbeg + wsp;
    if (wend > M) wend = M;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, M * d.in_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, M * d.g_C * Act<T>::B);

    f32x4 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;

    // (row, column) of pixel m0 + g, advanced by 16 pixels per iteration
    unsigned q0 = (wbeg < M ? wbeg : 0u) + g;
    unsigned oh0 = (q0 / W) % H, ow0 = q0 % W;
    for (unsigned m0 = wbeg; m0 < wend; m0 += 16) {
        float xa[4][NTAPS], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned m = m0 + 4 * u + g;
            unsigned ow = ow0 + 4 * u, oh = oh0;
            if (ow >= W) { ow -= W; oh = oh + 1 == H ? 0u : oh + 1; }
            const bool valid = m < wend;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int ih = (int)oh + d.tdh[t], iw = (int)ow + d.tdw[t];
                const bool ok = valid && (unsigned)ih < H && (unsigned)iw < W;
                const unsigned off = (unsigned)((int)m + d.tdh[t] * (int)W + d.tdw[t]) * d.in_C + d.in_coff + i;
                xa[u][t] = Act<T>::load1(r_x, ok ? off : ERF_OOB);
            }
            gv[u] = Act<T>::load1(r_g, valid ? (m * d.g_C + d.g_coff + i) : ERF_OOB);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bsum += gv[u];
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][t], gv[u], acc[t], 0, 0, 0);
        }
        ow0 += 16;
        if (ow0 >= W) { ow0 -= W; oh0 = oh0 + 1 == H ? 0u : oh0 + 1; }
    }

#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][t * 256 + (4 * g + r) * 16 + i] = acc[t][r];
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (g == 0) red[wave][NTAPS * 256 + i] = bsum;
    __syncthreads();
    float *P = d.partial + (long long)blockIdx.x * (NTAPS * 256 + NTAPS * 16);
    for (int e = tid; e < SLOTS; e += NW * 64) {
        float s = red[0][e];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][e];
        if (e < NTAPS * 256) P[e] = s;
        else
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) P[NTAPS * 256 + t * 16 + (e - NTAPS * 256)] = s;
    }
}
