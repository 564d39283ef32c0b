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
#include <type_traits>
#include "common.h"

// The 16x16x3 weights of a 16-channel convolution, staged once per block: every lane needs 12-16 of
// them at addresses 12-192 bytes apart, and as global gathers (64 cache lines per instruction, per
// wave) those loads cost more than the convolution's own traffic.  When the tensor is the dense
// [16][16][3] block torch holds (checked from the strides), the block copies its 768 floats to LDS
// with coalesced loads and the lanes pick their values there.  Returns the pointer to index with
// woff[t] + k*wsk + j*wsn: the LDS copy, or the tensor itself.  All threads of the block must call it.
__device__ __forceinline__ const float *stage_w16(const IgemmDesc &c, float *lds768)
{
    int wmax = c.woff[0] > c.woff[1] ? c.woff[0] : c.woff[1];
    wmax = (c.woff[2] > wmax ? c.woff[2] : wmax) + 15 * (c.wsk + c.wsn);
    const bool dense = wmax < 768 && c.wsk > 0 && c.wsn > 0 && c.woff[0] >= 0 && c.woff[1] >= 0 && c.woff[2] >= 0;
    if (!dense) return c.w;                 // block-uniform
    for (int e = threadIdx.x; e < 768; e += blockDim.x) lds768[e] = c.w[e];
    __syncthreads();
    return lds768;
}

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
    __shared__ float wlds[768];
    const float *wsrc = stage_w16(d, wlds);
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
        for (int m = 0; m < 4; ++m) wr[t][m] = wsrc[d.woff[t] + (4 * g + m) * d.wsk + i * d.wsn];

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

// ------------------------------------------------------------------------------------------------
// Two chained 16-channel 1-D convolutions in one launch (the pairs of a full-resolution
// non_bottleneck_1d: conv1x3 -> conv3x1 and conv3x1 -> conv1x3, forward and data gradient).
//
//   mid = epi_a( conv_a(in) )      taps shift along one image axis
//   out = epi_b( conv_b(mid) )     taps shift along the other
//
// These layers are HBM-bound, so the point is traffic and launches: the intermediate is read back
// from LDS instead of memory (and is not written at all when the caller does not keep it).  A block
// owns a TH x TW tile of output pixels and first forms `mid` on the tile extended by conv_b's reach
// (`halo` pixels on both sides along conv_b's axis; <= 13 % more conv_a work, which the matrix pipe
// has to spare), zero outside the image — conv_b's padding — then runs conv_b out of LDS.  Both
// phases use conv16_kernel's transposed MFMA 16x16x4 formulation (a lane holds 4 consecutive channels
// of one pixel) and its epilogue, so the results equal two conv16_kernel launches bit for bit.
struct Pair16Desc {
    IgemmDesc a, b;         // a.out = b.in = mid (a.out is written only with keep_mid)
    int keep_mid;
    int b_along_w;          // conv_b's taps shift along W (else along H)
    int halo;               // largest |shift| of conv_b's taps (1..4)
    int th, tw;             // tile: 8 x 64 when conv_b runs along W, 16 x 32 when it runs along H
    int tiles_h, tiles_w;   // tiles per image
};
#define P16_MAX_EXT (24 * 32 > 8 * 72 ? 24 * 32 : 8 * 72)

template <typename T>
__global__ __launch_bounds__(256) void pair16_kernel(const Pair16Desc d)
{
    // intermediate tile in its storage type (what a separate launch would have read back)
    __shared__ __attribute__((aligned(16))) unsigned char mids[P16_MAX_EXT * 16 * sizeof(T)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const IgemmDesc &A = d.a, &B = d.b;
    const int H = A.GH, W = A.GW;
    const unsigned P = (unsigned)A.N * H * W;

    int bid = (int)xcd_tile(blockIdx.x, gridDim.x);
    const int tw_i = bid % d.tiles_w; bid /= d.tiles_w;
    const int th_i = bid % d.tiles_h;
    const int n = bid / d.tiles_h;
    const int h0 = th_i * d.th, w0 = tw_i * d.tw;
    const int hb = d.b_along_w ? 0 : d.halo, wb = d.b_along_w ? d.halo : 0;
    const int EH = d.th + 2 * hb, EW = d.tw + 2 * wb, NE = EH * EW;

    // one convolution's constants for this lane: weights (tap, k-group element m), channel affine
    struct Side {
        float wr[3][4];
        f32x4 sc, sh;
        __amdgpu_buffer_rsrc_t r_res, r_mask, r_ns;
        bool has_res, has_mask, has_ns;
    };
    __shared__ float wlds[2][768];
    auto side_init = [&](const IgemmDesc &c, Side &s, float *wl) __attribute__((always_inline)) {
        const float *wsrc = stage_w16(c, wl);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) s.wr[t][m] = wsrc[c.woff[t] + (4 * g + m) * c.wsk + i * c.wsn];
        const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(c.epi.bias, 64u);
        s.sc = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        s.sh = buf_load4(r_bias, g * 16u);
        if (c.epi.bn.on) {
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(c.epi.bn.gamma + 4 * g);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(c.epi.bn.beta + 4 * g);
            const f32x4 mu = *reinterpret_cast<const f32x4 *>(c.epi.bn.mean + 4 * g);
            const f32x4 va = *reinterpret_cast<const f32x4 *>(c.epi.bn.var + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float inv = c.epi.bn.var_is_invstd ? va[r] : rsqrtf(va[r] + c.epi.bn.eps);
                s.sc[r] = ga[r] * inv;
                s.sh[r] = (s.sh[r] - mu[r]) * s.sc[r] + be[r];
            }
        }
        s.r_res = make_rsrc(c.epi.res, P * c.epi.res_C * Act<T>::B);
        s.r_mask = make_rsrc(c.epi.mask, P * c.epi.mask_C * Act<T>::B);
        s.r_ns = make_rsrc(c.epi.nscale, (unsigned)c.N * 64u);
        s.has_res = c.epi.res != nullptr; s.has_mask = c.epi.mask != nullptr; s.has_ns = c.epi.nscale != nullptr;
    };
    // conv16_kernel's epilogue for the lane's 4 channels of pixel px (the optional operands were
    // loaded by the caller, ahead of the MFMAs)
    auto finish = [&](const IgemmDesc &c, const Side &s, f32x4 acc, f32x4 rres, f32x4 rmask, f32x4 rns) __attribute__((always_inline)) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float e = acc[r] * s.sc[r] + s.sh[r];
            if (s.has_ns) e *= rns[r];
            e += rres[r];
            if (c.epi.relu) e = fmaxf(e, 0.0f);
            v[r] = rmask[r] > 0.0f ? e : 0.0f;
        }
        return v;
    };
    auto load_opt = [&](const IgemmDesc &c, const Side &s, unsigned px, unsigned bad, f32x4 &rres, f32x4 &rmask, f32x4 &rns) __attribute__((always_inline)) {
        rres = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        rmask = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        rns = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        if (s.has_res) rres = Act<T>::load4(s.r_res, (px * c.epi.res_C + c.epi.res_coff + 4 * g) | bad);
        if (s.has_mask) rmask = Act<T>::load4(s.r_mask, (px * c.epi.mask_C + c.epi.mask_coff + 4 * g) | bad);
        if (s.has_ns) rns = buf_load4(s.r_ns, ((unsigned)n * 16 + 4 * g) * 4u);
    };

    Side SA, SB;            // both convolutions' constants up front: one exposed latency
    side_init(A, SA, wlds[0]);
    side_init(B, SB, wlds[1]);
    // ------------------------------------------------------------------ phase 1: mid on the extended tile
    {
        const Side &S = SA;
        const __amdgpu_buffer_rsrc_t r_in = make_rsrc(A.in, P * A.in_C * Act<T>::B);
        constexpr int U = 4;
        const int ngroups = (NE + 15) / 16;
        for (int g0 = wave * U; g0 < ngroups; g0 += 4 * U) {
            f32x4 xv[U][3], rres[U], rmask[U], rns[U];
            int eidx[U];
            unsigned opx[U];
            bool inimg[U], interior[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = (g0 + u) * 16 + i;
                const bool live = e < NE;
                const int r = e / EW, c = e - r * EW;
                const int ih = h0 - hb + r, iw = w0 - wb + c;
                inimg[u] = live && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                interior[u] = inimg[u] && r >= hb && r < hb + d.th && c >= wb && c < wb + d.tw;
                eidx[u] = live ? e : -1;
                const unsigned px = ((unsigned)n * H + ih) * W + iw;
                opx[u] = px;
                const unsigned vbad = inimg[u] ? 0u : ERF_OOB;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int sh_ = ih + A.tdh[t], sw_ = iw + A.tdw[t];
                    const unsigned bad = (unsigned)((sh_ | sw_ | (H - 1 - sh_) | (W - 1 - sw_)) >> 31) & ERF_OOB;
                    const unsigned off = (((unsigned)n * H + sh_) * W + sw_) * A.in_C + A.in_coff + 4 * g;
                    xv[u][t] = Act<T>::load4(r_in, off | bad | vbad);
                }
                load_opt(A, S, px, vbad, rres[u], rmask[u], rns[u]);
            }
            f32x4 acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(S.wr[t][m], xv[u][t][m], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (eidx[u] < 0) continue;
                f32x4 v = finish(A, S, acc[u], rres[u], rmask[u], rns[u]);
                if (!inimg[u]) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};       // conv_b's zero padding
                Act<T>::st4(mids, (long long)eidx[u] * 16 + 4 * g, v);
                if (d.keep_mid && interior[u]) Act<T>::st4(A.out, (long long)(opx[u] * A.out_C + A.out_coff + 4 * g), v);
            }
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase 2: conv_b out of LDS
    {
        const Side &S = SB;
        constexpr int U = 4;
        const int NO = d.th * d.tw, ngroups = NO / 16;
        int esh[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) esh[t] = B.tdh[t] * EW + B.tdw[t];       // tap shift in extended-tile positions
        for (int g0 = wave * U; g0 < ngroups; g0 += 4 * U) {
            f32x4 xv[U][3], rres[U], rmask[U], rns[U];
            unsigned opx[U];
            bool valid[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = (g0 + u) * 16 + i;
                const bool live = g0 + u < ngroups;
                const int r = o / d.tw, c = o - r * d.tw;
                const int oh = h0 + r, ow = w0 + c;
                valid[u] = live && oh < H && ow < W;
                const int e = live ? (r + hb) * EW + c + wb : (hb * EW + wb);
#pragma unroll
                for (int t = 0; t < 3; ++t) xv[u][t] = Act<T>::ld4(mids, (long long)(e + esh[t]) * 16 + 4 * g);
                const unsigned px = ((unsigned)n * H + oh) * W + ow;
                opx[u] = px;
                load_opt(B, S, px, valid[u] ? 0u : ERF_OOB, rres[u], rmask[u], rns[u]);
            }
            f32x4 acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(S.wr[t][m], xv[u][t][m], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!valid[u]) continue;
                const f32x4 v = finish(B, S, acc[u], rres[u], rmask[u], rns[u]);
                Act<T>::st4(B.out, (long long)(opx[u] * B.out_C + B.out_coff + 4 * g), v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv16_kernel for bf16 tensors on v_mfma_f32_16x16x32_bf16: at 16 channels the fp32-MFMA kernel
// above is bound by instruction issue, not by memory (48 MFMAs of 32 cycles and about 200 address /
// conversion instructions per 64 pixels: four plain copies of the tensor take a fifth of the time of
// four convolutions).  Here the bf16 data goes from memory into the MFMA operand as it is:
//   D[j][pixel] = sum_k Wt[j][k] X[k][pixel],  k = 32 = two taps x 16 input channels per instruction
// so a tile of 16 pixels takes two 16-byte loads per lane and two MFMAs of 16 cycles (taps 0+1, then
// tap 2 with the upper half of k zero) instead of three 8-byte loads, twelve unpackings and twelve
// MFMAs.  Lane (i, g) = (lane & 15, lane >> 4) supplies pixel i, k = 8g..8g+7: tap g >> 1, channels
// 8 (g & 1) .. +7; it receives channels 4g..4g+3 of pixel i, as in conv16_kernel, so the epilogue and
// the store are the same.  U independent tiles per batch, G batches per wave.
template <int U, int G>
__global__ __launch_bounds__(256) void conv16_bf16_kernel(const IgemmDesc d)
{
    typedef bf16_t T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const unsigned M = (unsigned)d.N * d.GH * d.GW, GHW = (unsigned)d.GH * d.GW;
    const unsigned base0 = ((unsigned)blockIdx.x * 4 + wave) * (16 * U * G);
    __shared__ float wlds[768];
    const float *wsrc = stage_w16(d, wlds);
    if (base0 >= M) return;
    const unsigned Pin = (unsigned)d.N * d.in_H * d.in_W, Pout = (unsigned)d.N * d.out_H * d.out_W;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, Pin * d.in_C * 2u);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, Pout * d.epi.res_C * 2u);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, Pout * d.epi.mask_C * 2u);
    const __amdgpu_buffer_rsrc_t r_ns = make_rsrc(d.epi.nscale, (unsigned)d.N * 64u);
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(d.epi.bias, 64u);
    const bool has_ns = d.epi.nscale != nullptr, has_mask = d.epi.mask != nullptr, has_res = d.epi.res != nullptr;

    // weight operands: lane (j = i, g) holds Wt[j][k = 8g..8g+7]; second instruction: tap 2, upper k zero
    bf16x8 wa, wb;
    {
        const int t = g >> 1, c0 = 8 * (g & 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wa[e] = (__bf16)wsrc[d.woff[t] + (c0 + e) * d.wsk + i * d.wsn];
            const float w2 = wsrc[d.woff[2] + (c0 + e) * d.wsk + i * d.wsn];
            wb[e] = (__bf16)(g < 2 ? w2 : 0.0f);
        }
    }
    // this lane's tap shifts: first load tap g >> 1, second load tap 2 (lanes g >= 2 load nothing)
    const int dh1 = d.tdh[g >> 1], dw1 = d.tdw[g >> 1], dh2 = d.tdh[2], dw2 = d.tdw[2];
    const unsigned c8 = (unsigned)d.in_coff + 8u * (g & 1);
    const unsigned upper = g >= 2 ? ERF_OOB : 0u;

    unsigned tn = base0 / GHW;
    unsigned trem = base0 - tn * GHW;
    unsigned tgi = trem / (unsigned)d.GW, tgj = trem - tgi * d.GW;
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
        u32x4 x1[U], x2[U];
        f32x4 rres[U], rmask[U], rns[U];
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
            {
                const int ih = (int)gi + dh1, iw = (int)gj + dw1;
                const unsigned bad = (unsigned)((ih | iw | (d.in_H - 1 - ih) | (d.in_W - 1 - iw)) >> 31) & ERF_OOB;
                const unsigned off = (((n * d.in_H + ih) * d.in_W + iw) * d.in_C + c8) * 2u;
                x1[u] = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)(off | bad | vbad), 0, 0);
            }
            {
                const int ih = (int)gi + dh2, iw = (int)gj + dw2;
                const unsigned bad = (unsigned)((ih | iw | (d.in_H - 1 - ih) | (d.in_W - 1 - iw)) >> 31) & ERF_OOB;
                const unsigned off = (((n * d.in_H + ih) * d.in_W + iw) * d.in_C + c8) * 2u;
                x2[u] = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)(off | bad | vbad | upper), 0, 0);
            }
            rres[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            rmask[u] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
            rns[u] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
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
        for (int u = 0; u < U; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x1[u]), f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, __builtin_bit_cast(bf16x8, x2[u]), acc[u], 0, 0, 0);
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

// ------------------------------------------------------------------------------------------------
// conv16_bf16_kernel for images whose width is a multiple of 16 (every 16-channel layer of the
// networks at even input sizes): a tile of 16 pixels then never leaves its image row, so image, row
// and first column of a tile are wave-uniform and live in scalar registers, and a lane's share of
// every address is a constant formed once per wave.  PMC showed the general kernel bound by vector
// instruction issue (about 106 VALU instructions per tile and lane, half of them coordinates, wrap
// tests and 64-bit store addresses; VALU busy on every SIMD for half the kernel's duration): here a
// tile costs one addition per operand load, one bounds test per tap load, and the epilogue.
//   AW: the taps shift along the row (a lane's column decides whether its tap is inside the image);
//   otherwise along the column (the lane's tap row decides).
// Residual / mask / dropout factors / output go through buffer accesses with the tile's byte offset
// as the scalar offset: absent tensors are skipped with wave-uniform branches, a tile beyond the end
// presents an out-of-range offset (loads return zero, the store is dropped).
template <int U, int G, bool AW>
__global__ __launch_bounds__(256) void conv16_bf16_rows_kernel(const IgemmDesc d)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const unsigned W = (unsigned)d.GW, H = (unsigned)d.GH;
    const unsigned M = (unsigned)d.N * H * W, GHW = H * W;
    const unsigned base0 = ((unsigned)blockIdx.x * 4 + wave) * (16 * U * G);
    __shared__ float wlds[768];
    const float *wsrc = stage_w16(d, wlds);
    if (base0 >= M) return;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, M * d.in_C * 2u);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(d.out, M * d.out_C * 2u);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, d.epi.res ? M * d.epi.res_C * 2u : 0u);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, d.epi.mask ? M * d.epi.mask_C * 2u : 0u);
    const __amdgpu_buffer_rsrc_t r_ns = make_rsrc(d.epi.nscale, d.epi.nscale ? (unsigned)d.N * 64u : 0u);
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(d.epi.bias, 64u);
    const bool has_ns = d.epi.nscale != nullptr, has_mask = d.epi.mask != nullptr, has_res = d.epi.res != nullptr;

    // weight operands: lane (j = i, g) holds Wt[j][k = 8g..8g+7]; second instruction: tap 2, upper k zero
    bf16x8 wa, wb;
    {
        const int t = g >> 1, c0 = 8 * (g & 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wa[e] = (__bf16)wsrc[d.woff[t] + (c0 + e) * d.wsk + i * d.wsn];
            const float w2 = wsrc[d.woff[2] + (c0 + e) * d.wsk + i * d.wsn];
            wb[e] = (__bf16)(g < 2 ? w2 : 0.0f);
        }
    }
    // a lane's constant part of every byte offset (relative to the tile's first pixel); the tap loads
    // may reach backwards: their sum with the tile's offset is formed per tile and is non-negative
    // whenever the tap lies inside the image
    const int dh1 = d.tdh[g >> 1], dw1 = d.tdw[g >> 1], dh2 = d.tdh[2], dw2 = d.tdw[2];
    const int c8 = d.in_coff + 8 * (g & 1);
    const int lane_in1 = ((dh1 * (int)W + dw1 + i) * d.in_C + c8) * 2;
    const int lane_in2 = ((dh2 * (int)W + dw2 + i) * d.in_C + c8) * 2;
    // the coordinate the bounds test looks at, relative to the tile: column (AW) or row; lanes g >= 2
    // load nothing with the second instruction: their coordinate always fails
    const int rel1 = AW ? i + dw1 : dh1;
    const int rel2 = g >= 2 ? 0x40000000 : (AW ? i + dw2 : dh2);
    const unsigned lane_res = (unsigned)((i * d.epi.res_C + d.epi.res_coff + 4 * g) * 2);
    const unsigned lane_mask = (unsigned)((i * d.epi.mask_C + d.epi.mask_coff + 4 * g) * 2);
    const unsigned lane_out = (unsigned)((i * d.out_C + d.out_coff + 4 * g) * 2);
    const unsigned res_px = (unsigned)d.epi.res_C * 2u, mask_px = (unsigned)d.epi.mask_C * 2u;
    const unsigned in_px = (unsigned)d.in_C * 2u, out_px = (unsigned)d.out_C * 2u;

    unsigned tn = base0 / GHW;
    unsigned trem = base0 - tn * GHW;
    unsigned tgi = trem / W, tgj = trem - tgi * W;
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
    const bool relu = d.epi.relu != 0;

    for (int grp = 0; grp < G; ++grp) {
        const unsigned base = base0 + grp * (16 * U);
        if (base >= M) break;
        u32x4 x1[U], x2[U];
        u32x2 qres[U], qmask[U];
        f32x4 rns[U];
        unsigned opix[U];           // pixel index of the tile's first pixel | ERF_OOB beyond the end
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // wave-uniform: the tile's first pixel, its row / column, and whether the tile exists
            const bool live = base + u * 16 < M;
            const unsigned px0 = (tn * H + tgi) * W + tgj;
            const unsigned lim = live ? (AW ? W : H) : 0u;          // a dead tile fails every bounds test
            const unsigned pos = AW ? tgj : tgi;
            const unsigned in0 = px0 * in_px;
            const unsigned o1 = (pos + (unsigned)rel1) < lim ? in0 + (unsigned)lane_in1 : ERF_OOB;
            const unsigned o2 = (pos + (unsigned)rel2) < lim ? in0 + (unsigned)lane_in2 : ERF_OOB;
            x1[u] = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)o1, 0, 0);
            x2[u] = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)o2, 0, 0);
            const unsigned dead = live ? 0u : ERF_OOB;
            opix[u] = px0 | dead;
            // absent tensors are skipped with wave-uniform branches, as in conv16_bf16_kernel
            qres[u] = u32x2{0u, 0u};
            qmask[u] = u32x2{0x3f803f80u, 0x3f803f80u};          // four bf16 ones
            rns[u] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
            if (has_res) qres[u] = __builtin_amdgcn_raw_buffer_load_b64(r_res, (int)(lane_res | dead), (int)(px0 * res_px), 0);
            if (has_mask) qmask[u] = __builtin_amdgcn_raw_buffer_load_b64(r_mask, (int)(lane_mask | dead), (int)(px0 * mask_px), 0);
            if (has_ns) rns[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_ns, (int)(16u * g), (int)(tn * 64u), 0));
            tgj += 16;                                   // uniform advance to the next tile
            if (tgj >= W) {
                tgj = 0;
                if (++tgi == H) { tgi = 0; ++tn; }
            }
        }
        // The products and the epilogue form one basic block per specialisation (no branch between the
        // last MFMA and the first read of its result).  This kernel stores 8 bytes per lane; the wrong
        // values once seen in pixels 12-15 of an fp32 sibling came from its 16-byte stores with a
        // register offset (buf_store4_soff), not from the accumulators.
        auto finish = [&](auto relu_t, auto ns_t) __attribute__((always_inline)) {
            constexpr bool RELU = decltype(relu_t)::value, NS = decltype(ns_t)::value;
            f32x4 acc[U];
            const f32x4 zero = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, x1[u]), zero, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, __builtin_bit_cast(bf16x8, x2[u]), acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 rres = unpack_bf16x4(qres[u]), rmask = unpack_bf16x4(qmask[u]);
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e = acc[u][r] * sc[r] + sh[r];
                    if (NS) e *= rns[u][r];
                    e += rres[r];
                    if (RELU) e = fmaxf(e, 0.0f);
                    v[r] = rmask[r] > 0.0f ? e : 0.0f;
                }
                const unsigned dead = opix[u] & ERF_OOB;
                __builtin_amdgcn_raw_buffer_store_b64(pack_bf16x4(v), r_out, (int)(lane_out | dead),
                                                      (int)((opix[u] & ~ERF_OOB) * out_px), 0);
            }
        };
        using T1 = std::true_type;
        using T0 = std::false_type;
        if (relu) {
            if (has_ns) finish(T1{}, T1{});
            else finish(T1{}, T0{});
        } else {
            if (has_ns) finish(T0{}, T1{});
            else finish(T0{}, T0{});
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv16_kernel<float> in the rows form of conv16_bf16_rows_kernel: images whose width is a multiple
// of 16, taps along one axis.  A tile's image, row and first column are wave-uniform scalars, a lane's
// share of every byte offset is a per-wave constant, residual / mask / output use the tile's byte
// offset as the buffer instructions' scalar offset.  The products run in conv16_kernel's order
// (tap, k-quarter, tile), so the results are bit-identical to it.  Operand vectors are converted from
// the loads' integer vectors as WHOLE vectors: a bit cast applied to one element of a vector reads
// element 0 with this compiler.
template <int U, int G, bool AW>
__global__ __launch_bounds__(256) void conv16_rows_kernel(const IgemmDesc d)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const unsigned W = (unsigned)d.GW, H = (unsigned)d.GH;
    const unsigned M = (unsigned)d.N * H * W, GHW = H * W;
    const unsigned base0 = ((unsigned)blockIdx.x * 4 + wave) * (16 * U * G);
    __shared__ float wlds[768];
    const float *wsrc = stage_w16(d, wlds);
    if (base0 >= M) return;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, M * d.in_C * 4u);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(d.out, M * d.out_C * 4u);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, d.epi.res ? M * d.epi.res_C * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, d.epi.mask ? M * d.epi.mask_C * 4u : 0u);
    const __amdgpu_buffer_rsrc_t r_ns = make_rsrc(d.epi.nscale, d.epi.nscale ? (unsigned)d.N * 64u : 0u);
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(d.epi.bias, 64u);
    const bool has_ns = d.epi.nscale != nullptr, has_mask = d.epi.mask != nullptr, has_res = d.epi.res != nullptr;

    float wr[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) wr[t][m] = wsrc[d.woff[t] + (4 * g + m) * d.wsk + i * d.wsn];

    // a lane's constant part of the byte offsets, relative to the tile's first pixel (the tap loads
    // may reach backwards: their sum with the tile's offset is non-negative whenever the tap lies
    // inside the image), and the coordinate the bounds test looks at: column (AW) or row
    int lane_in[3], rel[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        lane_in[t] = ((d.tdh[t] * (int)W + d.tdw[t] + i) * d.in_C + d.in_coff + 4 * g) * 4;
        rel[t] = AW ? i + d.tdw[t] : d.tdh[t];
    }
    const unsigned lane_res = (unsigned)((i * d.epi.res_C + d.epi.res_coff + 4 * g) * 4);
    const unsigned lane_mask = (unsigned)((i * d.epi.mask_C + d.epi.mask_coff + 4 * g) * 4);
    const unsigned lane_out = (unsigned)((i * d.out_C + d.out_coff + 4 * g) * 4);
    const unsigned res_px = (unsigned)d.epi.res_C * 4u, mask_px = (unsigned)d.epi.mask_C * 4u;
    const unsigned in_px = (unsigned)d.in_C * 4u, out_px = (unsigned)d.out_C * 4u;

    unsigned tn = base0 / GHW;
    unsigned trem = base0 - tn * GHW;
    unsigned tgi = trem / W, tgj = trem - tgi * W;
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
    const bool relu = d.epi.relu != 0;
    const f32x4 zero4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, one4 = f32x4{1.0f, 1.0f, 1.0f, 1.0f};

    for (int grp = 0; grp < G; ++grp) {
        const unsigned base = base0 + grp * (16 * U);
        if (base >= M) break;
        f32x4 xv[U][3], rres[U], rmask[U], rns[U];
        unsigned opix[U];           // pixel index of the tile's first pixel | ERF_OOB beyond the end
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // wave-uniform: the tile's first pixel, its row / column, and whether the tile exists
            const bool live = base + u * 16 < M;
            const unsigned px0 = (tn * H + tgi) * W + tgj;
            const unsigned lim = live ? (AW ? W : H) : 0u;          // a dead tile fails every bounds test
            const unsigned pos = AW ? tgj : tgi;
            const unsigned in0 = px0 * in_px;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const unsigned o = (pos + (unsigned)rel[t]) < lim ? in0 + (unsigned)lane_in[t] : ERF_OOB;
                xv[u][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)o, 0, 0));
            }
            const unsigned dead = live ? 0u : ERF_OOB;
            opix[u] = px0 | dead;
            rres[u] = zero4;
            rmask[u] = one4;
            rns[u] = one4;
            if (has_res) rres[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)(lane_res | dead), (int)(px0 * res_px), 0));
            if (has_mask) rmask[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_mask, (int)(lane_mask | dead), (int)(px0 * mask_px), 0));
            if (has_ns) rns[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_ns, (int)(16u * g), (int)(tn * 64u), 0));
            tgj += 16;                                   // uniform advance to the next tile
            if (tgj >= W) {
                tgj = 0;
                if (++tgi == H) { tgi = 0; ++tn; }
            }
        }
        f32x4 acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = zero4;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float xe = xv[u][t][m];
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][m], xe, acc[u], 0, 0, 0);
                }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = acc[u][r] * sc[r] + sh[r];
                if (has_ns) e *= rns[u][r];
                e += rres[u][r];
                if (relu) e = fmaxf(e, 0.0f);
                v[r] = rmask[u][r] > 0.0f ? e : 0.0f;
            }
            const unsigned dead = opix[u] & ERF_OOB;
            buf_store4_soff(__builtin_bit_cast(u32x4, v), r_out, (int)(lane_out | dead), (int)((opix[u] & ~ERF_OOB) * out_px));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad16_kernel for bf16 tensors on v_mfma_f32_16x16x32_bf16 (32 pixels per instruction).  The
// contraction index is the pixel while memory holds [pixel][channel]: a wave stages 32 pixels of G and
// of X shifted by each tap (zero outside the image: three 16-byte loads per lane instead of forty-eight
// 2-byte ones) in its own 4 KiB of LDS and takes the operand fragments out with the transposing read
// ds_read_b64_tr_b16 (lane (i, g): channel i, pixels 8g..8g+7).  Slab output as wgrad16_kernel.
__device__ __forceinline__ bf16x8 tr_frag16(const unsigned char *tile, int lane)
{
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned char *a = tile + (8 * g + q) * 32 + 8 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(a + 4 * 32));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void wgrad16_bf16_kernel(const WgradDesc d)
{
    constexpr int NTAPS = 3, SLOTS = NTAPS * 256 + 16;
    __shared__ float red[NW][SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned char stage[NW][4][32 * 32];     // G, X(tap 0..2): [32 px][16 ch] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int pl = lane >> 1, half = lane & 1;              // staging role: pixel of the chunk, channel half
    const unsigned M = (unsigned)d.N * d.GH * d.GW;
    const unsigned H = (unsigned)d.GH, W = (unsigned)d.GW;
    const unsigned wsp = (unsigned)d.split_pixels / NW;              // multiple of 32 (host)
    const unsigned wbeg = blockIdx.x * (unsigned)d.split_pixels + wave * wsp;
    unsigned wend = wbeg + wsp;
    if (wend > M) wend = M;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, M * d.in_C * 2u);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, M * d.g_C * 2u);

    f32x4 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;

    // (row, column) of this lane's staging pixel, advanced by 32 pixels per chunk
    unsigned q0 = (wbeg < M ? wbeg : 0u) + pl;
    unsigned oh = (q0 / W) % H, ow = q0 % W;
    u32x4 rg, rx[NTAPS];
    auto load_chunk = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned m = m0 + pl;
        const bool valid = m < wend;
        rg = __builtin_amdgcn_raw_buffer_load_b128(r_g, (int)(valid ? (m * d.g_C + d.g_coff + 8 * half) * 2u : ERF_OOB), 0, 0);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int ih = (int)oh + d.tdh[t], iw = (int)ow + d.tdw[t];
            const bool ok = valid && (unsigned)ih < H && (unsigned)iw < W;
            const unsigned off = ((unsigned)((int)m + d.tdh[t] * (int)W + d.tdw[t]) * d.in_C + d.in_coff + 8 * half) * 2u;
            rx[t] = __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)(ok ? off : ERF_OOB), 0, 0);
        }
        ow += 32;
        while (ow >= W) { ow -= W; oh = oh + 1 == H ? 0u : oh + 1; }
    };
    unsigned char *st = &stage[wave][0][0];
    if (wbeg < wend) load_chunk(wbeg);
    for (unsigned m0 = wbeg; m0 < wend; m0 += 32) {
        *reinterpret_cast<u32x4 *>(st + pl * 32 + half * 16) = rg;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) *reinterpret_cast<u32x4 *>(st + (1 + t) * 1024 + pl * 32 + half * 16) = rx[t];
        if (m0 + 32 < wend) load_chunk(m0 + 32);
        __builtin_amdgcn_wave_barrier();
        const bf16x8 gf = tr_frag16(st, lane);
        bf16x8 xf[NTAPS];
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) xf[t] = tr_frag16(st + (1 + t) * 1024, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum += (float)gf[e];
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[t], gf, acc[t], 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
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


// wgrad16_bf16_kernel for fp32 tensors with split operands (x = xh + xm + xl in bf16, six of the nine
// partial products, fp32 accumulation: fp32 accuracy, see line_x3.h): a lane loads 8 channels of its
// staging pixel as two float4 per tensor, splits them and writes three bf16 planes per staged tile
// (12 KiB of LDS per wave); 18 MFMAs per 32 pixels.
template <int NW>
__global__ __launch_bounds__(NW * 64) void wgrad16_x3_kernel(const WgradDesc d)
{
    constexpr int NTAPS = 3, SLOTS = NTAPS * 256 + 16;
    __shared__ float red[NW][SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned char stage[NW][4][3][32 * 32];     // G, X(tap 0..2) x planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int pl = lane >> 1, half = lane & 1;              // staging role: pixel of the chunk, channel half
    const unsigned M = (unsigned)d.N * d.GH * d.GW;
    const unsigned H = (unsigned)d.GH, W = (unsigned)d.GW;
    const unsigned wsp = (unsigned)d.split_pixels / NW;              // multiple of 32 (host)
    const unsigned wbeg = blockIdx.x * (unsigned)d.split_pixels + wave * wsp;
    unsigned wend = wbeg + wsp;
    if (wend > M) wend = M;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, M * d.in_C * 4u);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, M * d.g_C * 4u);

    f32x4 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;

    unsigned q0 = (wbeg < M ? wbeg : 0u) + pl;
    unsigned oh = (q0 / W) % H, ow = q0 % W;
    f32x4 rg[2], rx[NTAPS][2];
    auto load_chunk = [&](unsigned m0) __attribute__((always_inline)) {
        const unsigned m = m0 + pl;
        const bool valid = m < wend;
        const unsigned og = valid ? (m * d.g_C + d.g_coff + 8 * half) * 4u : ERF_OOB;
        rg[0] = buf_load4(r_g, og);
        rg[1] = buf_load4(r_g, og + 16u);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int ih = (int)oh + d.tdh[t], iw = (int)ow + d.tdw[t];
            const bool ok = valid && (unsigned)ih < H && (unsigned)iw < W;
            const unsigned off = ok ? ((unsigned)((int)m + d.tdh[t] * (int)W + d.tdw[t]) * d.in_C + d.in_coff + 8 * half) * 4u : ERF_OOB;
            rx[t][0] = buf_load4(r_x, off);
            rx[t][1] = buf_load4(r_x, off + 16u);
        }
        ow += 32;
        while (ow >= W) { ow -= W; oh = oh + 1 == H ? 0u : oh + 1; }
    };
    // 8 values -> 16 bytes in each of the three planes of a staged tile
    auto put = [&](unsigned char *tile, const f32x4 (&v)[2]) __attribute__((always_inline)) {
        u32x4 hi, mid, lo;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32x2 h2 = pack_bf16x4(v[k]);
            const f32x4 r1 = v[k] - unpack_bf16x4(h2);
            const u32x2 m2 = pack_bf16x4(r1);
            const u32x2 l2 = pack_bf16x4(r1 - unpack_bf16x4(m2));
            hi[2 * k] = h2[0]; hi[2 * k + 1] = h2[1];
            mid[2 * k] = m2[0]; mid[2 * k + 1] = m2[1];
            lo[2 * k] = l2[0]; lo[2 * k + 1] = l2[1];
        }
        unsigned char *dst = tile + pl * 32 + half * 16;
        *reinterpret_cast<u32x4 *>(dst) = hi;
        *reinterpret_cast<u32x4 *>(dst + 1024) = mid;
        *reinterpret_cast<u32x4 *>(dst + 2048) = lo;
    };
    unsigned char *st = &stage[wave][0][0][0];
    if (wbeg < wend) load_chunk(wbeg);
    for (unsigned m0 = wbeg; m0 < wend; m0 += 32) {
        put(st, rg);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) put(st + (1 + t) * 3072, rx[t]);
        if (m0 + 32 < wend) load_chunk(m0 + 32);
        __builtin_amdgcn_wave_barrier();
        bf16x8 gf[3], xf[NTAPS][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            gf[p] = tr_frag16(st + p * 1024, lane);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) xf[t][p] = tr_frag16(st + (1 + t) * 3072 + p * 1024, lane);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum += ((float)gf[2][e] + (float)gf[1][e]) + (float)gf[0][e];
#pragma unroll
        for (int term = 0; term < 6; ++term) {
            // (X plane, G plane), small terms first: lh hl mm mh hm hh
            const int px = term == 0 ? 2 : (term == 1 || term == 4 || term == 5 ? 0 : 1);
            const int pg = term == 1 ? 2 : (term == 0 || term == 3 || term == 5 ? 0 : 1);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[t][px], gf[pg], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
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
