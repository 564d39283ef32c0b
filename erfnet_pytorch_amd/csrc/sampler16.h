// The stride-2 3x3 layers between the 16-channel full-resolution stage and the 64-channel stage
// (UpsamplerBlock 64->16 forward, DownsamplerBlock 16->64 data gradient, and the weight gradients
// of both) as streaming kernels on v_mfma_f32_16x16x4_f32.
//
// All four are the same pairing of a coarse grid (H x W, K = 16*KQ channels) with a fine grid
// (2H x 2W, 16 channels): coarse pixel (i+dh, j+dw) meets fine pixel (2i+a, 2j+b) through tap
// (kh, kw), where per axis (a, kh, dh) is one of (0,1,0), (1,0,1), (1,2,0).  Nine taps, four fine
// "classes" (a,b), four coarse "sources" (dh,dw) per coarse position.  With 16 channels on the
// fine side the generic 32x32 MFMA tile would be half padding and every output pixel would be
// written as a 64-byte piece by a different block; here one wave owns 16 (forward) or 4 (weight
// gradient) coarse positions with all their classes.
//
// MFMA 16x16x4 layouts (lane l: i = l & 15, g = l >> 4):
//   A[16][4]: lane holds A[i][g];  B[4][16]: lane holds B[g][i];  D[16][16]: reg r = D[4g + r][i].
#pragma once
#include "common.h"

#define S16_TAP_KH(t) ((t) / 3)
#define S16_TAP_KW(t) ((t) % 3)
#define S16_CLASS(t) (((S16_TAP_KH(t) == 1 ? 0 : 1) << 1) | (S16_TAP_KW(t) == 1 ? 0 : 1))
#define S16_SRC(t) (((S16_TAP_KH(t) == 0 ? 1 : 0) << 1) | (S16_TAP_KW(t) == 0 ? 1 : 0))

// fine[2i+a][2j+b][c] = epi( sum_taps sum_k coarse[i+dh][j+dw][k] * W[tap][k][c] ),  c < 16.
// Computed transposed (D[c][pos] = Wt * X) so that a lane ends up with four consecutive channels
// of one fine pixel per class: 16-byte loads and stores throughout.  The weights of all nine taps
// sit in LDS as [tap][k/16][c][16 k]: a wave's ds_read_b128 covers 1 KiB contiguous (no conflicts).
// A lane loads channels 16q+4g..+3 of its coarse source pixels; MFMA number m takes element m, so
// the instruction's k index g stands for channel 16q + 4g + m on both operands.
// Gradient of a 2x2 max-pool for a lane's four channels: xp[c] = the pooled tensor at the window's
// four pixels (c = 2*row + column), gp = the gradient at the window; the first largest value takes it
// (pool_bwd_kernel's order of comparisons).
__device__ __forceinline__ void s16_pool_grad(const f32x4 (&xp)[4], f32x4 gp, f32x4 (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int best = 0;
        float mx = xp[0][r];
        if (xp[1][r] > mx) { mx = xp[1][r]; best = 1; }
        if (xp[2][r] > mx) { mx = xp[2][r]; best = 2; }
        if (xp[3][r] > mx) { mx = xp[3][r]; best = 3; }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c][r] = best == c ? gp[r] : 0.0f;
    }
}

template <typename T, int KQ, bool POOL = false>
__global__ __launch_bounds__(256) void up16_fwd_kernel(const IgemmDesc d)
{
    __shared__ __attribute__((aligned(16))) float Wl[9 * KQ * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    if (d.wsn == 9 && d.wsk == 144 && ((unsigned long long)d.w & 15) == 0) {
        // torch's dense [K][16][3][3] weight: read in memory order (16-byte loads, a wave covers 1 KiB
        // contiguous) and scattered into the LDS image — the gather below costs a block 36 loads per
        // thread with one element per cache-line piece
        for (int v4 = tid; v4 < 9 * KQ * 64; v4 += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(d.w + 4 * v4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * v4 + c, k = e / 144, rem = e - k * 144, j = rem / 9, t = rem - j * 9;
                Wl[((t * KQ + (k >> 4)) * 16 + j) * 16 + (k & 15)] = v[c];
            }
        }
    } else {
        for (int idx = tid; idx < 9 * KQ * 256; idx += 256) {
            const int kk = idx & 15, j = (idx >> 4) & 15, q = (idx >> 8) % KQ, t = idx / (KQ * 256);
            Wl[idx] = d.w[t + (long long)(16 * q + kk) * d.wsk + (long long)j * d.wsn];
        }
    }
    __syncthreads();

    const unsigned H = (unsigned)d.GH, W = (unsigned)d.GW, M = (unsigned)d.N * H * W, HW = H * W;
    const unsigned Pout = M * 4u;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, M * d.in_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(d.out, Pout * d.out_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, Pout * d.epi.res_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, Pout * d.epi.mask_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(d.epi.bias, 64u);
    const bool has_mask = d.epi.mask != nullptr, relu = d.epi.relu != 0;
    constexpr bool pooled = POOL;       // (the host instantiates POOL for a launch with d.pool_x)
    const bool px_is_mask = d.pool_x == d.epi.mask;     // (wave-uniform)
    const __amdgpu_buffer_rsrc_t r_px = make_rsrc(d.pool_x, Pout * d.out_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_pg = make_rsrc(d.pool_g, M * d.pool_gC * Act<T>::B);

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

    // A wave walks its tiles in steps of 16 coarse channels (q): the four source loads of the NEXT step
    // (the next tile's first one behind a tile's last) are requested before the 36 MFMAs of the current
    // one, and the residual / mask of a tile before its last step — one round trip per tile and three or
    // four waves per SIMD left the matrix pipe idle two thirds of the time.
    struct Geo { unsigned n, y, x; bool valid; };
    const unsigned ntiles = (M + 15) / 16;
    auto geo_of = [&](unsigned tile) __attribute__((always_inline)) {
        Geo t;
        const unsigned p = tile * 16 + i;
        t.valid = tile < ntiles && p < M;
        const unsigned pp = t.valid ? p : 0;
        t.n = pp / HW;
        const unsigned rem = pp - t.n * HW;
        t.y = rem / W; t.x = rem - t.y * W;
        return t;
    };
    auto src_offs = [&](const Geo &t, unsigned (&off)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned ys = t.y + (s >> 1), xx = t.x + (s & 1);
            const bool ok = t.valid && ys < H && xx < W;
            off[s] = ok ? ((t.n * H + ys) * W + xx) * d.in_C + d.in_coff + 4 * g : ERF_OOB;
        }
    };
    auto load_q = [&](const unsigned (&off)[4], int q, f32x4 (&xq)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) xq[s] = Act<T>::load4(r_in, off[s] + q * 16u);      // (the out-of-range flag survives the addition)
    };
    constexpr int NB = KQ == 1 ? 1 : (KQ % 2 == 0 ? 2 : 3);       // step q of a tile sits in buffer q % NB; KQ % NB == 0
    const unsigned stride = gridDim.x * 4;
    unsigned tile = blockIdx.x * 4 + wave;
    Geo tc = geo_of(tile);
    unsigned oc[4];
    src_offs(tc, oc);
    f32x4 xq[NB][4];
    if (NB > 1) load_q(oc, 0, xq[0]);
    for (; tile < ntiles; tile += stride) {
        const Geo tn = geo_of(tile + stride);
        unsigned on[4];
        src_offs(tn, on);
        unsigned ooff[4];
        f32x4 rres[4], rmask[4];
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (NB == 1) load_q(oc, 0, xq[0]);
            else if (q + 1 < KQ) load_q(oc, q + 1, xq[(q + 1) % NB]);
            else load_q(on, 0, xq[0]);
            if (q == KQ - 1) {      // epilogue operands of the four fine pixels
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned px = (tc.n * 2 * H + 2 * tc.y + (c >> 1)) * 2 * W + 2 * tc.x + (c & 1);
                    ooff[c] = tc.valid ? (px * d.out_C + d.out_coff + 4 * g) : ERF_OOB;
                    rres[c] = Act<T>::load4(r_res, tc.valid ? (px * d.epi.res_C + d.epi.res_coff + 4 * g) : ERF_OOB);
                    rmask[c] = Act<T>::load4(r_mask, tc.valid ? (px * d.epi.mask_C + d.epi.mask_coff + 4 * g) : ERF_OOB);
                }
                if (pooled) {
                    f32x4 xp[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned px = (tc.n * 2 * H + 2 * tc.y + (c >> 1)) * 2 * W + 2 * tc.x + (c & 1);
                        xp[c] = px_is_mask ? rmask[c] : Act<T>::load4(r_px, tc.valid ? (px * d.out_C + 4 * g) : ERF_OOB);
                    }
                    const unsigned pos = (tc.n * H + tc.y) * W + tc.x;
                    const f32x4 gp = Act<T>::load4(r_pg, tc.valid ? (pos * d.pool_gC + d.pool_gcoff + 4 * g) : ERF_OOB);
                    s16_pool_grad(xp, gp, rres);
                }
            }
            // (nothing moves across a step's boundary: left alone, the compiler reads the weight fragments
            // of every step ahead of the first MFMA — 236 registers, two waves per SIMD instead of three)
            __builtin_amdgcn_sched_barrier(0);
            f32x4 wf[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                wf[t] = *reinterpret_cast<const f32x4 *>(&Wl[((t * KQ + q) * 16 + i) * 16 + 4 * g]);
            // taps innermost: consecutive MFMAs go to different accumulators
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    acc[S16_CLASS(t)] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][m], xq[q % NB][S16_SRC(t)][m],
                                                                             acc[S16_CLASS(t)], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = acc[c][r] * sc[r] + sh[r] + rres[c][r];
                if (relu) e = fmaxf(e, 0.0f);
                v[r] = (!has_mask || rmask[c][r] > 0.0f) ? e : 0.0f;
            }
            Act<T>::store4(r_out, ooff[c], v);
        }
        tc = tn;
#pragma unroll
        for (int s = 0; s < 4; ++s) oc[s] = on[s];
    }
}

// up16_fwd_kernel for bf16 tensors on v_mfma_f32_16x16x32_bf16: 32 coarse channels per instruction,
// the bf16 data straight from memory as the MFMA operand (16-byte loads, no unpacking): 9 * NK MFMAs
// of 16 cycles per 16 coarse positions instead of 9 * 4 * KQ of 32.  Weights in LDS as bf16
// [tap][k / 32][c][32 k] (channels beyond K zero); lane (i, g) holds k = 32 kk + 8 g .. + 7.
template <int KQ, bool POOL = false>
__global__ __launch_bounds__(256) void up16_fwd_bf16_kernel(const IgemmDesc d)
{
    typedef bf16_t T;
    constexpr int NK = (KQ + 1) / 2;
    __shared__ __attribute__((aligned(16))) bf16_t Wl[9 * NK * 16 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    if (d.wsn == 9 && d.wsk == 144 && ((unsigned long long)d.w & 15) == 0) {
        // (dense [K][16][3][3] weight read in memory order, as in up16_fwd_kernel)
        if (KQ & 1) {       // the upper half of the last 32-channel group has no weights: zeros
            for (int v4 = tid; v4 < 9 * NK * 16 * 4; v4 += 256)
                *reinterpret_cast<u32x4 *>(&Wl[8 * v4]) = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
        }
        for (int v4 = tid; v4 < 9 * KQ * 64; v4 += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(d.w + 4 * v4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * v4 + c, k = e / 144, rem = e - k * 144, j = rem / 9, t = rem - j * 9;
                Wl[((t * NK + (k >> 5)) * 16 + j) * 32 + (k & 31)] = f32_to_bf16(v[c]);
            }
        }
    } else {
        for (int idx = tid; idx < 9 * NK * 16 * 32; idx += 256) {
            const int e = idx & 31, j = (idx >> 5) & 15, kk = (idx >> 9) % NK, t = idx / (NK * 512);
            const int k = kk * 32 + e;
            Wl[idx] = f32_to_bf16(k < 16 * KQ ? d.w[t + (long long)k * d.wsk + (long long)j * d.wsn] : 0.0f);
        }
    }
    __syncthreads();

    const unsigned H = (unsigned)d.GH, W = (unsigned)d.GW, M = (unsigned)d.N * H * W, HW = H * W;
    const unsigned Pout = M * 4u;
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, M * d.in_C * 2u);
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(d.out, Pout * d.out_C * 2u);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, Pout * d.epi.res_C * 2u);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, Pout * d.epi.mask_C * 2u);
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(d.epi.bias, 64u);
    const bool has_mask = d.epi.mask != nullptr, relu = d.epi.relu != 0;
    constexpr bool pooled = POOL;       // (the host instantiates POOL for a launch with d.pool_x)
    const bool px_is_mask = d.pool_x == d.epi.mask;     // (wave-uniform)
    const __amdgpu_buffer_rsrc_t r_px = make_rsrc(d.pool_x, Pout * d.out_C * 2u);
    const __amdgpu_buffer_rsrc_t r_pg = make_rsrc(d.pool_g, M * d.pool_gC * 2u);

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
    // this lane's channels of MFMA kk: 32 kk + 8 g .. + 7 (beyond K: nothing is loaded, the weights are zero)
    unsigned cbad[NK];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) cbad[kk] = 32 * kk + 8 * g < 16 * KQ ? 0u : ERF_OOB;

    const unsigned ntiles = (M + 15) / 16;
    for (unsigned tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const unsigned p = tile * 16 + i;
        const bool valid = p < M;
        const unsigned pp = valid ? p : 0;
        const unsigned n = pp / HW, rem = pp - n * HW;
        const unsigned y = rem / W, x = rem - y * W;
        u32x4 xs[4][NK];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned ys = y + (s >> 1), xx = x + (s & 1);
            const unsigned bad = (valid && ys < H && xx < W) ? 0u : ERF_OOB;
            const unsigned off = (((n * H + ys) * W + xx) * d.in_C + d.in_coff + 8 * g) * 2u;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk)
                xs[s][kk] = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)((off + kk * 64u) | bad | cbad[kk]), 0, 0);
        }
        unsigned ooff[4];
        f32x4 rres[4], rmask[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned px = (n * 2 * H + 2 * y + (c >> 1)) * 2 * W + 2 * x + (c & 1);
            ooff[c] = valid ? (px * d.out_C + d.out_coff + 4 * g) : ERF_OOB;
            rres[c] = Act<T>::load4(r_res, valid ? (px * d.epi.res_C + d.epi.res_coff + 4 * g) : ERF_OOB);
            rmask[c] = Act<T>::load4(r_mask, valid ? (px * d.epi.mask_C + d.epi.mask_coff + 4 * g) : ERF_OOB);
        }
        if (pooled) {
            f32x4 xp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned px = (n * 2 * H + 2 * y + (c >> 1)) * 2 * W + 2 * x + (c & 1);
                xp[c] = px_is_mask ? rmask[c] : Act<T>::load4(r_px, valid ? (px * d.out_C + 4 * g) : ERF_OOB);
            }
            const f32x4 gp = Act<T>::load4(r_pg, valid ? (pp * d.pool_gC + d.pool_gcoff + 4 * g) : ERF_OOB);
            s16_pool_grad(xp, gp, rres);
        }
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            __builtin_amdgcn_sched_barrier(0);      // (keeps the next step's weight fragments out of this one's registers)
            bf16x8 wf[9];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                wf[t] = *reinterpret_cast<const bf16x8 *>(&Wl[((t * NK + kk) * 16 + i) * 32 + 8 * g]);
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[S16_CLASS(t)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], __builtin_bit_cast(bf16x8, xs[S16_SRC(t)][kk]),
                                                                            acc[S16_CLASS(t)], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = acc[c][r] * sc[r] + sh[r] + rres[c][r];
                if (relu) e = fmaxf(e, 0.0f);
                v[r] = (!has_mask || rmask[c][r] > 0.0f) ? e : 0.0f;
            }
            Act<T>::store4(r_out, ooff[c], v);
        }
    }
}

// dW[tap][k][j] = sum_pos coarse[pos + d(tap)][k] * fine[2 pos + class(tap)][j]   (k < 16*KQ, j < 16)
// MFMA roles: row = k within its 16-channel quad, contraction = 4 consecutive coarse positions,
// column = j.  Lane (i, g) reads one float per source/class of position 4u + g.  Each wave keeps
// 9*KQ accumulators over its range of positions; the four waves of a block add them up in LDS in
// a fixed order and the block writes one slab:
//   [9][K][16] weight partials, [9][16] bias rows (row 0 = sum of the fine tensor, rest zero),
//   [K] sums of the coarse tensor (the bias gradient when the coarse side is the convolution output).
template <typename T, int KQ>
__global__ __launch_bounds__(256) void up16_wgrad_kernel(const WgradDesc d)
{
    constexpr int NACC = 9 * KQ, NW_ = NACC * 256, SLOTS = NW_ + 16 + 16 * KQ;
    __shared__ float red[SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    // blockIdx.y selects the block's KQ channel quads of the coarse side (of d.K / 16 in total):
    // fewer accumulators per wave, more waves in flight to hide the load latency
    const int q0 = blockIdx.y * KQ, KQT = d.K / 16;
    const unsigned H = (unsigned)d.GH, W = (unsigned)d.GW, M = (unsigned)d.N * H * W, HW = H * W;
    const unsigned wsp = (unsigned)d.split_pixels / 4;            // multiple of 4 (host)
    const unsigned wbeg = blockIdx.x * (unsigned)d.split_pixels + wave * wsp;
    unsigned wend = wbeg + wsp;
    if (wend > M) wend = M;
    const __amdgpu_buffer_rsrc_t r_c = make_rsrc(d.in, M * d.in_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_f = make_rsrc(d.g, M * 4u * d.g_C * Act<T>::B);

    f32x4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float fsum = 0.0f, csum[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) csum[q] = 0.0f;

    struct Frag {
        float cs[4][KQ], fc[4];
    };
    // coordinates of position m0 + g, kept per lane and advanced by 4 positions per call (the groups
    // are loaded in ascending order): one division pair per wave instead of one per group
    unsigned cn, cy, cx;
    {
        const unsigned p0 = (wbeg < M ? wbeg : 0u) + g;
        cn = p0 / HW;
        const unsigned rem = p0 - cn * HW;
        cy = rem / W;
        cx = rem - cy * W;
    }
    // element offsets of the lane's four coarse sources and four fine classes: formed once per image row
    // (a group later they are 4 coarse / 8 fine pixels further) — formed per group, this arithmetic was
    // most of the kernel's instructions
    unsigned soff[4], foff[4];
    bool row1;
    auto offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            soff[s] = ((cn * H + cy + (s >> 1)) * W + cx + (s & 1)) * d.in_C + d.in_coff + i + q0 * 16u;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            foff[c] = ((cn * 2 * H + 2 * cy + (c >> 1)) * 2 * W + 2 * cx + (c & 1)) * d.g_C + d.g_coff + i;
        row1 = cy + 1 < H;
    };
    offsets();
    const unsigned sstep = 4u * d.in_C, fstep = 8u * d.g_C;
    auto load_group = [&](unsigned m0, Frag &f) {
        const unsigned p = m0 + g;
        const bool valid = p < wend;
        const bool col1 = cx + 1 < W;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok = valid && ((s >> 1) ? row1 : true) && ((s & 1) ? col1 : true);
#pragma unroll
            for (int q = 0; q < KQ; ++q) f.cs[s][q] = Act<T>::load1(r_c, ok ? soff[s] + q * 16u : ERF_OOB);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) f.fc[c] = Act<T>::load1(r_f, valid ? foff[c] : ERF_OOB);
        cx += 4;
        if (cx >= W) {
            while (cx >= W) {
                cx -= W;
                if (++cy == H) { cy = 0; ++cn; }
            }
            offsets();
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) soff[s] += sstep;
#pragma unroll
            for (int c = 0; c < 4; ++c) foff[c] += fstep;
        }
    };
    auto mfma_group = [&](const Frag &f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) fsum += f.fc[c];
#pragma unroll
        for (int q = 0; q < KQ; ++q) csum[q] += f.cs[0][q];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int q = 0; q < KQ; ++q)
                acc[t * KQ + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.cs[S16_SRC(t)][q], f.fc[S16_CLASS(t)],
                                                                       acc[t * KQ + q], 0, 0, 0);
    };

    // two register sets: the loads of the next four positions are in flight during the MFMAs
    Frag fa, fb;
    if (wbeg < wend) load_group(wbeg, fa);
    for (unsigned m0 = wbeg; m0 < wend; m0 += 8) {
        if (m0 + 4 < wend) load_group(m0 + 4, fb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(fa);
        __builtin_amdgcn_sched_barrier(0);
        if (m0 + 8 < wend) load_group(m0 + 8, fa);
        __builtin_amdgcn_sched_barrier(0);
        if (m0 + 4 < wend) mfma_group(fb);
        __builtin_amdgcn_sched_barrier(0);
    }

    // sums over the position index g held by the four lane groups
    fsum += __shfl_xor(fsum, 16);
    fsum += __shfl_xor(fsum, 32);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        csum[q] += __shfl_xor(csum[q], 16);
        csum[q] += __shfl_xor(csum[q], 32);
    }
    // block reduction: the waves add into LDS one after the other (fixed order)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // accumulator a = tap*KQ + q holds rows k = 16q + 4g + r, column j = i
                    const int e = ((a / KQ) * 16 * KQ + (a % KQ) * 16 + 4 * g + r) * 16 + i;
                    red[e] = (w ? red[e] : 0.0f) + acc[a][r];
                }
            if (g == 0) {
                red[NW_ + i] = (w ? red[NW_ + i] : 0.0f) + fsum;
#pragma unroll
                for (int q = 0; q < KQ; ++q) red[NW_ + 16 + 16 * q + i] = (w ? red[NW_ + 16 + 16 * q + i] : 0.0f) + csum[q];
            }
        }
        __syncthreads();
    }
    // slab of the whole layer: [9][K][16] weights, 9 bias rows (row 0 carries the fine-side sums,
    // written by the first quad block), [K] coarse-side sums; this block fills its quads' share
    const int nwt = 9 * KQT * 256;
    const int slab = nwt + 9 * 16 + 16 * KQT;
    float *P = d.partial + (long long)blockIdx.x * slab;
    for (int e = tid; e < NW_; e += 256) {
        const int t = e / (KQ * 256), rest = e - t * (KQ * 256);      // rest = (q*16 + krow)*16 + j
        P[(t * KQT + q0) * 256 + rest] = red[e];
    }
    if (blockIdx.y == 0)
        for (int e = tid; e < 9 * 16; e += 256) P[nwt + e] = e < 16 ? red[NW_ + e] : 0.0f;
    for (int e = tid; e < 16 * KQ; e += 256) P[nwt + 9 * 16 + 16 * q0 + e] = red[NW_ + 16 + e];
}

// up16_wgrad_kernel for bf16 tensors on v_mfma_f32_16x16x32_bf16: 32 coarse positions per instruction.
// A wave stages, for 32 positions, the four coarse sources (its block's 16-channel quad) and the four
// fine classes as [32][16] bf16 tiles in its own 8 KiB of LDS (eight 16-byte loads per lane instead of
// sixty-four scalar ones) and takes the operands out with the transposing read (tr_frag16: lane
// (i, g) = channel i, positions 8g..8g+7).  One quad per block (blockIdx.y), slab as up16_wgrad_kernel.
__device__ __forceinline__ bf16x8 s16_tr_frag(const unsigned char *tile, int lane)
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

__global__ __launch_bounds__(256) void up16_wgrad_bf16_kernel(const WgradDesc d)
{
    constexpr int NW_ = 9 * 256, SLOTS = NW_ + 16 + 16;
    __shared__ float red[SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][8][32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int pl = lane >> 1, half = lane & 1;
    const int q0 = blockIdx.y, KQT = d.K / 16;
    const unsigned H = (unsigned)d.GH, W = (unsigned)d.GW, M = (unsigned)d.N * H * W, HW = H * W;
    const unsigned wsp = (unsigned)d.split_pixels / 4;            // multiple of 32 (host)
    const unsigned wbeg = blockIdx.x * (unsigned)d.split_pixels + wave * wsp;
    unsigned wend = wbeg + wsp;
    if (wend > M) wend = M;
    const __amdgpu_buffer_rsrc_t r_c = make_rsrc(d.in, M * d.in_C * 2u);
    const __amdgpu_buffer_rsrc_t r_f = make_rsrc(d.g, M * 4u * d.g_C * 2u);

    f32x4 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float fsum = 0.0f, csum = 0.0f;

    unsigned cn, cy, cx;        // coordinates of this lane's staging position, advanced by 32 per chunk
    {
        const unsigned p0 = (wbeg < M ? wbeg : 0u) + pl;
        cn = p0 / HW;
        const unsigned rem = p0 - cn * HW;
        cy = rem / W;
        cx = rem - cy * W;
    }
    u32x4 rc[4], rf[4];
    auto load_chunk = [&](unsigned m0) __attribute__((always_inline)) {
        const bool valid = m0 + pl < wend;
        const unsigned n = cn, y = cy, x = cx;
        cx += 32;
        while (cx >= W) {
            cx -= W;
            if (++cy == H) { cy = 0; ++cn; }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned ys = y + (s >> 1), xx = x + (s & 1);
            const bool ok = valid && ys < H && xx < W;
            const unsigned off = (((n * H + ys) * W + xx) * d.in_C + d.in_coff + 16 * q0 + 8 * half) * 2u;
            rc[s] = __builtin_amdgcn_raw_buffer_load_b128(r_c, (int)(ok ? off : ERF_OOB), 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned px = (n * 2 * H + 2 * y + (c >> 1)) * 2 * W + 2 * x + (c & 1);
            rf[c] = __builtin_amdgcn_raw_buffer_load_b128(r_f, (int)(valid ? (px * d.g_C + d.g_coff + 8 * half) * 2u : ERF_OOB), 0, 0);
        }
    };
    unsigned char *st = &stage[wave][0][0];
    if (wbeg < wend) load_chunk(wbeg);
    for (unsigned m0 = wbeg; m0 < wend; m0 += 32) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4 *>(st + s * 1024 + pl * 32 + half * 16) = rc[s];
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4 *>(st + (4 + c) * 1024 + pl * 32 + half * 16) = rf[c];
        if (m0 + 32 < wend) load_chunk(m0 + 32);
        __builtin_amdgcn_wave_barrier();
        bf16x8 cf[4], ff[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) cf[s] = s16_tr_frag(st + s * 1024, lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) ff[c] = s16_tr_frag(st + (4 + c) * 1024, lane);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) fsum += (float)ff[c][e];
#pragma unroll
        for (int e = 0; e < 8; ++e) csum += (float)cf[0][e];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[S16_SRC(t)], ff[S16_CLASS(t)], acc[t], 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
    }
    fsum += __shfl_xor(fsum, 16);
    fsum += __shfl_xor(fsum, 32);
    csum += __shfl_xor(csum, 16);
    csum += __shfl_xor(csum, 32);
    // block reduction: the waves add into LDS one after the other (fixed order)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 9; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = (a * 16 + 4 * g + r) * 16 + i;        // tap a, row k = 4g + r, column j = i
                    red[e] = (w ? red[e] : 0.0f) + acc[a][r];
                }
            if (g == 0) {
                red[NW_ + i] = (w ? red[NW_ + i] : 0.0f) + fsum;
                red[NW_ + 16 + i] = (w ? red[NW_ + 16 + i] : 0.0f) + csum;
            }
        }
        __syncthreads();
    }
    const int nwt = 9 * KQT * 256;
    const int slab = nwt + 9 * 16 + 16 * KQT;
    float *P = d.partial + (long long)blockIdx.x * slab;
    for (int e = tid; e < NW_; e += 256) {
        const int t = e / 256, rest = e - t * 256;
        P[(t * KQT + q0) * 256 + rest] = red[e];
    }
    if (blockIdx.y == 0)
        for (int e = tid; e < 9 * 16; e += 256) P[nwt + e] = e < 16 ? red[NW_ + e] : 0.0f;
    for (int e = tid; e < 16; e += 256) P[nwt + 9 * 16 + 16 * q0 + e] = red[NW_ + 16 + e];
}
