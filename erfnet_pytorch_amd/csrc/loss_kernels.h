// CrossEntropyLoss(ignore_index) over NCHW logits, mean over the labelled pixels
// (the reference's CrossEntropyLoss2d, train/evaluation_functions.py:6-14).
//
// HBM-bound: the logits are read once in forward (which keeps only log-sum-exp per pixel) and once
// in backward, and the gradient is written once — torch's log_softmax + nll_loss pair moves the
// N x K x H x W tensor six times.  A thread owns one pixel; consecutive threads take consecutive
// pixels, so every class plane is read with fully coalesced 4-byte loads.
#pragma once
#include "common.h"

#define CE_BLOCK 256
#define CE_MAX_K 32

struct CeDesc {
    const float *logits;        // [N][K][HW]
    const long long *target;    // [N][HW]
    float *lse;                 // [N][HW]
    float *partial;             // [blocks][2]: sum of -log p, number of labelled pixels
    long long P;                // N * HW
    int HW, K;
    long long ignore_index;
};

template <bool SMALL_K>
__global__ __launch_bounds__(CE_BLOCK) void ce_fwd_kernel(const CeDesc d)
{
    __shared__ float red[2][CE_BLOCK / 64];
    const long long p = (long long)blockIdx.x * CE_BLOCK + threadIdx.x;
    float loss = 0.0f, cnt = 0.0f;
    if (p < d.P) {
        const long long n = p / d.HW;
        const float *x = d.logits + n * (long long)d.K * d.HW + (p - n * d.HW);
        const long long t = d.target[p];
        float lse;
        if (SMALL_K) {
            // one pass over the logits: the class values stay in registers.  Slots beyond K repeat
            // the last class (unconditional loads; harmless for the max, left out of the sum).
            float v[CE_MAX_K];
#pragma unroll
            for (int k = 0; k < CE_MAX_K; ++k) v[k] = x[(long long)(k < d.K ? k : d.K - 1) * d.HW];
            float m = v[0];
#pragma unroll
            for (int k = 1; k < CE_MAX_K; ++k) m = fmaxf(m, v[k]);
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < CE_MAX_K; ++k) s += k < d.K ? __expf(v[k] - m) : 0.0f;
            lse = m + __logf(s);
        } else {
            float m = -INFINITY;
            for (int k = 0; k < d.K; ++k) m = fmaxf(m, x[(long long)k * d.HW]);
            float s = 0.0f;
            for (int k = 0; k < d.K; ++k) s += __expf(x[(long long)k * d.HW] - m);
            lse = m + __logf(s);
        }
        d.lse[p] = lse;
        if (t != d.ignore_index && t >= 0 && t < d.K) {
            loss = lse - x[t * d.HW];
            cnt = 1.0f;
        }
    }
    // block sums in a fixed order (wave shuffle tree, then the four waves in sequence)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        loss += __shfl_xor(loss, o);
        cnt += __shfl_xor(cnt, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = loss;
        red[1][threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int w = 0; w < CE_BLOCK / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        d.partial[2 * (long long)blockIdx.x + 0] = a;
        d.partial[2 * (long long)blockIdx.x + 1] = b;
    }
}

// stats[0] = mean loss, stats[1] = number of labelled pixels.  One block, fixed summation order
// (double accumulators: tens of thousands of partials).
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float *partial, int nblocks, float *stats)
{
    __shared__ double red[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += partial[2 * (long long)i];
        b += partial[2 * (long long)i + 1];
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = (float)(red[0][0] / red[1][0]);     // 0/0 = NaN when nothing is labelled, as in torch
        stats[1] = (float)red[1][0];
    }
}

struct CeBwdDesc {
    const float *logits;
    const long long *target;
    const float *lse, *stats, *gy;      // gy: gradient of the scalar loss (device scalar)
    float *dlogits;
    long long P;
    int HW, K;
    long long ignore_index;
};

// dlogits[n][k][p] = (softmax_k - [k == target]) * gy / n_labelled for labelled pixels, else 0.
__global__ __launch_bounds__(CE_BLOCK) void ce_bwd_kernel(const CeBwdDesc d)
{
    const long long p = (long long)blockIdx.x * CE_BLOCK + threadIdx.x;
    if (p >= d.P) return;
    const long long n = p / d.HW;
    const long long off = n * (long long)d.K * d.HW + (p - n * d.HW);
    const float *x = d.logits + off;
    float *g = d.dlogits + off;
    const long long t = d.target[p];
    const bool valid = t != d.ignore_index && t >= 0 && t < d.K;
    const float scale = valid ? d.gy[0] / d.stats[1] : 0.0f;
    const float lse = d.lse[p];
    for (int k = 0; k < d.K; ++k) {
        const float pk = __expf(x[(long long)k * d.HW] - lse);
        g[(long long)k * d.HW] = valid ? (pk - (k == t ? 1.0f : 0.0f)) * scale : 0.0f;
    }
}
