// Per-class true-positive / false-positive / false-negative counts of a label map against its
// ground truth (the arithmetic of iouEval.addBatch, reference train/iouEval.py:21-70, for inputs
// given as class indices): one pass over the two int64 maps instead of two one-hot tensors and
// six reductions.  Counting is exact integer work: LDS histogram per block, per-block partials,
// fixed-order sum — no floating point, no global atomics, bitwise reproducible.
#pragma once
#include "common.h"

#define IOU_MAX_CLASSES 256

struct IouDesc {
    const long long *pred, *target;
    long long n;
    int n_classes;          // labels outside [0, n_classes) match no class
    int counted;            // classes [0, counted) are tallied
    int ignore;             // target label that cancels a false positive, or -1
    unsigned *partial;      // [blocks][3][counted]
    long long *counts;      // [3][counted]: tp, fp, fn (accumulated)
    int blocks;
};

__global__ __launch_bounds__(256) void iou_count_kernel(const IouDesc d)
{
    __shared__ unsigned hist[3 * IOU_MAX_CLASSES];
    for (int i = threadIdx.x; i < 3 * d.counted; i += 256) hist[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += stride) {
        const long long x = d.pred[i], y = d.target[i];
        const bool xin = x >= 0 && x < d.counted, yin = y >= 0 && y < d.counted;
        if (xin && x == y) {
            atomicAdd(&hist[x], 1u);                                   // tp
        } else {
            if (xin && y != d.ignore) atomicAdd(&hist[d.counted + x], 1u);      // fp
            if (yin) atomicAdd(&hist[2 * d.counted + y], 1u);                  // fn
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * d.counted; i += 256)
        d.partial[(long long)blockIdx.x * 3 * d.counted + i] = hist[i];
}

__global__ __launch_bounds__(256) void iou_sum_kernel(const IouDesc d)
{
    for (int i = threadIdx.x; i < 3 * d.counted; i += 256) {
        long long s = 0;
        for (int b = 0; b < d.blocks; ++b) s += d.partial[(long long)b * 3 * d.counted + i];
        d.counts[i] += s;
    }
}

// ClassificationAccuracy / MeanAccuracy of the reference (train/evaluation_functions.py:133-163):
// per target class, the number of pixels labelled with it and the number of those whose argmax over
// the class scores equals the label.  One pass over the NCHW scores (a thread owns a pixel, every
// class plane is read with coalesced 4-byte loads, the first largest value wins like argmax), LDS
// histogram per block, per-block partials, fixed-order sum.
struct AccDesc {
    const float *scores;        // [N][K][HW]
    const long long *target;    // [N][HW]; labels outside [0, K) are not counted
    long long P;                // N * HW
    int HW, K;
    unsigned *partial;          // [blocks][2][K]
    long long *counts;          // [2][K]: correct, total (accumulated)
    int blocks;
};

__global__ __launch_bounds__(256) void acc_count_kernel(const AccDesc d)
{
    __shared__ unsigned hist[2 * IOU_MAX_CLASSES];
    for (int i = threadIdx.x; i < 2 * d.K; i += 256) hist[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * 256;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < d.P; p += stride) {
        const long long t = d.target[p];
        if (t < 0 || t >= d.K) continue;
        const long long n = p / d.HW;
        const float *x = d.scores + n * (long long)d.K * d.HW + (p - n * d.HW);
        float best = x[0];
        int arg = 0;
        for (int k = 1; k < d.K; ++k) {
            const float v = x[(long long)k * d.HW];
            const bool take = v > best || (v != v && best == best);     // NaN wins, as in torch
            best = take ? v : best;
            arg = take ? k : arg;
        }
        atomicAdd(&hist[d.K + (int)t], 1u);
        if (arg == (int)t) atomicAdd(&hist[(int)t], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * d.K; i += 256)
        d.partial[(long long)blockIdx.x * 2 * d.K + i] = hist[i];
}

__global__ __launch_bounds__(256) void acc_sum_kernel(const AccDesc d)
{
    for (int i = threadIdx.x; i < 2 * d.K; i += 256) {
        long long s = 0;
        for (int b = 0; b < d.blocks; ++b) s += d.partial[(long long)b * 2 * d.K + i];
        d.counts[i] += s;
    }
}
