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
