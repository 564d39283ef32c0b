// Parameter updates of the training loops as multi-tensor kernels: one launch walks up to
// OPT_MAX_TENSORS parameter tensors (the whole model when its parameters live in flat buffers).
//   Adam   — train/main.py:213 (torch.optim.Adam with L2 weight decay, no amsgrad), stepped at :381
//   SGD    — imagenet/main.py:83-85 (momentum, L2 weight decay, no dampening, no Nesterov)
//   EMA    — train/main.py:140-142 (mean-teacher weights), called at :390
// Pure streaming work: Adam moves 28 B per element (p, g, m, v read; p, m, v written), SGD 20 B,
// EMA 12 B.  A block owns OPT_CHUNK consecutive elements of one tensor and finds that tensor by a
// scan over the (wave-uniform) chunk prefix table in the kernel arguments; 16-byte accesses when the
// tensor's pointers allow it.
#pragma once
#include <hip/hip_runtime.h>

#define OPT_MAX_TENSORS 32
#define OPT_BLOCK 256
#define OPT_VEC 4                                   // floats per access
#define OPT_ROUNDS 4                                // accesses per thread
#define OPT_CHUNK (OPT_BLOCK * OPT_VEC * OPT_ROUNDS)  // elements per block

enum { OPT_ADAM = 0, OPT_SGD = 1, OPT_EMA = 2 };

struct OptDesc {
    float *p[OPT_MAX_TENSORS];          // parameter (EMA: target), updated in place
    const float *g[OPT_MAX_TENSORS];    // gradient (EMA: source)
    float *m[OPT_MAX_TENSORS];          // Adam exp_avg / SGD momentum buffer
    float *v[OPT_MAX_TENSORS];          // Adam exp_avg_sq
    long long n[OPT_MAX_TENSORS];       // elements
    unsigned end_chunk[OPT_MAX_TENSORS];  // running total of chunks up to and including tensor i
    unsigned aligned;                   // bit i: every pointer of tensor i is 16-byte aligned
    int count;
    // Adam: c0 = 1 - beta1, c1 = beta2, c2 = 1 - beta2, c3 = lr / (1 - beta1^t), c4 = sqrt(1 - beta2^t), c5 = eps
    // SGD:  c0 = momentum, c3 = lr      EMA: c0 = 1 - discount, c1 = discount
    float c0, c1, c2, c3, c4, c5, wd;
};

template <int KIND>
__device__ __forceinline__ void opt_update(const OptDesc &d, float &p, const float g0, float &m, float &v)
{
    // Every operation rounds on its own: the vector and the scalar path of the kernel must agree bit
    // for bit, and the mean-teacher update must equal the reference's three tensor operations.  Plain
    // operators under this pragma (the __fmul_rn family are header functions compiled outside it, and
    // the back end fuses their results); the fused multiply-adds are the explicit ones.  Division
    // and square root are hipcc's correctly rounded defaults.
#pragma clang fp contract(off)
    if (KIND == OPT_ADAM) {
        // torch.optim.Adam (single-tensor form): grad += wd * p; exp_avg.lerp_(grad, 1 - beta1);
        // exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad;
        // p -= step_size * exp_avg / (sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps)
        const float g = d.wd != 0.0f ? __builtin_fmaf(d.wd, p, g0) : g0;
        m = __builtin_fmaf(d.c0, g - m, m);
        v = d.c1 * v + (d.c2 * g) * g;
        const float denom = __builtin_sqrtf(v) / d.c4 + d.c5;
        p = p - (d.c3 * m) / denom;
    } else if (KIND == OPT_SGD) {
        // torch.optim.SGD: grad += wd * p; buf = momentum * buf + grad (buf = grad on the first step:
        // the caller starts from a zero buffer, momentum * 0 + grad is the same number); p -= lr * buf
        const float g = d.wd != 0.0f ? __builtin_fmaf(d.wd, p, g0) : g0;
        m = d.c0 * m + g;
        p = p - d.c3 * m;
    } else {
        // target = target * (1 - discount) + discount * source, both products rounded as the
        // reference's two tensor multiplications are
        p = p * d.c0 + d.c1 * g0;
    }
}

template <int KIND>
__global__ __launch_bounds__(OPT_BLOCK) void opt_step_kernel(const OptDesc d)
{
    int t = 0;
    while (t < d.count - 1 && blockIdx.x >= d.end_chunk[t]) ++t;
    const unsigned first = t ? d.end_chunk[t - 1] : 0u;
    const long long base = (long long)(blockIdx.x - first) * OPT_CHUNK;
    const long long n = d.n[t];
    float *__restrict__ P = d.p[t];
    const float *__restrict__ G = d.g[t];
    float *__restrict__ M = d.m[t];
    float *__restrict__ V = d.v[t];
    if (((d.aligned >> t) & 1u) && base + OPT_CHUNK <= n) {
        float4 p[OPT_ROUNDS], g[OPT_ROUNDS], m[OPT_ROUNDS], v[OPT_ROUNDS];
#pragma unroll
        for (int r = 0; r < OPT_ROUNDS; ++r) {
            const long long i = base + ((long long)r * OPT_BLOCK + threadIdx.x) * OPT_VEC;
            p[r] = *(const float4 *)(P + i);
            g[r] = *(const float4 *)(G + i);
            if (KIND != OPT_EMA) m[r] = *(const float4 *)(M + i);
            if (KIND == OPT_ADAM) v[r] = *(const float4 *)(V + i);
        }
#pragma unroll
        for (int r = 0; r < OPT_ROUNDS; ++r) {
            const long long i = base + ((long long)r * OPT_BLOCK + threadIdx.x) * OPT_VEC;
            opt_update<KIND>(d, p[r].x, g[r].x, m[r].x, v[r].x);
            opt_update<KIND>(d, p[r].y, g[r].y, m[r].y, v[r].y);
            opt_update<KIND>(d, p[r].z, g[r].z, m[r].z, v[r].z);
            opt_update<KIND>(d, p[r].w, g[r].w, m[r].w, v[r].w);
            *(float4 *)(P + i) = p[r];
            if (KIND != OPT_EMA) *(float4 *)(M + i) = m[r];
            if (KIND == OPT_ADAM) *(float4 *)(V + i) = v[r];
        }
        return;
    }
    // ragged last chunk of a tensor, or a tensor that is not 16-byte aligned
    for (int r = 0; r < OPT_ROUNDS * OPT_VEC; ++r) {
        const long long i = base + (long long)r * OPT_BLOCK + threadIdx.x;
        if (i >= n) break;
        float p = P[i], m = 0.0f, v = 0.0f;
        if (KIND != OPT_EMA) m = M[i];
        if (KIND == OPT_ADAM) v = V[i];
        opt_update<KIND>(d, p, G[i], m, v);
        P[i] = p;
        if (KIND != OPT_EMA) M[i] = m;
        if (KIND == OPT_ADAM) V[i] = v;
    }
}
