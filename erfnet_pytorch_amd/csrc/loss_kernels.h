// CrossEntropyLoss(ignore_index) over NCHW logits, mean over the labelled pixels
// (the reference's CrossEntropyLoss2d, train/evaluation_functions.py:6-14).
//
// HBM-bound: the logits are read once in forward (which keeps only log-sum-exp per pixel) and once
// in backward, and the gradient is written once — torch's log_softmax + nll_loss pair moves the
// N x K x H x W tensor six times.  A thread owns one pixel; consecutive threads take consecutive
// pixels, so every class plane is read with fully coalesced 4-byte loads.
// A label outside [0, K) that is not ignore_index counts as unlabelled (no loss, zero gradient, not in
// the mean's denominator); torch's kernel stops the process with a device-side assert on such a label.
#pragma once
#include "common.h"
#include "aux_kernels.h"

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
    unsigned *cpartial;         // optional (K <= CE_MAX_K): [blocks][2][K] correct / labelled pixels per label class
};

template <bool SMALL_K>
__global__ __launch_bounds__(CE_BLOCK) void ce_fwd_kernel(const CeDesc d)
{
    __shared__ float red[2][CE_BLOCK / 64];
    // accuracy counts on the side: the arg-max class is known once the class values are in registers
    __shared__ unsigned hist[2 * CE_MAX_K];
    const bool counting = SMALL_K && d.cpartial != nullptr;
    if (counting) {
        if (threadIdx.x < 2 * CE_MAX_K) hist[threadIdx.x] = 0;
        __syncthreads();
    }
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
            if (counting && t >= 0 && t < d.K) {
                // first largest value, NaN wins: torch's argmax
                float best = v[0];
                int arg = 0;
#pragma unroll
                for (int k = 1; k < CE_MAX_K; ++k) {
                    const bool take = k < d.K && (v[k] > best || (v[k] != v[k] && best == best));
                    best = take ? v[k] : best;
                    arg = take ? k : arg;
                }
                atomicAdd(&hist[d.K + (int)t], 1u);
                if (arg == (int)t) atomicAdd(&hist[(int)t], 1u);
            }
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
    if (counting) {         // (the barrier above also orders the LDS atomics)
        if (threadIdx.x < 2 * d.K) d.cpartial[(long long)blockIdx.x * 2 * d.K + threadIdx.x] = hist[threadIdx.x];
    }
}

// counts[j] = sum over blocks of cpartial[block][j], j < 2K: one block per column, exact integers.
__device__ __forceinline__ void ce_count_sum_body(const unsigned *cpartial, int nblocks, int J, long long *counts, int j)
{
    __shared__ long long red[256];
    long long s = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += cpartial[(long long)i * J + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[j] = red[0];
}
__global__ __launch_bounds__(256) void ce_count_sum_kernel(const unsigned *cpartial, int nblocks, int J, long long *counts)
{
    ce_count_sum_body(cpartial, nblocks, J, counts, blockIdx.x);
}

// stats[0] = mean loss, stats[1] = number of labelled pixels.  One block, fixed summation order
// (double accumulators: tens of thousands of partials).
__device__ __forceinline__ void ce_reduce_body(const float *partial, int nblocks, float *stats)
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
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float *partial, int nblocks, float *stats)
{
    ce_reduce_body(partial, nblocks, stats);
}
// Both reductions behind a loss forward that also counts (accuracy metrics): block 0 the loss, blocks
// 1 .. J the count columns — the same code, one launch.
__global__ __launch_bounds__(256) void ce_reduce_counts_kernel(const float *partial, int nblocks, float *stats,
                                                               const unsigned *cpartial, int J, long long *counts)
{
    if (blockIdx.x == 0) ce_reduce_body(partial, nblocks, stats);
    else ce_count_sum_body(cpartial, nblocks, J, counts, (int)blockIdx.x - 1);
}

// Output convolution + CrossEntropyLoss in one pass, for the training loop whose only use of the logits
// is `criterion(outputs, targets)` (train/main.py:327,343-347): the logits of
// ConvTranspose2d(16, K, 2, stride=2) (train/net_classification.py:56,64) are formed in registers —
// the same fused-multiply-add chain as outconv_fwd_kernel, so the log-sum-exp kept for backward equals
// the two-kernel path's bit for bit — and never reach memory: 64 B of x and 32 B of labels read, 16 B
// of log-sum-exp written per input pixel instead of 4*K*4 B of logits written and read again.
// One thread per INPUT pixel (four output positions); block sums in a fixed order.
struct OutConvCeFwdDesc {
    const void *x;              // NHWC [N][H][W][16], storage type T
    const float *w, *b;         // [16][K][2][2], [K]
    float *wpk;                 // workspace: the weights as [K][a][b][16] followed by the K biases
    const long long *target;    // [N][2H][2W]
    float *lse;                 // [N][2H][2W]
    float *partial;             // [blocks][2]
    unsigned *cpartial;         // optional: [blocks][2][K]
    int N, H, W, K;
    long long ignore_index;
};

// Weights in the order the loss kernel walks them, biases behind: wpk[(o*16 + c)*4 + ab], wpk[K*64 + o].
__global__ __launch_bounds__(256) void outconv_ce_pack_kernel(const OutConvCeFwdDesc d)
{
    const int K = d.K;
    for (int i = threadIdx.x; i < K * 4 * OC_CIN; i += 256) {
        const int ab = i % 4, c = (i / 4) % OC_CIN, o = i / (OC_CIN * 4);
        d.wpk[i] = d.w[(c * K + o) * 4 + ab];
    }
    for (int i = threadIdx.x; i < K; i += 256) d.wpk[K * 4 * OC_CIN + i] = d.b ? d.b[i] : 0.0f;
}

// The weights are the same for every thread: they are read through the constant address space, i.e.
// by scalar loads into scalar registers that the multiply-adds take as an operand — no LDS traffic
// (one LDS read per four multiply-adds bounded the first version of this kernel at 92 us).
typedef const __attribute__((address_space(4))) float oc_cfloat;

// KC > 0: the class count as a compile-time constant (must equal d.K); 0: any count up to 32.
template <typename T, int KC>
__global__ __launch_bounds__(256) void outconv_ce_fwd_kernel(const OutConvCeFwdDesc d)
{
    __shared__ float red[2][4];
    __shared__ unsigned hist[2 * CE_MAX_K];
    const int K = KC > 0 ? KC : d.K;
    oc_cfloat *ws = (oc_cfloat *)(uintptr_t)d.wpk;
    oc_cfloat *bs = ws + K * 4 * OC_CIN;
    const bool counting = d.cpartial != nullptr;
    if (threadIdx.x < 2 * CE_MAX_K) hist[threadIdx.x] = 0;
    __syncthreads();
    const long long M = (long long)d.N * d.H * d.W;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.0f, cnt = 0.0f;
    if (m < M) {
        const int n = (int)(m / (d.H * d.W));
        const int rem = (int)(m - (long long)n * d.H * d.W);
        const int i = rem / d.W, j = rem - i * d.W;
        float xv[OC_CIN];
#pragma unroll
        for (int q = 0; q < OC_CIN / 4; ++q) {
            const f32x4 v = Act<T>::ld4(d.x, m * OC_CIN + q * 4);
            xv[q * 4] = v[0]; xv[q * 4 + 1] = v[1]; xv[q * 4 + 2] = v[2]; xv[q * 4 + 3] = v[3];
        }
        const int Wo = 2 * d.W;
        const long long o0 = ((long long)n * 2 * d.H + 2 * i) * Wo + 2 * j;     // output position (a = 0, b = 0)
        long long tg[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            // (two consecutive int64 labels: one 16-byte load; the row offset is even, so it is aligned)
            const longlong2 t2 = *reinterpret_cast<const longlong2 *>(d.target + o0 + (long long)a * Wo);
            tg[a * 2] = t2.x; tg[a * 2 + 1] = t2.y;
        }
        // first pass over the classes: the maxima and the arg-max; the labelled class's logit on the way
        float mx[4], best[4], tl[4];
        int arg[4];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) { mx[ab] = -INFINITY; best[ab] = 0.0f; arg[ab] = 0; tl[ab] = 0.0f; }
        auto logits4 = [&](int o, float (&r4)[4]) __attribute__((always_inline)) {
            oc_cfloat *w0 = &ws[o * 4 * OC_CIN];
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) r4[ab] = bs[o];
#pragma unroll
            for (int c = 0; c < OC_CIN; ++c)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) r4[ab] = fmaf(xv[c], w0[c * 4 + ab], r4[ab]);
        };
        auto first_pass = [&](int o, const float (&r4)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const float r = r4[ab];
                mx[ab] = o == 0 ? r : fmaxf(mx[ab], r);
                // first largest value, NaN wins: torch's argmax
                const bool take = o == 0 || r > best[ab] || (r != r && best[ab] == best[ab]);
                best[ab] = take ? r : best[ab];
                arg[ab] = take ? o : arg[ab];
                tl[ab] = (long long)o == tg[ab] ? r : tl[ab];
            }
        };
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (KC > 0) {
            // the class count is a compile-time constant: every logit stays in a register, one pass of
            // multiply-adds; maxima and sums are formed in class order as in the two-pass form
            float lg[KC][4];
#pragma unroll
            for (int o = 0; o < KC; ++o) {
                logits4(o, lg[o]);
                first_pass(o, lg[o]);
            }
#pragma unroll
            for (int o = 0; o < KC; ++o)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) sum[ab] += __expf(lg[o][ab] - mx[ab]);
        } else {
            for (int o = 0; o < K; ++o) {
                float r4[4];
                logits4(o, r4);
                first_pass(o, r4);
            }
            // second pass: the sum of exponentials in class order (the order of ce_fwd_kernel)
            for (int o = 0; o < K; ++o) {
                float r4[4];
                logits4(o, r4);
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) sum[ab] += __expf(r4[ab] - mx[ab]);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f32x2 l2;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ab = a * 2 + b;
                const float lse = mx[ab] + __logf(sum[ab]);
                l2[b] = lse;
                const long long t = tg[ab];
                if (t >= 0 && t < K) {
                    if (counting) {
                        atomicAdd(&hist[K + (int)t], 1u);
                        if (arg[ab] == (int)t) atomicAdd(&hist[(int)t], 1u);
                    }
                    if (t != d.ignore_index) { loss += lse - tl[ab]; cnt += 1.0f; }
                }
            }
            *reinterpret_cast<f32x2 *>(d.lse + o0 + (long long)a * Wo) = l2;
        }
    }
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
        for (int w = 0; w < 4; ++w) { a += red[0][w]; b += red[1][w]; }
        d.partial[2 * (long long)blockIdx.x + 0] = a;
        d.partial[2 * (long long)blockIdx.x + 1] = b;
    }
    if (counting && threadIdx.x < 2 * K) d.cpartial[(long long)blockIdx.x * 2 * K + threadIdx.x] = hist[threadIdx.x];
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

// ------------------------------------------------------------------------------------------------
// Masked regression losses of train/evaluation_functions.py: the mean, over the pixels whose target
// is not -1, of a per-pixel value v(p):
//   ML_L1     L1LossMasked (:89-97)                        v = |o - t|
//   ML_LL     LogLikelihoodLossMasked (:71-87)             v = log(2 pi s) + (c - t)^2 / s,  s = var + eps
//   ML_L1CP   L1LossClassProbMasked (:32-47)               v = sum_k prob_k |cost_k - t|
//   ML_LLCP   LogLikelihoodLossClassProbMasked (:49-69)    v = sum_k prob_k (log(2 pi s_k) + (cost_k - t)^2 / s_k)
// a: o / c / prob ([N][K][HW], K = 1 for the first two); b: var [N][HW] (ML_LL) or cost [K];
// c: var [K] (ML_LLCP).  torch builds these from five to nine elementwise kernels and a masked_select
// (a prefix sum and a compaction); here the inputs are read once per direction.
// A thread owns ML_PPT pixels ML_BLOCK apart, so every plane is read with coalesced 4-byte loads.
#define ML_BLOCK 256
#define ML_PPT 4
#define ML_MAX_K 256
enum { ML_L1 = 0, ML_LL = 1, ML_L1CP = 2, ML_LLCP = 3 };

struct MlDesc {
    const float *a, *b, *c, *target;
    const float *stats, *gy;            // backward: stats[1] = number of labelled pixels, gy = d loss
    float *da, *db;                     // backward: gradients with the shape of a, and of b for ML_LL
    float *partial;                     // forward [blocks][2]; backward (class kinds) [blocks][2K]
    long long P;                        // N * HW
    int HW, K;
    float eps;
};

#define ML_TWO_PI 6.283185307179586f

// per-class constants of the class-probability kinds: cost, log(2 pi s), 1 / s
template <int KIND>
__device__ __forceinline__ void ml_class_consts(const MlDesc &d, float *cost, float *lv, float *iv)
{
    if (KIND == ML_L1CP || KIND == ML_LLCP) {
        for (int k = threadIdx.x; k < d.K; k += ML_BLOCK) {
            cost[k] = d.b[k];
            if (KIND == ML_LLCP) {
                const float s = d.c[k] + d.eps;
                lv[k] = logf(ML_TWO_PI * s);
                iv[k] = 1.0f / s;
            }
        }
        __syncthreads();
    }
}

template <int KIND>
__global__ __launch_bounds__(ML_BLOCK) void ml_fwd_kernel(const MlDesc d)
{
    __shared__ float cost[ML_MAX_K], lv[ML_MAX_K], iv[ML_MAX_K];
    __shared__ float red[2][ML_BLOCK / 64];
    ml_class_consts<KIND>(d, cost, lv, iv);
    float loss = 0.0f, cnt = 0.0f;
#pragma unroll
    for (int i = 0; i < ML_PPT; ++i) {
        const long long p = ((long long)blockIdx.x * ML_PPT + i) * ML_BLOCK + threadIdx.x;
        if (p >= d.P) break;
        const float t = d.target[p];
        float v;
        if (KIND == ML_L1) {
            v = fabsf(d.a[p] - t);
        } else if (KIND == ML_LL) {
            const float s = d.b[p] + d.eps, e = d.a[p] - t;
            v = logf(ML_TWO_PI * s) + e * e / s;
        } else {
            const long long n = p / d.HW;
            const float *pr = d.a + n * (long long)d.K * d.HW + (p - n * d.HW);
            v = 0.0f;
            for (int k = 0; k < d.K; ++k) {
                const float e = cost[k] - t;
                const float wk = KIND == ML_L1CP ? fabsf(e) : lv[k] + e * e * iv[k];
                v += wk * pr[(long long)k * d.HW];
            }
        }
        if (t != -1.0f) {
            loss += v;
            cnt += 1.0f;
        }
    }
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
        float x = 0.0f, y = 0.0f;
#pragma unroll
        for (int w = 0; w < ML_BLOCK / 64; ++w) { x += red[0][w]; y += red[1][w]; }
        d.partial[2 * (long long)blockIdx.x + 0] = x;
        d.partial[2 * (long long)blockIdx.x + 1] = y;
    }
}

// Gradients.  With g = gy / (number of labelled pixels), on labelled pixels (zero elsewhere):
//   ML_L1    do = sign(o - t) g
//   ML_LL    dc = 2 (c - t) / s g,               dvar = (1 / s - (c - t)^2 / s^2) g
//   ML_L1CP  dprob_k = |cost_k - t| g,           dcost_k = sum_p prob_k sign(cost_k - t) g
//   ML_LLCP  dprob_k = (log(2 pi s_k) + e^2/s_k) g,   dcost_k = sum_p prob_k 2 e / s_k g,
//            dvar_k = sum_p prob_k (1 / s_k - e^2 / s_k^2) g          (e = cost_k - t)
// The per-class sums leave the block as partial[block][k] (cost) and partial[block][K + k] (var) and
// are added in a fixed order by ml_colsum_kernel.
template <int KIND>
__global__ __launch_bounds__(ML_BLOCK) void ml_bwd_kernel(const MlDesc d)
{
    __shared__ float cost[ML_MAX_K], lv[ML_MAX_K], iv[ML_MAX_K];
    __shared__ float red[2][ML_BLOCK / 64];
    ml_class_consts<KIND>(d, cost, lv, iv);
    const float g = d.gy[0] / d.stats[1];
    float t[ML_PPT];
    long long off[ML_PPT];      // offset of the pixel inside plane 0 of its image; -1 beyond the end
#pragma unroll
    for (int i = 0; i < ML_PPT; ++i) {
        const long long p = ((long long)blockIdx.x * ML_PPT + i) * ML_BLOCK + threadIdx.x;
        if (p < d.P) {
            t[i] = d.target[p];
            const long long n = p / d.HW;
            off[i] = (KIND == ML_L1 || KIND == ML_LL) ? p : n * (long long)d.K * d.HW + (p - n * d.HW);
        } else {
            t[i] = -1.0f;
            off[i] = -1;
        }
    }
    if (KIND == ML_L1 || KIND == ML_LL) {
#pragma unroll
        for (int i = 0; i < ML_PPT; ++i) {
            if (off[i] < 0) continue;
            const bool valid = t[i] != -1.0f;
            const float e = d.a[off[i]] - t[i];
            if (KIND == ML_L1) {
                d.da[off[i]] = valid ? (e > 0.0f ? g : e < 0.0f ? -g : 0.0f) : 0.0f;
            } else {
                const float is = 1.0f / (d.b[off[i]] + d.eps);
                d.da[off[i]] = valid ? 2.0f * e * is * g : 0.0f;
                d.db[off[i]] = valid ? (is - e * e * is * is) * g : 0.0f;
            }
        }
        return;
    }
    for (int k = 0; k < d.K; ++k) {
        float sc = 0.0f, sv = 0.0f;
#pragma unroll
        for (int i = 0; i < ML_PPT; ++i) {
            if (off[i] < 0) continue;
            const bool valid = t[i] != -1.0f;
            const long long o = off[i] + (long long)k * d.HW;
            const float e = cost[k] - t[i];
            const float pr = valid ? d.a[o] : 0.0f;
            if (KIND == ML_L1CP) {
                d.da[o] = valid ? fabsf(e) * g : 0.0f;
                sc += pr * (e > 0.0f ? g : e < 0.0f ? -g : 0.0f);
            } else {
                d.da[o] = valid ? (lv[k] + e * e * iv[k]) * g : 0.0f;
                sc += pr * 2.0f * e * iv[k] * g;
                sv += pr * (iv[k] - e * e * iv[k] * iv[k]) * g;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sc += __shfl_xor(sc, o);
            sv += __shfl_xor(sv, o);
        }
        __syncthreads();        // the previous class's sums have been read
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = sc;
            red[1][threadIdx.x >> 6] = sv;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float x = 0.0f, y = 0.0f;
#pragma unroll
            for (int w = 0; w < ML_BLOCK / 64; ++w) { x += red[0][w]; y += red[1][w]; }
            d.partial[(long long)blockIdx.x * 2 * d.K + k] = x;
            d.partial[(long long)blockIdx.x * 2 * d.K + d.K + k] = y;
        }
    }
}

// out0[j] = sum over rows of partial[row][j], out1[j] = ... partial[row][J + j]   (rows of 2J floats);
// one block per column pair, fixed order, double accumulators.
__global__ __launch_bounds__(256) void ml_colsum_kernel(const float *partial, int rows, int J, float *out0, float *out1)
{
    __shared__ double red[2][256];
    const int j = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < rows; i += 256) {
        a += partial[(long long)i * 2 * J + j];
        b += partial[(long long)i * 2 * J + J + j];
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
        if (out0) out0[j] = (float)red[0][0];
        if (out1) out1[j] = (float)red[1][0];
    }
}

// Mean squared difference of two tensors of n elements (the elementwise part of MSELossWeighted,
// train/evaluation_functions.py:122-131; the caller multiplies by scale_factor * weight):
// partial[block] = {sum (a - b)^2, elements}; the gradient is 2 (a - b) gy / n.
struct MseDesc {
    const float *a, *b, *gy;
    float *da, *partial;
    long long n;
    float scale;
};

__global__ __launch_bounds__(ML_BLOCK) void mse_fwd_kernel(const MseDesc d)
{
    __shared__ float red[ML_BLOCK / 64];
    float s = 0.0f, c = 0.0f;
#pragma unroll
    for (int i = 0; i < ML_PPT; ++i) {
        const long long p = ((long long)blockIdx.x * ML_PPT + i) * ML_BLOCK + threadIdx.x;
        if (p < d.n) {
            const float e = d.a[p] - d.b[p];
            s += e * e;
            c += 1.0f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
    }
    __shared__ float redc[ML_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; redc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float x = 0.0f, y = 0.0f;
#pragma unroll
        for (int w = 0; w < ML_BLOCK / 64; ++w) { x += red[w]; y += redc[w]; }
        d.partial[2 * (long long)blockIdx.x] = x;
        d.partial[2 * (long long)blockIdx.x + 1] = y;
    }
}

__global__ __launch_bounds__(ML_BLOCK) void mse_bwd_kernel(const MseDesc d)
{
    const float g = 2.0f * d.gy[0] * d.scale;      // scale = 1 / n
#pragma unroll
    for (int i = 0; i < ML_PPT; ++i) {
        const long long p = ((long long)blockIdx.x * ML_PPT + i) * ML_BLOCK + threadIdx.x;
        if (p < d.n) d.da[p] = (d.a[p] - d.b[p]) * g;
    }
}
