// Shared descriptors and helpers of the ERFNet HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Activation tensors are stored as fp32 or as bf16 (raw 16-bit patterns, bf16_t); arithmetic outside
// the bf16 MFMA operands is fp32 either way.  Kernels that serve both are templates on the storage
// type T and go through the helpers below; descriptors carry activation pointers as void*.
typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
// round to nearest even; NaN stays NaN (v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ u32x2 pack_bf16x4(f32x4 v)
{
    return __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));      // two v_cvt_pk_bf16_f32
}
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2 u)
{
    f32x4 v;
    v[0] = __builtin_bit_cast(float, u[0] << 16);
    v[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, u[1] << 16);
    v[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
    return v;
}

#define ERF_MAX_TAPS 9
#define ERF_MAX_GROUPS 4

// Per-channel affine applied in a kernel epilogue: BatchNorm in eval mode, or in train mode
// once the batch statistics are known.  v = (v - mean) * invstd * gamma + beta, where
// invstd = var_is_invstd ? var[c] : rsqrt(var[c] + eps).
struct BnAffine {
    const float *gamma, *beta, *mean, *var;
    float eps;
    int var_is_invstd;
    int on;
};

// Epilogue shared by the convolution, pooling and element-wise kernels, applied in this order:
//   v += bias[c]; v = bn(v); v *= nscale[n][c]; v += res[pix][c]; v = relu(v); v = mask[pix][c] > 0 ? v : 0
// res and mask are NHWC tensors addressed with the output's pixel index.
struct Epilogue {
    const float *bias;
    BnAffine bn;
    const float *nscale;   // [N][Nout]  (Dropout2d keep/scale factors)
    const void *res;       // activation tensors (storage type of the kernel)
    int res_C, res_coff;
    const void *mask;
    int mask_C, mask_coff;
    int relu;
    // ReLU masks as bit tensors (line_x3r_kernel only; both optional): one 32-bit word per pixel and
    // group of 32 channels, bit c % 32 of word [pixel][c / 32] set iff the value is > 0.
    unsigned *bits_out;         // written for this launch's output (after every other epilogue step)
    const unsigned *mask_bits;  // read in place of `mask` (same meaning: keep the value where the bit is set)
};

// Implicit-GEMM convolution: out[pix(m)][co] = epi( sum_t sum_k in[src(m,t)][k] * W[t][k][co] ).
//   m = (n, gi, gj) enumerates an output grid GH x GW per image;
//   source pixel  = (gi*ss + tdh[t], gj*ss + tdw[t]) of in[N][in_H][in_W][in_C] (+in_coff), zero outside;
//   output pixel  = (gi*os + goa[g], gj*os + gob[g]) of out[N][out_H][out_W][out_C] (+out_coff);
//   weights       : W[t][k][j] = w[woff[t] + k*wsk + j*wsn] in torch's layout, repacked per call
//                   into wp[t][k/32][j][32] (zero padded to Npad columns) by pack_weights_kernel;
//   tap groups    : blockIdx.z = g selects taps [gtap0[g], gtap0[g]+gntaps[g]) — the output-parity
//                   classes of a stride-2 transposed convolution; ordinary convolutions have 1 group.
struct IgemmDesc {
    const void *in;         // activations (storage type of the kernel)
    const float *w;         // weights in torch's layout, always fp32
    const void *wp;         // packed weights: fp32, or bf16 for the bf16 kernels
    void *out;
    Epilogue epi;
    int N, GH, GW;
    int in_H, in_W, in_C, in_coff, K;
    int ss;
    int tdh[ERF_MAX_TAPS], tdw[ERF_MAX_TAPS], woff[ERF_MAX_TAPS];
    int wsk, wsn;
    int ngroups, gtap0[ERF_MAX_GROUPS], gntaps[ERF_MAX_GROUPS], goa[ERF_MAX_GROUPS], gob[ERF_MAX_GROUPS];
    int out_H, out_W, out_C, out_coff, Nout, os;
    int dt;            // storage type of the activation tensors: 0 = fp32, 1 = bf16
    int Npad;          // padded column count of wp (multiple of the block's BN)
    int cfg;           // host side: tile configuration chosen by plan_igemm
    int vec_ok;        // in_C, in_coff, K all multiples of 4 and `in` 16-byte aligned
    int x3_plane;      // igemm_x3_kernel: bf16 elements per plane of the split packed weights
    // up16_fwd kernels only: the 2x2 max-pool's gradient formed in the epilogue (in place of a residual
    // tensor written by pool_bwd_kernel): pool_x = the pooled tensor (fine grid, out_C channels),
    // pool_g = the gradient at the coarse grid, channels [pool_gcoff, pool_gcoff + 16) of pool_gC
    const void *pool_x, *pool_g;
    int pool_gC, pool_gcoff;
    double *bn_partial;    // host side, may be null: line_x3_kernel leaves per-block column sums of the output
                           // and of its squares here ([blocks][2 * Nout]: train-mode BatchNorm statistics)
};

// Weight gradient: dW[t][k][j] = sum_m in[src(m,t)][k] * g[gpix(m,t)][j], reduced over pixels in
// `nsplit` slabs: partial[s][t][k][j] (k < K, j < Nout), plus bias row partial_b[s][j].
//   g pixel = (gi*os + toa[t], gj*os + tob[t]) of g[N][g_H][g_W][g_C] (+g_coff).
struct WgradDesc {
    const void *in, *g;     // activations / gradients (storage type of the kernel)
    float *partial;     // [nsplit][ntaps*K*Nout + Nout]
    int N, GH, GW;
    int in_H, in_W, in_C, in_coff, K;
    int ss, ntaps;
    int tdh[ERF_MAX_TAPS], tdw[ERF_MAX_TAPS], toa[ERF_MAX_TAPS], tob[ERF_MAX_TAPS];
    int g_H, g_W, g_C, g_coff, Nout, os;
    int nsplit, split_pixels;   // pixels per slab (multiple of 32)
    int itiles, jtiles;
    int vec_ok;
    int dt;            // storage type of in / g: 0 = fp32, 1 = bf16
};

// Global loads go through buffer descriptors: a lane whose operand is masked (tap outside the
// image, pixel beyond the range, optional tensor absent) presents an out-of-range offset and the
// hardware returns zero without touching memory.  Every load is therefore unconditional and its
// value is used as it is — no branch around a load and no select behind it, either of which makes
// the compiler wait for each load separately.  Descriptors cover at most 2^31 bytes (host check).
#define ERF_OOB 0x80000000u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    // readfirstlane makes the descriptor words provably wave-uniform; otherwise the compiler
    // wraps every buffer access in a per-lane "waterfall" loop
    const uintptr_t a = (uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned n = __builtin_amdgcn_readfirstlane(p ? bytes : 0u);
    float *q = reinterpret_cast<float *>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)n, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 0);
}

// 16-byte buffer store whose scalar offset sits in a REGISTER.  The store fetches its data registers
// after it has issued; an instruction right behind it that overwrites them (the next tile's first
// value landing in the register the allocator has just "freed") makes the last lanes store the new
// value of the first data register.  The compiler places the required wait state only for the
// immediate-offset form, so here one wait state that names the data as its operand (nothing that
// overwrites the registers can be scheduled in front of it) follows the store.
__device__ __forceinline__ void buf_store4_soff(u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
    asm volatile("s_nop 0" ::"v"(v) : "memory");
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, 0);
}

// Storage-type view of the same accesses: offsets are in ELEMENTS (bit 31 = out of range), a quad is
// four consecutive channels.
template <typename T> struct Act;
template <> struct Act<float> {
    static constexpr unsigned B = 4;
    static __device__ __forceinline__ f32x4 load4(__amdgpu_buffer_rsrc_t r, unsigned e) { return buf_load4(r, (e << 2) | (e & ERF_OOB)); }
    static __device__ __forceinline__ float load1(__amdgpu_buffer_rsrc_t r, unsigned e) { return buf_load1(r, (e << 2) | (e & ERF_OOB)); }
    static __device__ __forceinline__ f32x2 load2(__amdgpu_buffer_rsrc_t r, unsigned e)
    {
        return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)((e << 2) | (e & ERF_OOB)), 0, 0));
    }
    static __device__ __forceinline__ void store4(__amdgpu_buffer_rsrc_t r, unsigned e, f32x4 v) { buf_store4(r, (e << 2) | (e & ERF_OOB), v); }
    static __device__ __forceinline__ void store1(__amdgpu_buffer_rsrc_t r, unsigned e, float v) { buf_store1(r, (e << 2) | (e & ERF_OOB), v); }
    static __device__ __forceinline__ f32x4 ld4(const void *p, long long e) { return *reinterpret_cast<const f32x4 *>((const float *)p + e); }
    static __device__ __forceinline__ float ld1(const void *p, long long e) { return ((const float *)p)[e]; }
    static __device__ __forceinline__ void st4(void *p, long long e, f32x4 v) { *reinterpret_cast<f32x4 *>((float *)p + e) = v; }
    static __device__ __forceinline__ void st1(void *p, long long e, float v) { ((float *)p)[e] = v; }
    static __device__ __forceinline__ void st2(void *p, long long e, f32x2 v) { *reinterpret_cast<f32x2 *>((float *)p + e) = v; }
    static __device__ __forceinline__ f32x2 ld2(const void *p, long long e) { return *reinterpret_cast<const f32x2 *>((const float *)p + e); }
};
template <> struct Act<bf16_t> {
    static constexpr unsigned B = 2;
    static __device__ __forceinline__ f32x4 load4(__amdgpu_buffer_rsrc_t r, unsigned e)
    {
        return unpack_bf16x4(__builtin_amdgcn_raw_buffer_load_b64(r, (int)((e << 1) | (e & ERF_OOB)), 0, 0));
    }
    static __device__ __forceinline__ float load1(__amdgpu_buffer_rsrc_t r, unsigned e)
    {
        return bf16_to_f32(__builtin_amdgcn_raw_buffer_load_b16(r, (int)((e << 1) | (e & ERF_OOB)), 0, 0));
    }
    static __device__ __forceinline__ f32x2 load2(__amdgpu_buffer_rsrc_t r, unsigned e)
    {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, (int)((e << 1) | (e & ERF_OOB)), 0, 0);
        return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
    }
    static __device__ __forceinline__ void store4(__amdgpu_buffer_rsrc_t r, unsigned e, f32x4 v)
    {
        __builtin_amdgcn_raw_buffer_store_b64(pack_bf16x4(v), r, (int)((e << 1) | (e & ERF_OOB)), 0, 0);
    }
    static __device__ __forceinline__ void store1(__amdgpu_buffer_rsrc_t r, unsigned e, float v)
    {
        __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16(v), r, (int)((e << 1) | (e & ERF_OOB)), 0, 0);
    }
    static __device__ __forceinline__ f32x4 ld4(const void *p, long long e)
    {
        return unpack_bf16x4(*reinterpret_cast<const u32x2 *>((const bf16_t *)p + e));
    }
    static __device__ __forceinline__ float ld1(const void *p, long long e) { return bf16_to_f32(((const bf16_t *)p)[e]); }
    static __device__ __forceinline__ void st4(void *p, long long e, f32x4 v) { *reinterpret_cast<u32x2 *>((bf16_t *)p + e) = pack_bf16x4(v); }
    static __device__ __forceinline__ void st1(void *p, long long e, float v) { ((bf16_t *)p)[e] = f32_to_bf16(v); }
    static __device__ __forceinline__ f32x2 ld2(const void *p, long long e)
    {
        const unsigned u = *reinterpret_cast<const unsigned *>((const bf16_t *)p + e);
        return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
    }
    static __device__ __forceinline__ void st2(void *p, long long e, f32x2 v)
    {
        bf16x2 o;
        o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1];
        *reinterpret_cast<unsigned *>((bf16_t *)p + e) = __builtin_bit_cast(unsigned, o);
    }
};

// Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2.
// Tiles that share input rows (the taps of a convolution reach d rows up and down) should meet in
// one L2, so the tile index is remapped: XCD x works through the x-th contiguous eighth of the
// tiles in order.  Only a locality hint — correctness does not depend on the hardware's mapping.
__device__ __forceinline__ unsigned xcd_tile(unsigned bid, unsigned nb)
{
    const unsigned q = nb >> 3, r = nb & 7u, x = bid & 7u, slot = bid >> 3;
    return x < r ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
}

#define ERF_WGRAD_BATCH 4
struct WgradBatch {
    WgradDesc d[ERF_WGRAD_BATCH];
};

__device__ __forceinline__ float apply_bn(const BnAffine &bn, float v, int c)
{
    const float inv = bn.var_is_invstd ? bn.var[c] : rsqrtf(bn.var[c] + bn.eps);
    const float g = bn.gamma ? bn.gamma[c] : 1.0f;
    const float b = bn.beta ? bn.beta[c] : 0.0f;
    return (v - bn.mean[c]) * (inv * g) + b;
}

// Channel-only part of the epilogue folded into two numbers: v*sc + sh.
__device__ __forceinline__ void epi_channel_affine(const Epilogue &e, int c, float &sc, float &sh)
{
    sc = 1.0f;
    sh = e.bias ? e.bias[c] : 0.0f;
    if (e.bn.on) {
        const float inv = e.bn.var_is_invstd ? e.bn.var[c] : rsqrtf(e.bn.var[c] + e.bn.eps);
        const float g = (e.bn.gamma ? e.bn.gamma[c] : 1.0f) * inv;
        const float b = e.bn.beta ? e.bn.beta[c] : 0.0f;
        // ((v + bias) - mean) * g + b
        sh = (sh - e.bn.mean[c]) * g + b;
        sc = g;
    }
}

template <typename T = float>
__device__ __forceinline__ float epi_finish(const Epilogue &e, float v, long long pix, int n, int c,
                                            int Nout)
{
    if (e.nscale) v *= e.nscale[(long long)n * Nout + c];
    if (e.res) v += Act<T>::ld1(e.res, pix * e.res_C + e.res_coff + c);
    if (e.relu) v = fmaxf(v, 0.0f);
    if (e.mask) v = Act<T>::ld1(e.mask, pix * e.mask_C + e.mask_coff + c) > 0.0f ? v : 0.0f;
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter, i.e. it would wait for the output stores of the tile just finished and for the DMA of the
// next tile, which are meant to stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
