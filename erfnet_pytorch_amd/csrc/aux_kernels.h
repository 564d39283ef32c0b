// Memory-bound companions of the implicit-GEMM kernels: the 3-channel network-input downsampler,
// 2x2 max-pool (forward / backward), BatchNorm statistics / apply / backward, element-wise
// gradient masking and the 2x2 stride-2 output transposed convolution.  NHWC activations stored as
// fp32 or bf16 (template parameter T / AT), arithmetic in fp32.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// DownsamplerBlock on the network input (Cin = 3): direct 3x3 stride-2 convolution + 2x2 max-pool,
// one thread per output pixel.  x is read through its strides (NCHW images need no repacking).
// Weights [Cc][3][3][3] and bias are staged in LDS (broadcast reads).  Output NHWC [*, Cc+3].
struct FirstDownDesc {
    const void *x;       // image, storage type T
    long long sn, sc, sh, sw;
    const float *w;      // [Cc][27]
    void *out;           // [N][Ho][Wo][Cc+3], storage type T
    Epilogue epi;
    int N, H, W, Cc;
    unsigned x_elems;    // extent of x covered by its strides when it fits a buffer descriptor (below 2^29), else 0
};

#define FD_MAX_CC 61
template <typename T>
__global__ __launch_bounds__(256) void first_down_fwd_kernel(const FirstDownDesc d)
{
    __shared__ float ws[FD_MAX_CC * 28];
    const int Cout = d.Cc + 3;
    // the network's own first block (3 -> 16 channels) reads its weights through the scalar data path
    // (below); every other shape stages them in LDS
    const bool own = d.Cc == 13 && !d.epi.nscale && !d.epi.res && !d.epi.mask;
    if (!own) {
        for (int i = threadIdx.x; i < d.Cc * 28; i += 256) {
            const int co = i / 28, e = i % 28;
            ws[i] = e < 27 ? d.w[co * 27 + e] : 0.0f;
        }
        __syncthreads();
    }
    const int Ho = d.H / 2, Wo = d.W / 2;
    const long long M = (long long)d.N * Ho * Wo;
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int n = (int)(m / (Ho * Wo));
    const int rem = (int)(m - (long long)n * Ho * Wo);
    const int oh = rem / Wo, ow = rem - oh * Wo;
    float xin[27];
    // Image rows with unit pixel stride and even strides / base (NCHW images of even width): the three
    // columns 2ow-1 .. 2ow+1 of a row come from two aligned pair loads, (2ow-2, 2ow-1) and (2ow, 2ow+1),
    // each of which a wave reads as one contiguous run; otherwise element by element.
    const bool pairs = d.sw == 1 && !((d.sn | d.sc | d.sh) & 1) && !((uintptr_t)d.x & (2 * Act<T>::B - 1));
    if (pairs && d.x_elems) {
        // the eighteen pair loads go out together (buffer loads: a pair outside the image presents an
        // out-of-range offset and reads zero) — a branch around each load made the wave wait for one
        // memory round trip after the other
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, d.x_elems * Act<T>::B);
        const unsigned sn = (unsigned)d.sn, sc = (unsigned)d.sc, sh = (unsigned)d.sh;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = 2 * oh - 1 + kh;
                const bool rok = ih >= 0 && ih < d.H;
                const unsigned e = (unsigned)n * sn + (unsigned)ci * sc + (unsigned)ih * sh + 2u * (unsigned)ow;
                const f32x2 lo = Act<T>::load2(rx, (rok && ow > 0) ? e - 2u : ERF_OOB);
                const f32x2 hi = Act<T>::load2(rx, rok ? e : ERF_OOB);
                xin[ci * 9 + kh * 3 + 0] = lo[1];
                xin[ci * 9 + kh * 3 + 1] = hi[0];
                xin[ci * 9 + kh * 3 + 2] = hi[1];
            }
    } else if (pairs) {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = 2 * oh - 1 + kh;
                const bool rok = ih >= 0 && ih < d.H;
                const long long e = n * d.sn + ci * d.sc + (long long)ih * d.sh + 2 * ow;
                f32x2 lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
                if (rok && ow > 0) lo = Act<T>::ld2(d.x, e - 2);
                if (rok) hi = Act<T>::ld2(d.x, e);
                xin[ci * 9 + kh * 3 + 0] = lo[1];
                xin[ci * 9 + kh * 3 + 1] = hi[0];
                xin[ci * 9 + kh * 3 + 2] = hi[1];
            }
    } else {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ih = 2 * oh - 1 + kh, iw = 2 * ow - 1 + kw;
                    float v = 0.0f;
                    if (ih >= 0 && iw >= 0 && ih < d.H && iw < d.W)
                        v = Act<T>::ld1(d.x, n * d.sn + ci * d.sc + ih * d.sh + iw * d.sw);
                    xin[ci * 9 + kh * 3 + kw] = v;
                }
    }
    const long long o0 = m * Cout;
    if (own) {
        // all 16 outputs of the pixel are formed in registers and leave as four 16-byte stores (same
        // arithmetic as the general path below).  The 351 weights are the same for every thread: read
        // through the constant address space they arrive by scalar loads and enter the multiply-adds as
        // scalar operands — staged in LDS, the compiler read all of them ahead of the first multiply-add
        // into 351 vector registers, which left the kernel one wave per SIMD.
        typedef const float __attribute__((address_space(4))) *const_f32;
        const const_f32 wc = (const_f32)(unsigned long long)d.w;
        // The per-channel constants of the 16 outputs (epi_channel_affine's arithmetic) are formed once,
        // from vector loads that every lane makes at the same address (an absent vector reads zero): read
        // by scalar loads inside the channel loop they cost each channel a dependent round trip or two.
        float scv[16], shv[16];
        {
            const BnAffine &bn = d.epi.bn;
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(d.epi.bias, 13u * 4u);
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(bn.gamma, bn.on ? 64u : 0u), rbe = make_rsrc(bn.beta, bn.on ? 64u : 0u);
            const __amdgpu_buffer_rsrc_t rmu = make_rsrc(bn.mean, bn.on ? 64u : 0u), rva = make_rsrc(bn.var, bn.on ? 64u : 0u);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = buf_load4(rb, q * 16u), g4 = buf_load4(rg, q * 16u), be4 = buf_load4(rbe, q * 16u);
                const f32x4 mu4 = buf_load4(rmu, q * 16u), va4 = buf_load4(rva, q * 16u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sc = 1.0f, sh = b4[e];
                    if (bn.on) {            // (uniform)
                        const float inv = bn.var_is_invstd ? va4[e] : rsqrtf(va4[e] + bn.eps);
                        const float g = (bn.gamma ? g4[e] : 1.0f) * inv;
                        sh = (sh - mu4[e]) * g + (bn.beta ? be4[e] : 0.0f);
                        sc = g;
                    }
                    scv[q * 4 + e] = sc; shv[q * 4 + e] = sh;
                }
            }
        }
        float v[16];
#pragma unroll
        for (int co = 0; co < 13; ++co) {
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < 27; ++e) acc = fmaf(xin[e], wc[co * 27 + e], acc);
            v[co] = acc * scv[co] + shv[co];
        }
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float a = xin[ci * 9 + 4], b = xin[ci * 9 + 5], c = xin[ci * 9 + 7], e = xin[ci * 9 + 8];
            float mx = a;
            if (b > mx) mx = b;
            if (c > mx) mx = c;
            if (e > mx) mx = e;
            v[13 + ci] = mx * scv[13 + ci] + shv[13 + ci];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 st;
#pragma unroll
            for (int e = 0; e < 4; ++e) st[e] = d.epi.relu ? fmaxf(v[q * 4 + e], 0.0f) : v[q * 4 + e];
            Act<T>::st4(d.out, o0 + q * 4, st);
        }
        return;
    }
    for (int co = 0; co < d.Cc; ++co) {
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < 27; ++e) acc = fmaf(xin[e], ws[co * 28 + e], acc);
        float sc, sh;
        epi_channel_affine(d.epi, co, sc, sh);
        Act<T>::st1(d.out, o0 + co, epi_finish<T>(d.epi, acc * sc + sh, m, n, co, Cout));
    }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        // window (2oh..2oh+1, 2ow..2ow+1) = taps kh,kw in {1,2}
        const float a = xin[ci * 9 + 4], b = xin[ci * 9 + 5], c = xin[ci * 9 + 7], e = xin[ci * 9 + 8];
        float mx = a;
        if (b > mx) mx = b;
        if (c > mx) mx = c;
        if (e > mx) mx = e;
        const int co = d.Cc + ci;
        Epilogue ep = d.epi;
        ep.bias = nullptr;
        float sc, sh;
        epi_channel_affine(ep, co, sc, sh);
        Act<T>::st1(d.out, o0 + co, epi_finish<T>(d.epi, mx * sc + sh, m, n, co, Cout));
    }
}

// Weight/bias gradient of the above: dW[co][e] = sum_pix g[pix][co] * patch[pix][e] over the block's
// pixel slab (Cc <= 16).
struct FirstDownWgradDesc {
    const void *x;       // storage type T
    long long sn, sc, sh, sw;
    const void *g;       // [M][g_C] gradient wrt the pre-activation concat, conv channels first (T)
    float *partial;      // [nsplit][Cc*27 + Cc]
    int N, H, W, Cc, g_C;
    int split_pixels;
    unsigned x_elems;    // extent of x covered by its strides
};

// MFMA 16x16x4: row = conv channel co, contraction = 4 consecutive output pixels, column = patch
// element e (27 taps*channels, column 27 = 1 for the bias; two column tiles).  Lane (i, g) reads
// g[pix g][co = i] and the two patch elements e = i, 16 + i of pixel g straight from global memory
// (buffer loads: out-of-image taps, pixels beyond the slab and padding columns read zero).  Each
// wave covers a quarter of the block's pixel slab; the four waves are summed through LDS.
template <typename AT>
__global__ __launch_bounds__(256) void first_down_wgrad_kernel(const FirstDownWgradDesc d)
{
    __shared__ float red[4][16 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const unsigned Ho = d.H / 2, Wo = d.W / 2, HWo = Ho * Wo, M = (unsigned)d.N * HWo;
    const unsigned wsp = (unsigned)d.split_pixels / 4;                 // multiple of 4 (host)
    const unsigned wbeg = blockIdx.x * (unsigned)d.split_pixels + wave * wsp;
    unsigned wend = wbeg + wsp;
    if (wend > M) wend = M;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.x, d.x_elems * Act<AT>::B);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, M * d.g_C * Act<AT>::B);
    // this lane's two patch elements: e -> (ci, kh, kw); e = 27 is the bias column, e > 27 padding
    int eci[2], ekh[2], ekw[2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int e = 16 * T + i;
        eci[T] = e / 9; ekh[T] = (e % 9) / 3; ekw[T] = e % 3;
    }
    const bool bias_lane = i == 11;            // e = 27 in the second tile
    f32x4 acc[2];
    acc[0] = acc[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    struct Frag { float gv, pv[2]; };
    auto load_group = [&](unsigned m0, Frag &f) {
        const unsigned m = m0 + g;
        const bool valid = m < wend;
        const unsigned mm = valid ? m : 0;
        const unsigned n = mm / HWo, rem = mm - n * HWo;
        const unsigned oh = rem / Wo, ow = rem - oh * Wo;
        f.gv = Act<AT>::load1(r_g, (valid && i < d.Cc) ? (mm * d.g_C + i) : ERF_OOB);
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            const int ih = 2 * (int)oh - 1 + ekh[T], iw = 2 * (int)ow - 1 + ekw[T];
            const bool ok = valid && 16 * T + i < 27 && ih >= 0 && iw >= 0 && ih < d.H && iw < d.W;
            const long long off = (long long)n * d.sn + eci[T] * d.sc + ih * d.sh + iw * d.sw;
            f.pv[T] = Act<AT>::load1(r_x, ok ? (unsigned)off : ERF_OOB);
        }
        if (bias_lane) f.pv[1] = valid ? 1.0f : 0.0f;
    };
    auto mfma_group = [&](const Frag &f) {
#pragma unroll
        for (int T = 0; T < 2; ++T) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.gv, f.pv[T], acc[T], 0, 0, 0);
    };
    // four groups in flight
    for (unsigned m0 = wbeg; m0 < wend; m0 += 16) {
        Frag f[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) load_group(m0 + 4 * u, f[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) mfma_group(f[u]);
    }
    // D[row = co = 4g + r][col = e = 16T + i]
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * g + r) * 32 + 16 * T + i] = acc[T][r];
    __syncthreads();
    float *P = d.partial + (long long)blockIdx.x * (d.Cc * 27 + d.Cc);
    for (int idx = tid; idx < 16 * 32; idx += 256) {
        const int co = idx >> 5, e = idx & 31;
        if (co >= d.Cc || e > 27) continue;
        const float v = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
        if (e < 27) P[co * 27 + e] = v;
        else P[d.Cc * 27 + co] = v;
    }
}

// out[i] = sum_s partial[s][i].  Block = 32 elements x 8 slab groups; the 8 group sums are added in
// a fixed order through LDS (bitwise reproducible).
__global__ __launch_bounds__(256) void reduce_flat_kernel(const float *partial, float *dw, float *db,
                                                          int nsplit, int nw, int nb)
{
    __shared__ float red[8][33];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    float s = 0.0f;
    if (idx < nw + nb) {
        // (eight slabs per trip, requested together, added in slab order: see outconv_wreduce_kernel)
        int i = g;
        for (; i + 56 < nsplit; i += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(i + 8 * u) * (nw + nb) + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < nsplit; i += 8) s += partial[(long long)i * (nw + nb) + idx];
    }
    red[g][e] = s;
    __syncthreads();
    if (g != 0 || idx >= nw + nb) return;
    float tot = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) tot += red[k][e];
    if (idx < nw) dw[idx] = tot;
    else if (db) db[idx - nw] = tot;
}

// ------------------------------------------------------------------------------------------------
// 2x2 stride-2 max-pool of in[N][H][W][C] into channels [out_coff, out_coff+C) of
// out[N][H/2][W/2][out_C], with the shared epilogue.  One thread per (output pixel, 4 channels).
struct PoolDesc {
    const void *in;      // storage type T
    void *out;
    Epilogue epi;
    int N, H, W, C, out_C, out_coff;
};

template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const PoolDesc d)
{
    const int Ho = d.H / 2, Wo = d.W / 2, C4 = d.C / 4;
    const long long total = (long long)d.N * Ho * Wo * C4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C4) * 4;
    const long long m = idx / C4;
    const int n = (int)(m / (Ho * Wo));
    const int rem = (int)(m - (long long)n * Ho * Wo);
    const int oh = rem / Wo, ow = rem - oh * Wo;
    const long long p = (((long long)n * d.H + 2 * oh) * d.W + 2 * ow) * d.C + c;
    const f32x4 v00 = Act<T>::ld4(d.in, p);
    const f32x4 v01 = Act<T>::ld4(d.in, p + d.C);
    const f32x4 v10 = Act<T>::ld4(d.in, p + (long long)d.W * d.C);
    const f32x4 v11 = Act<T>::ld4(d.in, p + (long long)d.W * d.C + d.C);
    f32x4 o;
    Epilogue ep = d.epi;
    ep.bias = nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float mx = v00[e];
        if (v01[e] > mx) mx = v01[e];
        if (v10[e] > mx) mx = v10[e];
        if (v11[e] > mx) mx = v11[e];
        const int co = d.out_coff + c + e;
        float sc, sh;
        epi_channel_affine(ep, co, sc, sh);
        o[e] = epi_finish<T>(d.epi, mx * sc + sh, m, n, co, d.out_C);
    }
    Act<T>::st4(d.out, m * d.out_C + d.out_coff + c, o);
}

// Max-pool backward: dx[n][2oh+a][2ow+b][c] = (first maximum of the window in scan order) ?
// g[n][oh][ow][g_coff+c] : 0.  Writes every element of dx.
struct PoolBwdDesc {
    const void *in, *g;  // storage type T
    void *dx;
    int N, H, W, C, g_C, g_coff;
};

template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const PoolBwdDesc d)
{
    const int Ho = d.H / 2, Wo = d.W / 2;
    const long long total = (long long)d.N * Ho * Wo * d.C;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % d.C);
    const long long m = idx / d.C;
    const int n = (int)(m / (Ho * Wo));
    const int rem = (int)(m - (long long)n * Ho * Wo);
    const int oh = rem / Wo, ow = rem - oh * Wo;
    const long long base = (((long long)n * d.H + 2 * oh) * d.W + 2 * ow) * d.C + c;
    const long long o01 = d.C, o10 = (long long)d.W * d.C, o11 = o10 + d.C;
    const float v00 = Act<T>::ld1(d.in, base), v01 = Act<T>::ld1(d.in, base + o01), v10 = Act<T>::ld1(d.in, base + o10),
                v11 = Act<T>::ld1(d.in, base + o11);
    int best = 0;
    float mx = v00;
    if (v01 > mx) { mx = v01; best = 1; }
    if (v10 > mx) { mx = v10; best = 2; }
    if (v11 > mx) { mx = v11; best = 3; }
    const float gv = Act<T>::ld1(d.g, m * d.g_C + d.g_coff + c);
    Act<T>::st1(d.dx, base, best == 0 ? gv : 0.0f);
    Act<T>::st1(d.dx, base + o01, best == 1 ? gv : 0.0f);
    Act<T>::st1(d.dx, base + o10, best == 2 ? gv : 0.0f);
    Act<T>::st1(d.dx, base + o11, best == 3 ? gv : 0.0f);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm.  Statistics are accumulated per thread in fp32 over a short run of pixels, combined in
// LDS and added to global double accumulators (sums[0..C) = sum x, sums[C..2C) = sum x^2 or the
// second moment named below).
// bn_stats_kernel: x [M][C].   bn_bwd_stats_kernel: sum g and sum g*xhat with g = gy*(mask>0)*nscale.
struct BnStatsDesc {
    const void *x;         // pre-BN tensor [M][C]            (x, gy, mask: storage type T)
    const void *gy;        // backward only
    const void *mask;      // backward only, may be NULL: gy counted where mask > 0
    const float *nscale;   // backward only, may be NULL: [N][C]
    const float *mean, *invstd;
    double *sums;          // [2C], zeroed by the caller
    long long M;
    int C, pix_per_image, rows_per_block;
};

template <bool BWD, typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const BnStatsDesc d)
{
    __shared__ float red[2][256];
    const int C = d.C;
    const int tpr = 256 / C > 0 ? 256 / C : 1;       // threads sharing a channel (C <= 256)
    const int c = threadIdx.x % C, rr = threadIdx.x / C;
    float s0 = 0.0f, s1 = 0.0f;
    if (rr < tpr) {
        const long long r0 = (long long)blockIdx.x * d.rows_per_block;
        long long r1 = r0 + d.rows_per_block;
        if (r1 > d.M) r1 = d.M;
        const float mean = BWD ? d.mean[c] : 0.0f, inv = BWD ? d.invstd[c] : 0.0f;
        for (long long r = r0 + rr; r < r1; r += tpr) {
            const float xv = Act<T>::ld1(d.x, r * C + c);
            if (BWD) {
                float g = Act<T>::ld1(d.gy, r * C + c);
                if (d.mask && !(Act<T>::ld1(d.mask, r * C + c) > 0.0f)) g = 0.0f;
                if (d.nscale) g *= d.nscale[(r / d.pix_per_image) * C + c];
                s0 += g;
                s1 += g * (xv - mean) * inv;
            } else {
                s0 += xv;
                s1 += xv * xv;
            }
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < C) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < tpr; ++k) { a += red[0][threadIdx.x + k * C]; b += red[1][threadIdx.x + k * C]; }
        atomicAdd(&d.sums[threadIdx.x], a);
        atomicAdd(&d.sums[C + threadIdx.x], b);
    }
}

// Vectorised statistics for C % 4 == 0: a thread owns four channels (16-byte loads) and walks rows
// b*RP + rr, b*RP + rr + G*RP, ... (RP = 1024 / C rows per block step, G blocks); a block's sums go,
// as doubles, to its own row of `partial` [G][2C] — no atomics, so the result is bitwise
// reproducible once bn_reduce_kernel has added the rows in a fixed order.
struct BnStats4Desc {
    BnStatsDesc s;
    double *partial;       // [gridDim.x][2C]
};

template <bool BWD, typename T>
__global__ __launch_bounds__(256) void bn_stats4_kernel(const BnStats4Desc dd)
{
    const BnStatsDesc &d = dd.s;
    __shared__ float red[2][1024];
    const int C = d.C, C4 = C / 4;
    const int RP = 256 / C4;                          // rows per block step (C <= 256: RP >= 4)
    const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
    f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (rr < RP) {
        f32x4 mean = s0, inv = s0;
        if (BWD) {
            mean = *reinterpret_cast<const f32x4 *>(d.mean + 4 * cq);
            inv = *reinterpret_cast<const f32x4 *>(d.invstd + 4 * cq);
        }
        // the rows of a trip are loaded together (several loads in flight per lane) and added in row
        // order, so the sums equal those of a row-by-row walk bit for bit
        const bool has_mask = BWD && d.mask != nullptr, has_ns = BWD && d.nscale != nullptr;
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        auto add_row = [&](f32x4 xv, f32x4 g, f32x4 mk, f32x4 ns) __attribute__((always_inline)) {
            if (BWD) {
                if (has_mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = mk[e] > 0.0f ? g[e] : 0.0f;
                }
                if (has_ns) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] *= ns[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] += g[e]; s1[e] += g[e] * (xv[e] - mean[e]) * inv[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] += xv[e]; s1[e] += xv[e] * xv[e]; }
            }
        };
        constexpr int SU = 4;
        const long long stride = (long long)gridDim.x * RP;
        long long r = (long long)blockIdx.x * RP + rr;
        for (; r + (SU - 1) * stride < d.M; r += SU * stride) {
            f32x4 xv[SU], g[SU], mk[SU], ns[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) xv[u] = Act<T>::ld4(d.x, (r + u * stride) * C + 4 * cq);
#pragma unroll
            for (int u = 0; u < SU; ++u) g[u] = BWD ? Act<T>::ld4(d.gy, (r + u * stride) * C + 4 * cq) : zero4;
#pragma unroll
            for (int u = 0; u < SU; ++u) mk[u] = has_mask ? Act<T>::ld4(d.mask, (r + u * stride) * C + 4 * cq) : zero4;
#pragma unroll
            for (int u = 0; u < SU; ++u)
                ns[u] = has_ns ? *reinterpret_cast<const f32x4 *>(d.nscale + ((r + u * stride) / d.pix_per_image) * C + 4 * cq) : zero4;
#pragma unroll
            for (int u = 0; u < SU; ++u) add_row(xv[u], g[u], mk[u], ns[u]);
        }
        for (; r < d.M; r += stride) {
            const f32x4 g = BWD ? Act<T>::ld4(d.gy, r * C + 4 * cq) : zero4;
            const f32x4 mk = has_mask ? Act<T>::ld4(d.mask, r * C + 4 * cq) : zero4;
            const f32x4 ns = has_ns ? *reinterpret_cast<const f32x4 *>(d.nscale + (r / d.pix_per_image) * C + 4 * cq) : zero4;
            add_row(Act<T>::ld4(d.x, r * C + 4 * cq), g, mk, ns);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][rr * C + 4 * cq + e] = s0[e];
            red[1][rr * C + 4 * cq + e] = s1[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        double a = 0.0;
        for (int k = 0; k < RP; ++k) a += red[which][k * C + c];
        dd.partial[(long long)blockIdx.x * 2 * C + i] = a;
    }
}

// sums[i] = sum over the G rows of partial[.][i], i < n (= 2C): a block takes 8 elements x 32 row
// groups (thread (g, e) adds rows g, g+32, ...), the group sums are then added in a fixed order.
__global__ __launch_bounds__(256) void bn_reduce_kernel(const double *partial, double *sums, int G, int n)
{
    __shared__ double red[32][9];
    const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int idx = blockIdx.x * 8 + e;
    double a = 0.0;
    if (idx < n) {
#pragma unroll 4
        for (int i = g; i < G; i += 32) a += partial[(long long)i * n + idx];
    }
    red[g][e] = a;
    __syncthreads();
    if (g != 0 || idx >= n) return;
    double tot = red[0][e];
#pragma unroll
    for (int k = 1; k < 32; ++k) tot += red[k][e];
    sums[idx] = tot;
}

// mean / invstd from the sums; running statistics updated as nn.BatchNorm2d does in training
// (momentum, unbiased variance, step counter).
__device__ __forceinline__ void bn_finalize_channel(double sum, double sumsq, int c, float *save_mean, float *save_invstd,
                                                    float *run_mean, float *run_var, long long M, float momentum,
                                                    float eps)
{
    const double mean = sum / (double)M;
    double var = sumsq / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
    if (run_var) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
}

__global__ void bn_finalize_kernel(const double *sums, float *save_mean, float *save_invstd,
                                   float *run_mean, float *run_var, long long *nbt, long long M, int C,
                                   float momentum, float eps)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    bn_finalize_channel(sums[c], sums[C + c], c, save_mean, save_invstd, run_mean, run_var, M, momentum, eps);
}

// bn_reduce_kernel for both sums of a channel followed by the finalisation, in one launch: a block
// takes 8 channels x 32 row groups; the additions run in bn_reduce_kernel's order.
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const double *partial, int G, float *save_mean,
                                                                 float *save_invstd, float *run_mean, float *run_var,
                                                                 long long *nbt, long long M, int C, float momentum,
                                                                 float eps)
{
    __shared__ double red[2][32][9];
    const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + e;
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int i = g; i < G; i += 32) {
            a += partial[(long long)i * 2 * C + c];
            b += partial[(long long)i * 2 * C + C + c];
        }
    }
    red[0][g][e] = a;
    red[1][g][e] = b;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
    if (g != 0 || c >= C) return;
    double s0 = red[0][0][e], s1 = red[1][0][e];
#pragma unroll
    for (int k = 1; k < 32; ++k) { s0 += red[0][k][e]; s1 += red[1][k][e]; }
    bn_finalize_channel(s0, s1, c, save_mean, save_invstd, run_mean, run_var, M, momentum, eps);
}

// Eval mode needs mean/invstd in the same form for backward.
__global__ void bn_eval_prepare_kernel(const float *run_mean, const float *run_var, float *save_mean,
                                       float *save_invstd, int C, float eps)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save_mean[c] = run_mean[c];
    save_invstd[c] = 1.0f / sqrtf(run_var[c] + eps);
}

// Element-wise pass over [M][C] (float4): out = epilogue(in).  Used for BatchNorm apply (+dropout
// scale, residual, ReLU) and for gradient masking.
struct EltDesc {
    const void *in;        // storage type T
    void *out;
    Epilogue epi;
    long long M;
    int C, pix_per_image;
};

template <typename T>
__global__ __launch_bounds__(256) void elt_kernel(const EltDesc d)
{
    const int C4 = d.C / 4;
    const long long total = d.M * C4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C4) * 4;
        const long long m = idx / C4;
        const int n = (int)(m / d.pix_per_image);
        const f32x4 v = Act<T>::ld4(d.in, m * d.C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sc, sh;
            epi_channel_affine(d.epi, c + e, sc, sh);
            o[e] = epi_finish<T>(d.epi, v[e] * sc + sh, m, n, c + e, d.C);
        }
        Act<T>::st4(d.out, m * d.C + c, o);
    }
}

// The same pass for channel counts whose float4 columns divide the block (256 % (C/4) == 0) and
// 16-byte addressable optional tensors: a thread keeps one channel quad, so the per-channel affine
// is formed once, rows advance without any division, and every tensor moves as 16-byte accesses.
#define ROWS_U 4
template <typename T>
__global__ __launch_bounds__(256) void elt_rows_kernel(const EltDesc d)
{
    const int C = d.C, C4 = C / 4, RP = 256 / C4;
    const int c = (threadIdx.x % C4) * 4, rr = threadIdx.x / C4;
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) epi_channel_affine(d.epi, c + e, sc[e], sh[e]);
    const Epilogue &ep = d.epi;
    const bool has_res = ep.res != nullptr, has_mask = ep.mask != nullptr, has_ns = ep.nscale != nullptr;
    const bool relu = ep.relu != 0;
    // (host: C % 32 == 0 when given) the output's "> 0" bits, word [row][c / 32], bit c % 32: a thread
    // holds four of them, eight consecutive lanes a word
    unsigned *const bits_out = ep.bits_out;
    // ROWS_U rows per trip with every load issued before the first use: a wave keeps ROWS_U 16-byte
    // loads per tensor in flight (a single one left the memory system under-subscribed: 2.9 TB/s)
    auto finish = [&](long long r, f32x4 v, f32x4 ns, f32x4 rs, f32x4 mk) __attribute__((always_inline)) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] * sc[e] + sh[e];
        if (has_ns) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= ns[e];
        }
        if (has_res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += rs[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.0f);
        }
        if (has_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = mk[e] > 0.0f ? o[e] : 0.0f;
        }
        Act<T>::st4(d.out, r * C + c, o);
        if (bits_out) {
            unsigned w = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) w |= (o[e] > 0.0f ? 1u : 0u) << e;
            w <<= (unsigned)(c & 31);
            w |= (unsigned)__shfl_xor((int)w, 1);
            w |= (unsigned)__shfl_xor((int)w, 2);
            w |= (unsigned)__shfl_xor((int)w, 4);
            if ((threadIdx.x & 7) == 0) bits_out[r * (C / 32) + (c >> 5)] = w;
        }
    };
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const long long stride = (long long)gridDim.x * RP;
    long long r = (long long)blockIdx.x * RP + rr;
    for (; r + (ROWS_U - 1) * stride < d.M; r += ROWS_U * stride) {
        f32x4 v[ROWS_U], ns[ROWS_U], rs[ROWS_U], mk[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) v[u] = Act<T>::ld4(d.in, (r + u * stride) * C + c);
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            rs[u] = has_res ? Act<T>::ld4(ep.res, (r + u * stride) * ep.res_C + ep.res_coff + c) : zero4;
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            mk[u] = has_mask ? Act<T>::ld4(ep.mask, (r + u * stride) * ep.mask_C + ep.mask_coff + c) : zero4;
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            ns[u] = has_ns ? *reinterpret_cast<const f32x4 *>(ep.nscale + ((r + u * stride) / d.pix_per_image) * C + c) : zero4;
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) finish(r + u * stride, v[u], ns[u], rs[u], mk[u]);
    }
    for (; r < d.M; r += stride) {
        const f32x4 v = Act<T>::ld4(d.in, r * C + c);
        const f32x4 rs = has_res ? Act<T>::ld4(ep.res, r * ep.res_C + ep.res_coff + c) : zero4;
        const f32x4 mk = has_mask ? Act<T>::ld4(ep.mask, r * ep.mask_C + ep.mask_coff + c) : zero4;
        const f32x4 ns = has_ns ? *reinterpret_cast<const f32x4 *>(ep.nscale + (r / d.pix_per_image) * C + c) : zero4;
        finish(r, v, ns, rs, mk);
    }
}

// BatchNorm backward apply: with g = gy*(mask>0)*nscale,
//   train: dx = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M);  eval: dx = gamma*invstd*g.
// dgamma = sum_gx, dbeta = sum_g are written by block 0.
struct BnBwdDesc {
    const void *x, *gy, *mask;      // storage type T
    const float *nscale, *gamma, *mean, *invstd;
    const double *sums;
    void *dx;                       // storage type T
    float *dgamma, *dbeta;
    long long M;
    int C, pix_per_image, training;
};

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdDesc d)
{
    const int C4 = d.C / 4;
    const long long total = d.M * C4;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < d.C; c += 256) {
            if (d.dbeta) d.dbeta[c] = (float)d.sums[c];
            if (d.dgamma) d.dgamma[c] = (float)d.sums[d.C + c];
        }
    const double invM = 1.0 / (double)d.M;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C4) * 4;
        const long long m = idx / C4;
        const f32x4 xv = Act<T>::ld4(d.x, m * d.C + c);
        f32x4 g = Act<T>::ld4(d.gy, m * d.C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = c + e;
            float gv = g[e];
            if (d.mask && !(Act<T>::ld1(d.mask, m * d.C + ch) > 0.0f)) gv = 0.0f;
            if (d.nscale) gv *= d.nscale[(m / d.pix_per_image) * d.C + ch];
            const float gi = (d.gamma ? d.gamma[ch] : 1.0f) * d.invstd[ch];
            if (d.training) {
                const float xh = (xv[e] - d.mean[ch]) * d.invstd[ch];
                const float mg = (float)(d.sums[ch] * invM), mgx = (float)(d.sums[d.C + ch] * invM);
                o[e] = gi * (gv - mg - xh * mgx);
            } else {
                o[e] = gi * gv;
            }
        }
        Act<T>::st4(d.dx, m * d.C + c, o);
    }
}

// Row-structured variant (256 % (C/4) == 0, 16-byte aligned tensors): per-channel constants once
// per thread, no divisions, 16-byte accesses for the mask and the dropout factors as well.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const BnBwdDesc d)
{
    const int C = d.C, C4 = C / 4, RP = 256 / C4;
    const int c = (threadIdx.x % C4) * 4, rr = threadIdx.x / C4;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < C; k += 256) {
            if (d.dbeta) d.dbeta[k] = (float)d.sums[k];
            if (d.dgamma) d.dgamma[k] = (float)d.sums[C + k];
        }
    const double invM = 1.0 / (double)d.M;
    float gi[4], mean[4], inv[4], mg[4], mgx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = c + e;
        inv[e] = d.invstd[ch];
        mean[e] = d.mean[ch];
        gi[e] = (d.gamma ? d.gamma[ch] : 1.0f) * inv[e];
        mg[e] = (float)(d.sums[ch] * invM);
        mgx[e] = (float)(d.sums[C + ch] * invM);
    }
    const bool has_mask = d.mask != nullptr, has_ns = d.nscale != nullptr, training = d.training != 0;
    auto finish = [&](long long r, f32x4 xv, f32x4 g, f32x4 mk, f32x4 ns) __attribute__((always_inline)) {
        if (has_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = mk[e] > 0.0f ? g[e] : 0.0f;
        }
        if (has_ns) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= ns[e];
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (training) {
                const float xh = (xv[e] - mean[e]) * inv[e];
                o[e] = gi[e] * (g[e] - mg[e] - xh * mgx[e]);
            } else {
                o[e] = gi[e] * g[e];
            }
        }
        Act<T>::st4(d.dx, r * C + c, o);
    };
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const long long stride = (long long)gridDim.x * RP;
    long long r = (long long)blockIdx.x * RP + rr;
    for (; r + (ROWS_U - 1) * stride < d.M; r += ROWS_U * stride) {       // all loads of ROWS_U rows first
        f32x4 xv[ROWS_U], g[ROWS_U], mk[ROWS_U], ns[ROWS_U];
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) xv[u] = Act<T>::ld4(d.x, (r + u * stride) * C + c);
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) g[u] = Act<T>::ld4(d.gy, (r + u * stride) * C + c);
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) mk[u] = has_mask ? Act<T>::ld4(d.mask, (r + u * stride) * C + c) : zero4;
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u)
            ns[u] = has_ns ? *reinterpret_cast<const f32x4 *>(d.nscale + ((r + u * stride) / d.pix_per_image) * C + c) : zero4;
#pragma unroll
        for (int u = 0; u < ROWS_U; ++u) finish(r + u * stride, xv[u], g[u], mk[u], ns[u]);
    }
    for (; r < d.M; r += stride) {
        const f32x4 mk = has_mask ? Act<T>::ld4(d.mask, r * C + c) : zero4;
        const f32x4 ns = has_ns ? *reinterpret_cast<const f32x4 *>(d.nscale + (r / d.pix_per_image) * C + c) : zero4;
        finish(r, Act<T>::ld4(d.x, r * C + c), Act<T>::ld4(d.gy, r * C + c), mk, ns);
    }
}

// ------------------------------------------------------------------------------------------------
// Output ConvTranspose2d(CIN, ncls, 2, stride=2): out[n][o][2i+a][2j+b] = bias[o] + sum_c x[c]*w[c][o][a][b].
// One thread per input pixel; x (NHWC) in registers, weights in LDS as [o][a][b][c]; the thread
// writes float2 {b=0,1} per (o,a) into the NCHW output — consecutive lanes are consecutive j, so a
// wave writes 512 contiguous bytes per store.  SOFTMAX fuses softmax over o (ncls <= 32).
#define OC_CIN 16
#define OC_MAX_NCLS 32
struct OutConvDesc {
    const void *x;         // NHWC, storage type T
    const float *w, *b;
    void *y;               // NCHW logits, storage type TO
    int N, H, W, ncls;
};

template <bool SOFTMAX, typename T, typename TO>
__global__ __launch_bounds__(256) void outconv_fwd_kernel(const OutConvDesc d)
{
    __shared__ __attribute__((aligned(16))) float ws[OC_MAX_NCLS * 4 * OC_CIN];
    __shared__ float bs[OC_MAX_NCLS];
    for (int i = threadIdx.x; i < d.ncls * 4 * OC_CIN; i += 256) {
        const int c = i % OC_CIN, ab = (i / OC_CIN) % 4, o = i / (OC_CIN * 4);
        ws[i] = d.w[(c * d.ncls + o) * 4 + ab];
    }
    for (int i = threadIdx.x; i < d.ncls; i += 256) bs[i] = d.b ? d.b[i] : 0.0f;
    __syncthreads();
    const long long M = (long long)d.N * d.H * d.W;
    // grid-stride over the input pixels: the weight tile in LDS is filled once per block
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    const int n = (int)(m / (d.H * d.W));
    const int rem = (int)(m - (long long)n * d.H * d.W);
    const int i = rem / d.W, j = rem - i * d.W;
    float xv[OC_CIN];
#pragma unroll
    for (int q = 0; q < OC_CIN / 4; ++q) {
        const f32x4 v = Act<T>::ld4(d.x, m * OC_CIN + q * 4);
        xv[q * 4] = v[0]; xv[q * 4 + 1] = v[1]; xv[q * 4 + 2] = v[2]; xv[q * 4 + 3] = v[3];
    }
    const int Ho = 2 * d.H, Wo = 2 * d.W;
    if (!SOFTMAX) {
        for (int o = 0; o < d.ncls; ++o) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float r0 = bs[o], r1 = bs[o];
                const float *w0 = &ws[((o * 2 + a) * 2 + 0) * OC_CIN];
                const float *w1 = &ws[((o * 2 + a) * 2 + 1) * OC_CIN];
#pragma unroll
                for (int c = 0; c < OC_CIN; ++c) { r0 = fmaf(xv[c], w0[c], r0); r1 = fmaf(xv[c], w1[c], r1); }
                f32x2 st = {r0, r1};
                Act<TO>::st2(d.y, (((long long)n * d.ncls + o) * Ho + 2 * i + a) * Wo + 2 * j, st);
            }
        }
    } else {
        // two passes over the classes per output position: max / sum of exp, then normalise
        float mx[4], sum[4];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) { mx[ab] = -INFINITY; sum[ab] = 0.0f; }
        for (int o = 0; o < d.ncls; ++o)
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                float r = bs[o];
                const float *w0 = &ws[(o * 4 + ab) * OC_CIN];
#pragma unroll
                for (int c = 0; c < OC_CIN; ++c) r = fmaf(xv[c], w0[c], r);
                mx[ab] = fmaxf(mx[ab], r);
            }
        for (int o = 0; o < d.ncls; ++o)
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                float r = bs[o];
                const float *w0 = &ws[(o * 4 + ab) * OC_CIN];
#pragma unroll
                for (int c = 0; c < OC_CIN; ++c) r = fmaf(xv[c], w0[c], r);
                sum[ab] += expf(r - mx[ab]);
            }
        for (int o = 0; o < d.ncls; ++o)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float r[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float v = bs[o];
                    const float *w0 = &ws[((o * 2 + a) * 2 + b) * OC_CIN];
#pragma unroll
                    for (int c = 0; c < OC_CIN; ++c) v = fmaf(xv[c], w0[c], v);
                    r[b] = expf(v - mx[a * 2 + b]) / sum[a * 2 + b];
                }
                f32x2 st = {r[0], r[1]};
                Act<TO>::st2(d.y, (((long long)n * d.ncls + o) * Ho + 2 * i + a) * Wo + 2 * j, st);
            }
    }
    }
}

// Class index of the largest logit per output position, written as int64 [N][1][2H][2W]: the
// `outputs.max(1)[1]` of the evaluation scripts without the logits ever reaching memory.  The
// logits are formed exactly as in outconv_fwd_kernel (same fma order); ties go to the lowest index.
template <typename T>
__global__ __launch_bounds__(256) void outconv_argmax_kernel(const OutConvDesc d, long long *labels)
{
    __shared__ __attribute__((aligned(16))) float ws[OC_MAX_NCLS * 4 * OC_CIN];
    __shared__ float bs[OC_MAX_NCLS];
    for (int i = threadIdx.x; i < d.ncls * 4 * OC_CIN; i += 256) {
        const int c = i % OC_CIN, ab = (i / OC_CIN) % 4, o = i / (OC_CIN * 4);
        ws[i] = d.w[(c * d.ncls + o) * 4 + ab];
    }
    for (int i = threadIdx.x; i < d.ncls; i += 256) bs[i] = d.b ? d.b[i] : 0.0f;
    __syncthreads();
    const long long M = (long long)d.N * d.H * d.W;
    // grid-stride over the input pixels: the weight tile in LDS is filled once per block
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    const int n = (int)(m / (d.H * d.W));
    const int rem = (int)(m - (long long)n * d.H * d.W);
    const int i = rem / d.W, j = rem - i * d.W;
    float xv[OC_CIN];
#pragma unroll
    for (int q = 0; q < OC_CIN / 4; ++q) {
        const f32x4 v = Act<T>::ld4(d.x, m * OC_CIN + q * 4);
        xv[q * 4] = v[0]; xv[q * 4 + 1] = v[1]; xv[q * 4 + 2] = v[2]; xv[q * 4 + 3] = v[3];
    }
    float best[4];
    int arg[4];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) { best[ab] = -INFINITY; arg[ab] = 0; }
    for (int o = 0; o < d.ncls; ++o)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            float r = bs[o];
            const float *w0 = &ws[(o * 4 + ab) * OC_CIN];
#pragma unroll
            for (int c = 0; c < OC_CIN; ++c) r = fmaf(xv[c], w0[c], r);
            // NaN wins like in torch.max; otherwise the first largest value
            const bool take = r > best[ab] || (r != r && best[ab] == best[ab]);
            best[ab] = take ? r : best[ab];
            arg[ab] = take ? o : arg[ab];
        }
    const int Ho = 2 * d.H, Wo = 2 * d.W;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        long long *dst = labels + ((long long)n * Ho + 2 * i + a) * Wo + 2 * j;
        dst[0] = arg[a * 2 + 0];
        dst[1] = arg[a * 2 + 1];
    }
    }
}

// dx[n][i][j][c] = sum_{o,a,b} gy[n][o][2i+a][2j+b] * w[c][o][a][b]   (gy NCHW, dx NHWC)
struct OutConvBwdDesc {
    const void *x;         // NHWC, storage type T
    const float *w, *gy;   // gy: gradient of the fp32 logits, NCHW
    void *dx;              // NHWC, storage type T
    float *partial;
    int N, H, W, ncls, split_pixels;
    int mask_dx;
};

template <typename T>
__global__ __launch_bounds__(256) void outconv_dgrad_kernel(const OutConvBwdDesc d)
{
    __shared__ __attribute__((aligned(16))) float ws[OC_MAX_NCLS * 4 * OC_CIN];
    for (int i = threadIdx.x; i < d.ncls * 4 * OC_CIN; i += 256) {
        const int c = i % OC_CIN, ab = (i / OC_CIN) % 4, o = i / (OC_CIN * 4);
        ws[i] = d.w[(c * d.ncls + o) * 4 + ab];
    }
    __syncthreads();
    const long long M = (long long)d.N * d.H * d.W;
    // grid-stride over the input pixels: the weight tile in LDS is filled once per block
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
    const int n = (int)(m / (d.H * d.W));
    const int rem = (int)(m - (long long)n * d.H * d.W);
    const int i = rem / d.W, j = rem - i * d.W;
    const int Ho = 2 * d.H, Wo = 2 * d.W;
    float acc[OC_CIN];
#pragma unroll
    for (int c = 0; c < OC_CIN; ++c) acc[c] = 0.0f;
    for (int o = 0; o < d.ncls; ++o)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const f32x2 g = *reinterpret_cast<const f32x2 *>(
                d.gy + (((long long)n * d.ncls + o) * Ho + 2 * i + a) * Wo + 2 * j);
            const float *w0 = &ws[((o * 2 + a) * 2 + 0) * OC_CIN];
            const float *w1 = &ws[((o * 2 + a) * 2 + 1) * OC_CIN];
#pragma unroll
            for (int c = 0; c < OC_CIN; ++c) acc[c] = fmaf(g[0], w0[c], fmaf(g[1], w1[c], acc[c]));
        }
#pragma unroll
    for (int q = 0; q < OC_CIN / 4; ++q) {
        f32x4 v = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
        if (d.mask_dx) {          // gradient handed to a ReLU-ended producer, masked here by x > 0
            const f32x4 xv = Act<T>::ld4(d.x, m * OC_CIN + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = xv[e] > 0.0f ? v[e] : 0.0f;
        }
        Act<T>::st4(d.dx, m * OC_CIN + q * 4, v);
    }
    }
}

// dw[c][o][a][b] = sum_pix x[pix][c]*gy[o][pix,a,b]; db[o] = sum gy.  Thread (oab, cq) accumulates
// 4 input channels for one (o,a,b) over the block's pixel slab from LDS tiles; a 5th "channel
// group" row carries the bias sum.  Slab layout: [ncls*4][16] then [ncls*4] bias partials.
template <typename AT>
__global__ __launch_bounds__(256) void outconv_wgrad_kernel(const OutConvBwdDesc d)
{
    // MFMA 16x16x4: row = input channel c, contraction = 4 consecutive input pixels, column = (o,a,b)
    // in tiles of 16.  Lane (i, g) reads x[pix g][c = i] and, per column tile, the logit gradient
    // gy[n][o][2y+a][2x+b] of pixel g.  A wave covers a quarter of the block's pixel slab.
    constexpr int MAXT = OC_MAX_NCLS * 4 / 16;
    __shared__ float red[4][MAXT * 256 + MAXT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nab = d.ncls * 4, nt = (nab + 15) / 16;
    const unsigned HW = (unsigned)d.H * d.W, M = (unsigned)d.N * HW;
    const unsigned Ho = 2 * d.H, Wo = 2 * d.W;
    const unsigned wsp = (unsigned)d.split_pixels / 4;                 // multiple of 4 (host)
    const unsigned wbeg = blockIdx.x * (unsigned)d.split_pixels + wave * wsp;
    unsigned wend = wbeg + wsp;
    if (wend > M) wend = M;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.x, M * OC_CIN * Act<AT>::B);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.gy, M * 4u * d.ncls * 4u);
    // per column tile: this lane's (o, a, b) as an element offset into one image's gy, or padding
    unsigned coff[MAXT];
#pragma unroll
    for (int T = 0; T < MAXT; ++T) {
        const int oab = 16 * T + i;
        coff[T] = oab < nab ? ((unsigned)(oab >> 2) * Ho + ((oab >> 1) & 1)) * Wo + (oab & 1) : ERF_OOB;
    }
    f32x4 acc[MAXT];
    float bsum[MAXT];
#pragma unroll
    for (int T = 0; T < MAXT; ++T) { acc[T] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; bsum[T] = 0.0f; }

    struct Frag { float xv, gv[MAXT]; };
    auto load_group = [&](unsigned m0, Frag &f) {
        const unsigned m = m0 + g;
        const bool valid = m < wend;
        const unsigned mm = valid ? m : 0;
        const unsigned n = mm / HW, rem = mm - n * HW;
        const unsigned y = rem / d.W, x = rem - y * d.W;
        f.xv = Act<AT>::load1(r_x, valid ? (mm * OC_CIN + i) : ERF_OOB);
        const unsigned base = n * d.ncls * Ho * Wo + 2 * y * Wo + 2 * x;
#pragma unroll
        for (int T = 0; T < MAXT; ++T)
            if (T < nt) f.gv[T] = buf_load1(r_g, (valid && coff[T] != ERF_OOB) ? (base + coff[T]) * 4u : ERF_OOB);
    };
    auto mfma_group = [&](const Frag &f) {
#pragma unroll
        for (int T = 0; T < MAXT; ++T)
            if (T < nt) {
                bsum[T] += f.gv[T];
                acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.xv, f.gv[T], acc[T], 0, 0, 0);
            }
    };
    for (unsigned m0 = wbeg; m0 < wend; m0 += 8) {
        Frag f[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) load_group(m0 + 4 * u, f[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) mfma_group(f[u]);
    }
    // D[row = c = 4g + r][col = oab = 16T + i]; bias sums over the pixel index g
#pragma unroll
    for (int T = 0; T < MAXT; ++T) {
        if (T >= nt) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(16 * T + i) * 16 + 4 * g + r] = acc[T][r];
        float bs = bsum[T];
        bs += __shfl_xor(bs, 16);
        bs += __shfl_xor(bs, 32);
        if (g == 0) red[wave][MAXT * 256 + 16 * T + i] = bs;
    }
    __syncthreads();
    float *P = d.partial + (long long)blockIdx.x * (nab * OC_CIN + nab);
    for (int idx = tid; idx < nab * OC_CIN; idx += 256)
        P[idx] = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
    for (int idx = tid; idx < nab; idx += 256) {
        const int e = MAXT * 256 + idx;
        P[nab * OC_CIN + idx] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    }
}

// dw[c][o][a][b] = sum_s P[s][oab][c];  db[o] = sum_s sum_ab P[s][bias][o*4+ab].
// Block = 32 slab elements x 8 slab groups, fixed summation order.
__global__ __launch_bounds__(256) void outconv_wreduce_kernel(const float *partial, float *dw, float *db,
                                                              int nsplit, int ncls)
{
    __shared__ float red[8][33];
    const int nab = ncls * 4, nw = nab * OC_CIN, slab = nw + nab;
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    float s = 0.0f;
    if (idx < nw) {
        // eight slabs of this group per trip, all requested before the first addition (added in the same
        // order as one by one: the sum is bit-identical, the loads no longer wait for each other)
        int i = g;
        for (; i + 56 < nsplit; i += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(i + 8 * u) * slab + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < nsplit; i += 8) s += partial[(long long)i * slab + idx];
    } else if (idx < nw + ncls) {
        const int o = idx - nw;
        int i = g;
        for (; i + 24 < nsplit; i += 32) {      // four slabs per trip (sixteen loads), same order of additions
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) v[u][ab] = partial[(long long)(i + 8 * u) * slab + nw + o * 4 + ab];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) s += v[u][ab];
        }
        for (; i < nsplit; i += 8)
            for (int ab = 0; ab < 4; ++ab) s += partial[(long long)i * slab + nw + o * 4 + ab];
    }
    red[g][e] = s;
    __syncthreads();
    if (g != 0 || idx >= nw + ncls) return;
    float tot = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) tot += red[k][e];
    if (idx < nw) {
        const int c = idx % OC_CIN, oab = idx / OC_CIN;
        dw[c * nab + oab] = tot;
    } else if (db) {
        db[idx - nw] = tot;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of CrossEntropyLoss(ignore_index) THROUGH the output ConvTranspose2d in one pass: the
// gradient of the logits, (softmax - onehot) * gy / n_labelled, is formed in registers from the
// block input x (the logits are recomputed: 16 x 4*ncls products per input pixel), and feeds the
// data gradient and the weight gradient at once; it never reaches memory.  Separate kernels write
// and read that tensor (N x ncls x 2H x 2W floats, the largest of the network) three times.
//
// All three products run on v_mfma_f32_16x16x4_f32 (lane (i, g) = (lane & 15, lane >> 4) supplies
// A[i][g] and B[g][i], receives D[4g + r][i]).  A wave takes 16 input pixels at a time, pixel = i:
//   logits  D[oab][pixel] = sum_c W[c][oab] x[pixel][c]        k <-> channel 4g + kk, kk = 0..3
//           -> lane holds class o = 4T + g, positions ab = r = 0..3 of tile T
//   G       = (exp(logit - lse) - [o == target]) * scale       in that layout
//   dx      D[c][pixel] = sum_oab W[c][oab] G[oab][pixel]      k <-> oab = 16T + 4g + r: G as held
//   dW      D[c][oab] = sum_pixel x[pixel][c] G[pixel][oab]    k <-> pixel: G and x go through a
//           per-wave LDS tile to put the pixel into the contraction position
// Partial dW / db leave the block in the slab format of outconv_wgrad_kernel.
struct OutConvCeDesc {
    const void *x;              // NHWC [N][H][W][16], storage type T
    const float *w, *b;         // [16][ncls][2][2], [ncls] or null
    const long long *target;    // [N][2H][2W]
    const float *lse;           // [N][2H][2W] log-sum-exp of the logits (saved by the loss forward)
    const float *stats, *gy;    // stats[1] = labelled pixels; gy = gradient of the scalar loss
    void *dx;                   // NHWC like x, or null
    float *partial;
    int N, H, W, ncls, mask_dx;
    long long ignore_index;
    unsigned groups, gpb;       // 16-pixel groups in total / per block
};

template <int NT, typename T>
__global__ __launch_bounds__(256) void outconv_ce_bwd_kernel(const OutConvCeDesc d)
{
    constexpr int PG = NT * 16 + ((NT * 16) % 64 == 0 ? 16 : 0);       // row pitch of the G tile
    __shared__ __attribute__((aligned(16))) float Gs[4][16 * PG];
    __shared__ __attribute__((aligned(16))) float Xs[4][16 * 16];
    __shared__ float red[4][NT * 256 + NT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nab = d.ncls * 4;
    const unsigned HW = (unsigned)d.H * d.W, M = (unsigned)d.N * HW;
    const unsigned Ho = 2 * d.H, Wo = 2 * d.W;

    // weights in the two operand layouts, bias of this lane's classes
    float wa[NT][4], wd[NT][4], bo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int oab_a = 16 * t + i;
        const int o = 4 * t + g;
        bo[t] = (o < d.ncls && d.b) ? d.b[o] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wa[t][e] = oab_a < nab ? d.w[(4 * g + e) * nab + oab_a] : 0.0f;
            wd[t][e] = o < d.ncls ? d.w[i * nab + 16 * t + 4 * g + e] : 0.0f;
        }
    }
    const float scale = d.gy[0] / d.stats[1];
    // bf16 tensors: the three products on v_mfma_f32_16x16x32_bf16 (16 cycles for 32 contraction slots,
    // against 32 cycles for 4 on the fp32 instruction).  Lane (i, g) supplies slots 8g .. 8g+7 of both
    // operands; which contraction element a slot stands for is free as long as both operands agree, so
    // the slots follow what the lane already holds: channel 4g + j for the logits (j < 4, the upper four
    // slots are zero), class tile 2S + j/4, position j%4 of the lane's class for the data gradient (the
    // gradient of the logits as it sits in the accumulators), pixel 4j + g for the weight gradient.  The
    // weights enter the logits as two bf16 terms (hi + mid: 2^-17 relative, so that exp(logit - lse) agrees
    // with the forward's log-sum-exp) and the data gradient as one; the gradient of the logits is rounded
    // to bf16 on its way into the two products, as every activation gradient of the bf16 path is.
    constexpr bool B16 = sizeof(T) == 2;
    constexpr int NS = (NT + 1) / 2;
    auto pack2 = [](float a, float b) __attribute__((always_inline)) {
        return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
    };
    u32x4 wl_hi[NT], wl_mid[NT], wdx[NS];
    if (B16) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float hi[4], mid[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = bf16_to_f32(f32_to_bf16(wa[t][e]));
                mid[e] = wa[t][e] - hi[e];
            }
            wl_hi[t] = u32x4{pack2(hi[0], hi[1]), pack2(hi[2], hi[3]), 0u, 0u};
            wl_mid[t] = u32x4{pack2(mid[0], mid[1]), pack2(mid[2], mid[3]), 0u, 0u};
        }
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            const bool two = 2 * S + 1 < NT;
            const int t1 = two ? 2 * S + 1 : 2 * S;
            wdx[S] = u32x4{pack2(wd[2 * S][0], wd[2 * S][1]), pack2(wd[2 * S][2], wd[2 * S][3]),
                           two ? pack2(wd[t1][0], wd[t1][1]) : 0u, two ? pack2(wd[t1][2], wd[t1][3]) : 0u};
        }
    }
    auto mfma32 = [](u32x4 a, u32x4 b, f32x4 c) __attribute__((always_inline)) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };

    f32x4 accw[NT];
    float bsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { accw[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; bsum[t] = 0.0f; }
    float *gs = Gs[wave], *xs = Xs[wave];

    const unsigned g0 = blockIdx.x * d.gpb;
    unsigned g1 = g0 + d.gpb;
    if (g1 > d.groups) g1 = d.groups;
    for (unsigned grp = g0 + wave; grp < g1; grp += 4) {
        const unsigned m = grp * 16 + i;
        const bool valid = m < M;
        const unsigned mm = valid ? m : 0;
        const unsigned n = mm / HW, rem = mm - n * HW;
        const unsigned y = rem / d.W, x = rem - y * d.W;
        f32x4 xq = {0.0f, 0.0f, 0.0f, 0.0f};
        u32x4 xb = {0u, 0u, 0u, 0u};            // bf16: the lane's four channels as they sit in memory
        if (B16) {
            if (valid) {
                const u32x2 xr = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const bf16_t *>(d.x) + (long long)mm * OC_CIN + 4 * g);
                xb[0] = xr[0]; xb[1] = xr[1];
            }
            xq = f32x4{__builtin_bit_cast(float, xb[0] << 16), __builtin_bit_cast(float, xb[0] & 0xffff0000u),
                       __builtin_bit_cast(float, xb[1] << 16), __builtin_bit_cast(float, xb[1] & 0xffff0000u)};
        } else if (valid) xq = Act<T>::ld4(d.x, (long long)mm * OC_CIN + 4 * g);
        const long long ob = ((long long)n * Ho + 2 * y) * Wo + 2 * x;
        float lse[4];
        long long tg[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const f32x2 l2 = *reinterpret_cast<const f32x2 *>(d.lse + ob + (long long)a * Wo);
            lse[2 * a] = l2[0]; lse[2 * a + 1] = l2[1];
            tg[2 * a] = d.target[ob + (long long)a * Wo];
            tg[2 * a + 1] = d.target[ob + (long long)a * Wo + 1];
        }
        // logits of classes 4T + g at the four output positions of pixel i
        f32x4 lg[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) lg[t] = f32x4{bo[t], bo[t], bo[t], bo[t]};
        if (B16) {
#pragma unroll
            for (int t = 0; t < NT; ++t) lg[t] = mfma32(wl_mid[t], xb, lg[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) lg[t] = mfma32(wl_hi[t], xb, lg[t]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = 0; t < NT; ++t) lg[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][kk], xq[kk], lg[t], 0, 0, 0);
        }
        // gradient of the logits
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool lab = valid && tg[r] != d.ignore_index && tg[r] >= 0 && tg[r] < d.ncls;
            const float s = lab ? scale : 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int o = 4 * t + g;
                const float p = __expf(lg[t][r] - lse[r]) - ((long long)o == tg[r] ? 1.0f : 0.0f);
                lg[t][r] = (lab && o < d.ncls) ? p * s : 0.0f;
            }
        }
        // data gradient (two accumulators: the dependent latency of the instruction exceeds its issue time)
        if (d.dx) {
            f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
            if (B16) {
#pragma unroll
                for (int S = 0; S < NS; ++S) {
                    const bool two = 2 * S + 1 < NT;
                    const int t1 = two ? 2 * S + 1 : 2 * S;
                    const u32x4 gop = u32x4{pack2(lg[2 * S][0], lg[2 * S][1]), pack2(lg[2 * S][2], lg[2 * S][3]),
                                            two ? pack2(lg[t1][0], lg[t1][1]) : 0u, two ? pack2(lg[t1][2], lg[t1][3]) : 0u};
                    if (S & 1) a1 = mfma32(wdx[S], gop, a1);
                    else a0 = mfma32(wdx[S], gop, a0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[t][0], lg[t][0], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[t][1], lg[t][1], a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[t][2], lg[t][2], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wd[t][3], lg[t][3], a1, 0, 0, 0);
                }
            }
            f32x4 v = a0 + a1;
            if (d.mask_dx) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xq[e] > 0.0f ? v[e] : 0.0f;
            }
            if (valid) Act<T>::st4(d.dx, (long long)mm * OC_CIN + 4 * g, v);
        }
        // weight gradient: pixel into the contraction position through the wave's LDS tile
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(&gs[i * PG + 16 * t + 4 * g]) = lg[t];
        *reinterpret_cast<f32x4 *>(&xs[i * 16 + 4 * g]) = xq;
        __builtin_amdgcn_wave_barrier();
        if (B16) {
            float xa[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xa[kk] = xs[(4 * kk + g) * 16 + i];
            const u32x4 xop = u32x4{pack2(xa[0], xa[1]), pack2(xa[2], xa[3]), 0u, 0u};     // (exact: x is bf16)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float gb[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    gb[kk] = gs[(4 * kk + g) * PG + 16 * t + i];
                    bsum[t] += gb[kk];
                }
                accw[t] = mfma32(xop, u32x4{pack2(gb[0], gb[1]), pack2(gb[2], gb[3]), 0u, 0u}, accw[t]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float xa = xs[(4 * kk + g) * 16 + i];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float gb = gs[(4 * kk + g) * PG + 16 * t + i];
                    bsum[t] += gb;
                    accw[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, gb, accw[t], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // D[row = c = 4g + r][col = oab = 16T + i]; bias sums over the pixel index g
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(16 * t + i) * 16 + 4 * g + r] = accw[t][r];
        float bs = bsum[t];
        bs += __shfl_xor(bs, 16);
        bs += __shfl_xor(bs, 32);
        if (g == 0) red[wave][NT * 256 + 16 * t + i] = bs;
    }
    __syncthreads();
    float *P = d.partial + (long long)blockIdx.x * (nab * OC_CIN + nab);
    for (int idx = tid; idx < nab * OC_CIN; idx += 256)
        P[idx] = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
    for (int idx = tid; idx < nab; idx += 256) {
        const int e = NT * 256 + idx;
        P[nab * OC_CIN + idx] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    }
}
