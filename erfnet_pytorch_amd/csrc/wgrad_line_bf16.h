// Weight gradient of the 3-tap 1-D convolutions of a non_bottleneck_1d on bf16 tensors
// (gfx950, v_mfma_f32_32x32x16_bf16, contraction index = pixel, fp32 accumulation and output).
//
//   dW[t][k][j] = sum_p X[p + (t-1)*dil along the conv axis][k] * G[p][j],   db[j] = sum_p G[p][j]
//
// Same walk as wgrad_line_kernel: a block takes segments of 64 consecutive positions of an image
// line, stages the G rows of the segment and the X rows of the segment extended by `dil` positions
// on both sides (zero outside the line) and feeds the three taps from the same X tile by shifting the
// row index.  Both MFMA operands have the PIXEL as their contraction index while the tensors are
// stored [pixel][channel]; the fragments (8 consecutive pixels of one channel per lane) come out of
// LDS through the transposing read ds_read_b64_tr_b16, two reads per fragment.  Tile rows are
// 64 bytes longer than the data, which spreads the four rows a transposed read touches over
// different banks.  Partial sums go to the slab format of wgrad_f32_kernel ([3][K][N] + bias rows,
// summed in a fixed order by reduce_partials_kernel).
#pragma once
#include "common.h"
#include "wgrad_line.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

// 8 consecutive rows (pixels) r0 .. r0+7 of column c0 + (lane & 15 | lane-group bit) as one MFMA
// operand fragment.  tile: LDS byte address of the tile, rows `pitch` bytes apart.
// Lane l: half h = l >> 5 takes rows r0 + 8h .. +7; its 16-lane group covers columns
// cbase + 16*((l >> 4) & 1) .. +15 and the lane receives column + (l & 15).
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *tile, int pitch, int r0, int cbase, int lane)
{
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned char *a = tile + (r0 + 8 * (g >> 1) + q) * pitch + (cbase + 16 * (g & 1) + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(a + 4 * pitch));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// NC channels on both sides; block tile = all NC input channels x all NC output channels,
// GI*IB*32 = NC = GJ*JB*32, 4 or 8 waves.  grid = (slabs, 1, convolutions).
template <int NC, int IB, int JB, int GI, int GJ, int MINB>
__global__ __launch_bounds__(GI *GJ * 64, MINB) void wgrad_line_bf16_kernel(const WgradLineBatch batch)
{
    const WgradLineDesc &d = batch.d[blockIdx.z];
    constexpr int L = 64, XR = 96, NT = GI * GJ * 64;
    static_assert(GI * IB * 32 == NC && GJ * JB * 32 == NC && (GI * GJ == 4 || GI * GJ == 8), "tile shape");
    constexpr int CPR = NC / 8;                             // 16-byte chunks per row
    constexpr int PITCH = NC * 2 + 64;                      // bytes
    constexpr int XP = XR * CPR / NT, GP = L * CPR / NT;
    static_assert(XR * CPR % NT == 0 && L * CPR % NT == 0, "whole passes");
    __shared__ __attribute__((aligned(16))) unsigned char Xs[XR * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[L * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / GJ, wj = wave % GJ;
    const int l31 = lane & 31, h = lane >> 5;
    const int dil = d.dil;
    const int s0 = blockIdx.x * d.spb;
    const int s1 = s0 + d.spb < d.nseg ? s0 + d.spb : d.nseg;

    f32x16 acc[3][IB][JB];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < IB; ++a)
#pragma unroll
            for (int b = 0; b < JB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.0f;
    float bsum[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) bsum[b] = 0.0f;

    const unsigned P = (unsigned)d.N * d.img_pixels;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, P * NC * 2u);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, P * NC * 2u);
    u32x4 xreg[XP], greg[GP];

    auto load_seg = [&](int s) {
        const int line = s / d.segs_per_line;
        const int q0 = (s - line * d.segs_per_line) * L;
        const int n = line / d.lines_per_img;
        const int ql = line - n * d.lines_per_img;
        const unsigned base = (unsigned)n * d.img_pixels + (unsigned)ql * d.qstride;
        const int xrows = L + 2 * dil;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, c = (slot % CPR) * 8;
            const int q = q0 - dil + r;
            unsigned off = ((base + (unsigned)(q * d.pstride)) * NC + c) * 2u;
            off |= (unsigned)(((xrows - 1 - r) | q | (d.Len - 1 - q)) >> 31) & ERF_OOB;
            xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)off, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, c = (slot % CPR) * 8;
            const int q = q0 + r;
            unsigned off = ((base + (unsigned)(q * d.pstride)) * NC + c) * 2u;
            off |= (unsigned)((d.Len - 1 - q) >> 31) & ERF_OOB;
            greg[i] = __builtin_amdgcn_raw_buffer_load_b128(r_g, (int)off, 0, 0);
        }
    };
    auto store_seg = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            *reinterpret_cast<u32x4 *>(Xs + (slot / CPR) * PITCH + (slot % CPR) * 16) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            *reinterpret_cast<u32x4 *>(Gs + (slot / CPR) * PITCH + (slot % CPR) * 16) = greg[i];
        }
    };

    if (s0 < s1) load_seg(s0);
    for (int s = s0; s < s1; ++s) {
        store_seg();
        __syncthreads();
        if (s + 1 < s1) load_seg(s + 1);
        // k-step ks = positions 16 ks .. 16 ks + 15 of the segment; the X-tile row of tap t is
        // position + t*dil (row 0 of the X tile is position -dil)
#pragma unroll
        for (int ks = 0; ks < L / 16; ++ks) {
            bf16x8 gf[JB], xf[3][IB];
#pragma unroll
            for (int b = 0; b < JB; ++b) gf[b] = tr_frag(Gs, PITCH, 16 * ks, (wj * JB + b) * 32, lane);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < IB; ++a) xf[t][a] = tr_frag(Xs, PITCH, 16 * ks + t * dil, (wi * IB + a) * 32, lane);
            if (wi == 0) {      // column sums of G (the bias gradient): one wave row adds them up
#pragma unroll
                for (int b = 0; b < JB; ++b)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[b] += (float)gf[b][e];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < IB; ++a)
#pragma unroll
                    for (int b = 0; b < JB; ++b)
                        acc[t][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[t][a], gf[b], acc[t][a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    // partial slab [3][K][N], then the bias rows ([3][N]; row 0 carries the column sums of G)
    const long long slab = 3LL * NC * NC + 3 * NC;
    float *Pp = d.partial + (long long)blockIdx.x * slab;
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        const int j = (wj * JB + b) * 32 + l31;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int a = 0; a < IB; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = (wi * IB + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    Pp[((long long)t * NC + k) * NC + j] = acc[t][a][b][r];
                }
        if (wi == 0) {
            const float other = __shfl_xor(bsum[b], 32);
            if (h == 0) Pp[3LL * NC * NC + j] = bsum[b] + other;
        }
    }
}

// The same weight gradient on fp32 tensors, computed on the bf16 matrix pipe: both operands are split
// into three bf16 terms (x = xh + xm + xl exactly, see line_x3.h) while they are staged into LDS
// (three planes per tile), and six of the nine partial products are accumulated in fp32:
//   x g ~= xh gh + xh gm + xm gh + xm gm + xh gl + xl gh        (the rest is below 2^-24 |x g|)
// KG wave groups share a block's tiles and split the k-steps (positions) of every segment between
// them; each group owns a partial slab (slab index blockIdx.x * KG + group), so no sum crosses waves.
// XR = rows of the X tile (64 positions + the dilation on both sides, at most XR - 64): 96 serves every
// dilation up to 16; 72 (dilations up to 4) lets two 64-channel blocks share a CU.
template <int NC, int IB, int JB, int GI, int GJ, int KG, int MINB, int XR = 96>
__global__ __launch_bounds__(GI *GJ *KG * 64, MINB) void wgrad_line_x3_kernel(const WgradLineBatch batch)
{
    const WgradLineDesc &d = batch.d[blockIdx.z];
    constexpr int L = 64, NT = GI * GJ * KG * 64;
    static_assert(GI * IB * 32 == NC && GJ * JB * 32 == NC && (4 % KG) == 0, "tile shape");
    constexpr int Q = NC / 4;                               // float4 per row
    constexpr int PITCH = NC * 2 + 64;                      // bytes per row of a bf16 plane
    constexpr int XSZ = XR * PITCH, GSZ = L * PITCH;
    constexpr int XP = (XR * Q + NT - 1) / NT, GP = L * Q / NT;
    static_assert(L * Q % NT == 0, "whole passes");
    static_assert(3 * (XSZ + GSZ) * MINB <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char Xs[3 * XSZ];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[3 * GSZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave / (GI * GJ), wq = wave % (GI * GJ);
    const int wi = wq / GJ, wj = wq % GJ;
    const int l31 = lane & 31, h = lane >> 5;
    const int dil = d.dil;
    const int s0 = blockIdx.x * d.spb;
    const int s1 = s0 + d.spb < d.nseg ? s0 + d.spb : d.nseg;

    f32x16 acc[3][IB][JB];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < IB; ++a)
#pragma unroll
            for (int b = 0; b < JB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.0f;
    float bsum[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) bsum[b] = 0.0f;

    const unsigned P = (unsigned)d.N * d.img_pixels;
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, P * NC * 4u);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, P * NC * 4u);
    f32x4 xreg[XP], greg[GP];

    auto load_seg = [&](int s) __attribute__((always_inline)) {
        const int line = s / d.segs_per_line;
        const int q0 = (s - line * d.segs_per_line) * L;
        const int n = line / d.lines_per_img;
        const int ql = line - n * d.lines_per_img;
        const unsigned base = (unsigned)n * d.img_pixels + (unsigned)ql * d.qstride;
        const int xrows = L + 2 * dil;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / Q, c = (slot % Q) * 4;
            const int q = q0 - dil + r;
            unsigned off = (base + (unsigned)(q * d.pstride)) * NC + c;
            off |= (unsigned)(((xrows - 1 - r) | q | (d.Len - 1 - q)) >> 31) & ERF_OOB;      // (also rows >= XR)
            xreg[i] = Act<float>::load4(r_x, off);
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / Q, c = (slot % Q) * 4;
            const int q = q0 + r;
            unsigned off = (base + (unsigned)(q * d.pstride)) * NC + c;
            off |= (unsigned)((d.Len - 1 - q) >> 31) & ERF_OOB;
            greg[i] = Act<float>::load4(r_g, off);
        }
    };
    auto split_store = [&](unsigned char *dst, int plane_bytes, f32x4 v) __attribute__((always_inline)) {
        const u32x2 hi = pack_bf16x4(v);
        const f32x4 r1 = v - unpack_bf16x4(hi);
        const u32x2 mid = pack_bf16x4(r1);
        const u32x2 lo = pack_bf16x4(r1 - unpack_bf16x4(mid));
        *reinterpret_cast<u32x2 *>(dst) = hi;
        *reinterpret_cast<u32x2 *>(dst + plane_bytes) = mid;
        *reinterpret_cast<u32x2 *>(dst + 2 * plane_bytes) = lo;
    };
    auto store_seg = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            if (XR * Q % NT == 0 || slot < XR * Q) split_store(Xs + (slot / Q) * PITCH + (slot % Q) * 8, XSZ, xreg[i]);
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            split_store(Gs + (slot / Q) * PITCH + (slot % Q) * 8, GSZ, greg[i]);
        }
    };

    if (s0 < s1) load_seg(s0);
    for (int s = s0; s < s1; ++s) {
        store_seg();
        __syncthreads();
        if (s + 1 < s1) load_seg(s + 1);
#pragma unroll
        for (int kk = 0; kk < L / 16 / KG; ++kk) {
            const int ks = kg * (L / 16 / KG) + kk;
            bf16x8 gf[3][JB], xf[3][3][IB];                 // [plane][...]
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int b = 0; b < JB; ++b) gf[p][b] = tr_frag(Gs + p * GSZ, PITCH, 16 * ks, (wj * JB + b) * 32, lane);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int a = 0; a < IB; ++a)
                        xf[p][t][a] = tr_frag(Xs + p * XSZ, PITCH, 16 * ks + t * dil, (wi * IB + a) * 32, lane);
            {   // column sums of G (the bias gradient): every wave row adds them up (no branch in the
                // k-step sequence: the scheduler moves fragment reads across k-steps), row 0 stores them
#pragma unroll
                for (int b = 0; b < JB; ++b)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[b] += ((float)gf[2][b][e] + (float)gf[1][b][e]) + (float)gf[0][b][e];
            }
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                // (X plane, G plane), small terms first: lh hl mm mh hm hh
                const int px = term == 0 ? 2 : (term == 1 || term == 4 || term == 5 ? 0 : 1);
                const int pg = term == 1 ? 2 : (term == 0 || term == 3 || term == 5 ? 0 : 1);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int a = 0; a < IB; ++a)
#pragma unroll
                        for (int b = 0; b < JB; ++b)
                            acc[t][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[px][t][a], gf[pg][b], acc[t][a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    const long long slab = 3LL * NC * NC + 3 * NC;
    float *Pp = d.partial + ((long long)blockIdx.x * KG + kg) * slab;
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        const int j = (wj * JB + b) * 32 + l31;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int a = 0; a < IB; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = (wi * IB + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    Pp[((long long)t * NC + k) * NC + j] = acc[t][a][b][r];
                }
        if (wi == 0) {
            const float other = __shfl_xor(bsum[b], 32);
            if (h == 0) Pp[3LL * NC * NC + j] = bsum[b] + other;
        }
    }
}
