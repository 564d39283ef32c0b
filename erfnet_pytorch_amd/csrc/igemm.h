// Implicit-GEMM convolution kernels on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
//
// Forward / data-gradient kernel (igemm_f32_kernel)
//   M = output pixels, N = output channels, K = taps x input channels, NHWC activations.
//   One K-step = one tap x KC (16 or 32) input channels.  The A tile (BM pixels x KC channels,
//   gathered at the tap-shifted source pixels, zero outside the image) and the B tile (KC x BN
//   pre-packed weights) are loaded into registers one step ahead and written to a double-buffered
//   LDS image:
//     A: [BM][KC] floats, 16-byte chunks XOR-swizzled by row  -> conflict-free ds_write_b128 and
//        ds_read_b128 (lanes = rows);
//     B: [BN][KC+4] floats (k contiguous, padded row pitch)   -> conflict-free ds_read_b128.
//   A wave owns WM x WN blocks of 32x32 outputs.  A lane's ds_read_b128 delivers 4 consecutive k for
//   its row; MFMA number m of the group takes element m, so the 2-deep k of the instruction is
//   {8q+m, 8q+4+m} — A and B use the same permutation, the sum over k is unchanged.
//
// Weight-gradient kernel (wgrad_f32_kernel)
//   D[k][j] += sum_pixels X[pix][k] * G[pix][j] per tap; the MFMA's i index is the input channel,
//   its k index the pixel.  Pixels are split over blockIdx.x; each block writes its partial
//   [tap][k][j] tile to a slab, a second kernel (reduce_partials_kernel) sums the slabs in a fixed
//   order (bitwise reproducible) and scatters into torch's weight layout.
#pragma once
#include "common.h"

// Channels per K-step KC is 16 or 32.  A-tile rows hold KC floats in 16-byte chunks, XOR-swizzled so
// that both the staging ds_write_b128 and the fragment ds_read_b128 (lanes = rows) are free of bank
// conflicts; B-tile rows (one per output channel) are KC + 4 floats apart for the same reason.
template <int KC>
__device__ __forceinline__ int a_swz(int row, int chunk)
{
    return KC == 32 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
}

// Weights are repacked once per call from torch's layout into the order the B tile is consumed:
//   wp[tap][k-chunk][j][KC]  (k = contraction channel, j = output channel, zero padded to Npad
//   columns and to whole KC-channel chunks), so a K-step's B tile is one contiguous run that is
//   copied with 16-byte loads.  Up to 32 convolutions (a chain of eight non_bottleneck_1d) per launch.
struct PackDesc {
    const float *w;
    float *wp;
    int ntaps, K, Nout, Npad, wsk, wsn, kc;    // kc = channels per K-step (16 or 32); 0 = nothing to pack here
    int woff[ERF_MAX_TAPS];
    int line;                                  // 1 = bf16 MFMA fragments for the line kernel (K = Nout = 64 / 128) into lwp; 2 = three split planes
    bf16_t *lwp;
    int bf16;                                  // 1 = wp receives bf16 values (igemm_bf16_kernel), same order; 3 = three
                                               // bf16 planes wp[plane][...] of the split value (igemm_x3_kernel)
};
#define ERF_PACK_BATCH 32
struct PackBatch {
    PackDesc p[ERF_PACK_BATCH];
};

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackBatch b)
{
    const PackDesc &d = b.p[blockIdx.y];
    const int KC = d.kc;
    const int kchunks = (d.K + KC - 1) / KC;
    const int total = d.ntaps * kchunks * d.Npad * KC;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int kk = idx % KC;
        int r = idx / KC;
        const int j = r % d.Npad;
        r /= d.Npad;
        const int kc = r % kchunks, t = r / kchunks;
        const int k = kc * KC + kk;
        float v = 0.0f;
        if (k < d.K && j < d.Nout) v = d.w[d.woff[t] + (long long)k * d.wsk + (long long)j * d.wsn];
        if (d.bf16 == 3) {          // v = hi + mid + lo exactly (as line_pack_kernel splits)
            bf16_t *o = reinterpret_cast<bf16_t *>(d.wp);
            const bf16_t hi = f32_to_bf16(v);
            const float r1 = v - bf16_to_f32(hi);
            const bf16_t mid = f32_to_bf16(r1);
            o[idx] = hi;
            o[(long long)total + idx] = mid;
            o[2LL * total + idx] = f32_to_bf16(r1 - bf16_to_f32(mid));
        } else if (d.bf16) reinterpret_cast<bf16_t *>(d.wp)[idx] = f32_to_bf16(v);
        else d.wp[idx] = v;
    }
}

struct RowInfo {          // per output pixel of the block tile
    int src_base;         // n * in_H * in_W   (pixel index), -1 = row beyond M
    int hs_ws;            // (gi*ss) | (gj*ss) << 16
    int out_pix;          // output pixel index (n*out_H + oh)*out_W + ow
    int n;
};

// Main loop: LDS double buffering with one barrier per K-step.  In step s a wave first writes the
// tile of step s+1 (loaded into registers during step s-1) into the buffer that was read in step
// s-1, issues the global loads of step s+2, and then runs the MFMAs of step s from the other buffer.
// T = storage type of the activation tensors (fp32, or bf16 widened to fp32 on its way into LDS: the
// arithmetic is the fp32 MFMA either way).
template <int WM, int WN, int GM, int GN, int KC, int MINW, bool VEC, typename T = float>
__global__ __launch_bounds__(GM *GN * 64, MINW) void igemm_f32_kernel(const IgemmDesc d)
{
    constexpr int BM = GM * WM * 32, BN = GN * WN * 32, NT = GM * GN * 64;
    constexpr int CPR = KC / 4;                                  // 16-byte chunks per tile row
    constexpr int BPITCH = KC + 4;
    constexpr int A_SLOTS = BM * CPR, B_SLOTS = BN * CPR;        // float4 slots per tile
    constexpr int A_PASSES = (A_SLOTS + NT - 1) / NT, B_PASSES = (B_SLOTS + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float As[2][BM * KC];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BPITCH];
    __shared__ RowInfo rows[BM];
    __shared__ int taps[ERF_MAX_TAPS * 2];   // {dh, dw} per tap, read with ds_read: keeps the tap
                                             // lookup off the vector-memory counter

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < ERF_MAX_TAPS) {
        taps[tid * 2 + 0] = d.tdh[tid];
        taps[tid * 2 + 1] = d.tdw[tid];
    }
    const int wm = wave / GN, wn = wave % GN;
    const int l31 = lane & 31, h = lane >> 5;
    const int grp = blockIdx.z;
    const long long M = (long long)d.N * d.GH * d.GW;
    const long long m0 = (long long)xcd_tile(blockIdx.x, gridDim.x) * BM;
    const int j0 = blockIdx.y * BN;

    for (int r = tid; r < BM; r += NT) {
        const long long m = m0 + r;
        RowInfo ri;
        if (m < M) {
            const int n = (int)(m / (d.GH * d.GW));
            const int rem = (int)(m - (long long)n * d.GH * d.GW);
            const int gi = rem / d.GW, gj = rem - gi * d.GW;
            ri.src_base = n * d.in_H * d.in_W;
            ri.hs_ws = (gi * d.ss) | ((gj * d.ss) << 16);
            ri.out_pix = (n * d.out_H + gi * d.os + d.goa[grp]) * d.out_W + gj * d.os + d.gob[grp];
            ri.n = n;
        } else {
            ri.src_base = -1; ri.hs_ws = 0; ri.out_pix = -1; ri.n = 0;
        }
        rows[r] = ri;
    }
    __syncthreads();

    const int kchunks = (d.K + KC - 1) / KC;
    const int tap0 = __builtin_amdgcn_readfirstlane(d.gtap0[grp]);
    const int nsteps = __builtin_amdgcn_readfirstlane(d.gntaps[grp]) * kchunks;

    // per-column epilogue constants (channel affine, "column is padding" offset bit), loaded here so
    // that their latency hides behind the main loop
    float sc[WN], sh[WN];
    unsigned cbad[WN], cc[WN];
#pragma unroll
    for (int b = 0; b < WN; ++b) {
        const int c = j0 + (wn * WN + b) * 32 + l31;
        const bool cok = c < d.Nout;
        cc[b] = (unsigned)c;
        cbad[b] = cok ? 0u : ERF_OOB;
        epi_channel_affine(d.epi, cok ? c : 0, sc[b], sh[b]);
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    struct Stage {
        f32x4 a[A_PASSES], b[B_PASSES];
    };
    Stage st0;
    // The A operand is read through a buffer descriptor: a tap outside the image, a row beyond M
    // or a padded channel presents an out-of-range offset and loads zero, so every load is issued
    // unconditionally and its value is used as it is (a branch around a load, or a select behind
    // it, makes the compiler wait for each load separately and serialises their latencies).
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, (unsigned)d.N * d.in_H * d.in_W * d.in_C * Act<T>::B);

    auto load_step = [&](int s, Stage &R) {
        const int t = tap0 + s / kchunks;
        const int kc = s % kchunks, kc0 = kc * KC;
        const int dh = taps[t * 2 + 0], dw = taps[t * 2 + 1];
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, ac = slot % CPR;
            const RowInfo ri = rows[r < BM ? r : 0];
            const int ih = (ri.hs_ws & 0xffff) + dh, iw = (ri.hs_ws >> 16) + dw;
            const int kk = kc0 + ac * 4;
            const bool ok = slot < A_SLOTS && ri.src_base >= 0 && (unsigned)ih < (unsigned)d.in_H &&
                            (unsigned)iw < (unsigned)d.in_W && kk < d.K;
            unsigned off = (unsigned)(ri.src_base + ih * d.in_W + iw) * d.in_C + d.in_coff + kk;
            off |= ok ? 0u : ERF_OOB;
            if (VEC) {
                R.a[i] = Act<T>::load4(r_in, off);
            } else {        // channel counts / alignment that rule out 16-byte loads (rare)
#pragma unroll
                for (int e = 0; e < 4; ++e) R.a[i][e] = Act<T>::load1(r_in, (off + e) | (kk + e < d.K ? 0u : ERF_OOB));
            }
        }
        const float *wt = (const float *)d.wp + ((long long)(t * kchunks + kc) * d.Npad + j0) * KC;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int slot = tid + i * NT;
            R.b[i] = *reinterpret_cast<const f32x4 *>(wt + (slot < B_SLOTS ? slot : 0) * 4);
        }
    };

    auto store_step = [&](int buf, const Stage &R) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, ac = slot % CPR;
            if (A_SLOTS % NT == 0 || slot < A_SLOTS)
                *reinterpret_cast<f32x4 *>(&As[buf][r * KC + (a_swz<KC>(r, ac) << 2)]) = R.a[i];
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int slot = tid + i * NT;
            if (B_SLOTS % NT == 0 || slot < B_SLOTS)
                *reinterpret_cast<f32x4 *>(&Bs[buf][(slot / CPR) * BPITCH + (slot % CPR) * 4]) = R.b[i];
        }
    };

    // One K-step: write tile s+1 (held in R) to the other LDS buffer, reuse R for the loads of
    // tile s+2, run the MFMAs of tile s.  One barrier per step.
    auto k_step = [&](int s, Stage &R) {
        const int buf = s & 1;
        if (s + 1 < nsteps) store_step(buf ^ 1, R);
        if (s + 2 < nsteps) load_step(s + 2, R);
        const int kc0 = (s % kchunks) * KC;
        const float *Ab = As[buf], *Bb = Bs[buf];
#pragma unroll
        for (int q = 0; q < KC / 8; ++q) {
            if (kc0 + q * 8 >= d.K) break;     // wave-uniform: short last chunk
            f32x4 af[WM], bf[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                const int row = (wm * WM + a) * 32 + l31;
                af[a] = *reinterpret_cast<const f32x4 *>(&Ab[row * KC + (a_swz<KC>(row, 2 * q + h) << 2)]);
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                const int col = (wn * WN + b) * 32 + l31;
                bf[b] = *reinterpret_cast<const f32x4 *>(&Bb[col * BPITCH + q * 8 + h * 4]);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][m], bf[b][m], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    };

    // Prologue: the loads of the first two steps are in flight together (the second set of staging
    // registers is free here, the accumulators are not live yet), so only one memory latency is
    // exposed before the first MFMA.
    {
        Stage st1;
        load_step(0, st0);
        if (nsteps > 1) load_step(1, st1);
        store_step(0, st0);
        if (nsteps > 1) st0 = st1;
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) k_step(s, st0);

    // Epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // All global accesses go through buffer descriptors with 32-bit byte offsets: a row beyond M,
    // a padded column or an absent optional tensor (zero records) is an out-of-range access, which
    // loads zero and drops the store, so the code has no per-element branch or 64-bit address.
    const unsigned Pout = (unsigned)d.N * d.out_H * d.out_W;
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(d.out, Pout * d.out_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, Pout * d.epi.res_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, Pout * d.epi.mask_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_ns = make_rsrc(d.epi.nscale, (unsigned)d.N * d.Nout * 4u);
    const bool has_mask = d.epi.mask != nullptr, has_ns = d.epi.nscale != nullptr, relu = d.epi.relu != 0;
#pragma unroll
    for (int a = 0; a < WM; ++a) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {      // eight rows at a time keeps the temporaries small
            // rows beyond M carry out_pix = -1: the sign bit, kept as bit 31 of every byte offset,
            // puts the access out of range (plain bit arithmetic: a select here turns into branches)
            unsigned opx[8], onn[8], rbad[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = hf * 8 + q;
                const int row = (wm * WM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const RowInfo ri = rows[row];
                opx[q] = (unsigned)ri.out_pix;
                rbad[q] = (unsigned)(ri.out_pix >> 31) & ERF_OOB;
                onn[q] = (unsigned)ri.n * d.Nout;
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                float rres[8], rmask[8], rns[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned bad = rbad[q] | cbad[b];
                    rres[q] = Act<T>::load1(r_res, (opx[q] * d.epi.res_C + d.epi.res_coff + cc[b]) | bad);
                    rmask[q] = Act<T>::load1(r_mask, (opx[q] * d.epi.mask_C + d.epi.mask_coff + cc[b]) | bad);
                    rns[q] = buf_load1(r_ns, ((onn[q] + cc[b]) * 4u) | bad);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = acc[a][b][hf * 8 + q] * sc[b] + sh[b];
                    v = (has_ns ? v * rns[q] : v) + rres[q];
                    if (relu) v = fmaxf(v, 0.0f);
                    v = (!has_mask || rmask[q] > 0.0f) ? v : 0.0f;
                    Act<T>::store1(r_out, (opx[q] * d.out_C + d.out_coff + cc[b]) | rbad[q] | cbad[b], v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient.  Block tile: IBLK = GI*IB*32 input channels x JBLK = GJ*JB*32 output channels of
// one tap; blockIdx.x = pixel slab, blockIdx.y = (tap, i-tile, j-tile).
// LDS: X tile [PM pixels][IBLK], G tile [PM pixels][JBLK]; MFMA operands are read with ds_read_b32
// (lanes walk the channel, the two lane halves take pixel 2s and 2s+1).  PM = pixels per step:
// 32 for the large tiles; the small-channel tiles take 64-128 so that a step carries enough MFMAs
// for its two barriers.
template <int IB, int JB, int GI, int GJ, bool VEC, int PM = 32, typename T = float>
__global__ __launch_bounds__(GI *GJ * 64) void wgrad_f32_kernel(const WgradBatch batch)
{
    const WgradDesc &d = batch.d[blockIdx.z];      // up to four convolutions per launch
    constexpr int IBLK = GI * IB * 32, JBLK = GJ * JB * 32, NT = GI * GJ * 64;
    static_assert(PM <= GI * GJ * 64 && PM % 32 == 0, "one thread per pixel of a step keeps the pixel table");
    constexpr int XV = PM * IBLK / 4, GV = PM * JBLK / 4;        // float4 slots per tile
    constexpr int XP = (XV + NT - 1) / NT, GP = (GV + NT - 1) / NT;
    constexpr int XPITCH = IBLK, GPITCH = JBLK;
    __shared__ __attribute__((aligned(16))) float Xs[PM * XPITCH];
    __shared__ __attribute__((aligned(16))) float Gs[PM * GPITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / GJ, wj = wave % GJ;
    const int l31 = lane & 31, h = lane >> 5;

    int y = blockIdx.y;
    const int jt = y % d.jtiles; y /= d.jtiles;
    const int it = y % d.itiles; y /= d.itiles;
    const int t = y;
    const int i0 = it * IBLK, j0 = jt * JBLK;
    const int dh = d.tdh[t], dw = d.tdw[t], oa = d.toa[t], ob = d.tob[t];

    // pixel indices fit 31 bits (checked on the host): 32-bit index arithmetic throughout
    const unsigned M = (unsigned)d.N * d.GH * d.GW, GHW = (unsigned)d.GH * d.GW;
    const unsigned mbeg = blockIdx.x * (unsigned)d.split_pixels;
    unsigned mend = mbeg + d.split_pixels;
    if (mend > M) mend = M;
    const int nsteps = mbeg < mend ? (int)((mend - mbeg + PM - 1) / PM) : 0;

    f32x16 acc[IB][JB];
#pragma unroll
    for (int a = 0; a < IB; ++a)
#pragma unroll
        for (int b = 0; b < JB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bsum[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) bsum[b] = 0.0f;

    f32x4 xreg[XP], greg[GP];

    // Pixel table of the step being loaded: the first PM threads each follow one pixel of the
    // 32-pixel window (coordinates advanced incrementally, one division pair in the prologue),
    // everybody else reads the table instead of dividing per load.
    //   x: n*in_H*in_W (source image base, -1 = beyond the slab)   y: gi*ss   z: gj*ss
    //   w: pixel index of the gradient tensor for this tap
    __shared__ int4 ptab[2][PM];
    unsigned tn = 0, tgi = 0, tgj = 0;
    if (tid < PM) {
        const unsigned m = mbeg + tid;
        const unsigned mm = m < M ? m : 0;
        tn = mm / GHW;
        const unsigned rem = mm - tn * GHW;
        tgi = rem / (unsigned)d.GW;
        tgj = rem - tgi * d.GW;
    }
    auto table_step = [&](int s) {      // threads < PM: publish pixel mbeg + s*PM + tid, then advance
        if (tid < PM) {
            const unsigned m = mbeg + (unsigned)s * PM + tid;
            int4 e;
            e.x = m < mend ? (int)(tn * d.in_H * d.in_W) : -1;
            e.y = (int)tgi * d.ss;
            e.z = (int)tgj * d.ss;
            e.w = (int)((tn * d.g_H + tgi * d.os + oa) * d.g_W + tgj * d.os + ob);
            ptab[s & 1][tid] = e;
            tgj += PM;
            while (tgj >= (unsigned)d.GW) {
                tgj -= d.GW;
                if (++tgi == (unsigned)d.GH) { tgi = 0; ++tn; }
            }
        }
    };

    // operands through buffer descriptors: masked slots load zero (see igemm_f32_kernel)
    const __amdgpu_buffer_rsrc_t r_x = make_rsrc(d.in, (unsigned)d.N * d.in_H * d.in_W * d.in_C * Act<T>::B);
    const __amdgpu_buffer_rsrc_t r_g = make_rsrc(d.g, (unsigned)d.N * d.g_H * d.g_W * d.g_C * Act<T>::B);
    auto load_step = [&](int s) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            const int p = slot / (IBLK / 4), c = (slot % (IBLK / 4)) * 4;
            const int4 e = ptab[s & 1][p < PM ? p : 0];
            const int ih = e.y + dh, iw = e.z + dw;
            const bool ok = slot < XV && e.x >= 0 && i0 + c < d.K && (unsigned)ih < (unsigned)d.in_H &&
                            (unsigned)iw < (unsigned)d.in_W;
            unsigned off = (unsigned)(e.x + ih * d.in_W + iw) * d.in_C + d.in_coff + i0 + c;
            off |= ok ? 0u : ERF_OOB;
            if (VEC) xreg[i] = Act<T>::load4(r_x, off);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) xreg[i][q] = Act<T>::load1(r_x, (off + q) | (i0 + c + q < d.K ? 0u : ERF_OOB));
            }
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            const int p = slot / (JBLK / 4), c = (slot % (JBLK / 4)) * 4;
            const int4 e = ptab[s & 1][p < PM ? p : 0];
            const bool ok = slot < GV && e.x >= 0 && j0 + c < d.Nout;
            unsigned off = (unsigned)e.w * d.g_C + d.g_coff + j0 + c;
            off |= ok ? 0u : ERF_OOB;
            if (VEC) greg[i] = Act<T>::load4(r_g, off);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) greg[i][q] = Act<T>::load1(r_g, (off + q) | (j0 + c + q < d.Nout ? 0u : ERF_OOB));
            }
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int slot = tid + i * NT;
            if (slot < XV) *reinterpret_cast<f32x4 *>(&Xs[(slot / (IBLK / 4)) * XPITCH + (slot % (IBLK / 4)) * 4]) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int slot = tid + i * NT;
            if (slot < GV) *reinterpret_cast<f32x4 *>(&Gs[(slot / (JBLK / 4)) * GPITCH + (slot % (JBLK / 4)) * 4]) = greg[i];
        }
    };

    if (nsteps > 0) {
        table_step(0);
        __syncthreads();
        load_step(0);
    }
    for (int s = 0; s < nsteps; ++s) {
        store_step();
        if (s + 1 < nsteps) table_step(s + 1);
        __syncthreads();
        if (s + 1 < nsteps) load_step(s + 1);
        // operand fragments of pixel pair pp+1 are read while the MFMAs of pair pp run
        float af[2][IB], bf[2][JB];
        auto read_pair = [&](int pp, int set) {
            const int p = 2 * pp + h;
#pragma unroll
            for (int a = 0; a < IB; ++a) af[set][a] = Xs[p * XPITCH + (wi * IB + a) * 32 + l31];
#pragma unroll
            for (int b = 0; b < JB; ++b) bf[set][b] = Gs[p * GPITCH + (wj * JB + b) * 32 + l31];
        };
        read_pair(0, 0);
#pragma unroll
        for (int pp = 0; pp < PM / 2; ++pp) {
            const int cur = pp & 1;
            if (pp + 1 < PM / 2) read_pair(pp + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of this pair's MFMAs
#pragma unroll
            for (int b = 0; b < JB; ++b) bsum[b] += bf[cur][b];
#pragma unroll
            for (int a = 0; a < IB; ++a)
#pragma unroll
                for (int b = 0; b < JB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // partial slab: [tap][k][j], then one bias row (column sums of G) per tap
    const long long slab = (long long)d.ntaps * d.K * d.Nout + (long long)d.ntaps * d.Nout;
    float *P = d.partial + (long long)blockIdx.x * slab;
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        const int j = j0 + (wj * JB + b) * 32 + l31;
        if (j >= d.Nout) continue;
#pragma unroll
        for (int a = 0; a < IB; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = i0 + (wi * IB + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (k < d.K) P[((long long)t * d.K + k) * d.Nout + j] = acc[a][b][r];
            }
        }
        // column sums of this tap's G pixels, taken once per tap (first i-tile, first wave row)
        if (it == 0 && wi == 0) {
            const float other = __shfl_xor(bsum[b], 32);
            if (h == 0) P[(long long)d.ntaps * d.K * d.Nout + (long long)t * d.Nout + j] = bsum[b] + other;
        }
    }
}

// dst[woff[t] + k*wsk + j*wsn] = sum_s partial[s][t][k][j];
// db[j] = sum_s sum_{t in bias_taps} partial[s][bias row t][j]   (fixed order: reproducible).
// A slab may carry K more sums behind the bias rows (column sums of the k-side operand); with
// kside != 0 those, not the bias rows, are the bias gradient: db[k] = sum_s partial[s][tail k].
struct ReduceDesc {
    const float *partial;
    float *dw, *db;
    int nsplit, ntaps, K, Nout;
    int woff[ERF_MAX_TAPS], wsk, wsn;
    unsigned bias_taps;
    int tail, kside;        // tail = floats behind the bias rows in a slab (0 or K)
};

struct ReduceBatch {
    ReduceDesc r[ERF_WGRAD_BATCH];
};

// s + partial[g][off] + partial[g+8][off] + ... in that order, eight slabs requested per trip (added one
// by one the loads wait for each other: 64 dependent round trips at 512 slabs).
__device__ __forceinline__ float reduce_column8(const float *partial, long long slab, long long off, int g, int nsplit, float s)
{
    int i = g;
    for (; i + 56 < nsplit; i += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(i + 8 * u) * slab + off];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < nsplit; i += 8) s += partial[(long long)i * slab + off];
    return s;
}

// Block = 32 consecutive slab elements x 8 slab groups: thread (g, e) sums slabs g, g+8, ... of
// element e, the 8 group sums are then added in a fixed order through LDS.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceBatch batch)
{
    const ReduceDesc &d = batch.r[blockIdx.y];
    __shared__ float red[8][33];
    const long long nw = (long long)d.ntaps * d.K * d.Nout;
    const long long slab = nw + (long long)d.ntaps * d.Nout + d.tail;
    const int nb = d.kside ? d.K : d.Nout;                      // bias elements
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long idx = (long long)blockIdx.x * 32 + e;
    float s = 0.0f;
    if (idx < nw) {
        s = reduce_column8(d.partial, slab, idx, g, d.nsplit, s);
    } else if (idx < nw + nb) {
        const int j = (int)(idx - nw);
        if (d.kside) {
            const long long off = nw + (long long)d.ntaps * d.Nout + j;
            s = reduce_column8(d.partial, slab, off, g, d.nsplit, s);
        } else {
            for (int t = 0; t < d.ntaps; ++t) {
                if (!((d.bias_taps >> t) & 1u)) continue;
                s = reduce_column8(d.partial, slab, nw + (long long)t * d.Nout + j, g, d.nsplit, s);
            }
        }
    }
    red[g][e] = s;
    __syncthreads();
    if (g != 0 || idx >= nw + nb) return;
    float tot = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) tot += red[k][e];
    if (idx < nw) {
        const int j = (int)(idx % d.Nout);
        const int k = (int)((idx / d.Nout) % d.K);
        const int t = (int)(idx / ((long long)d.Nout * d.K));
        d.dw[d.woff[t] + (long long)k * d.wsk + (long long)j * d.wsn] = tot;
    } else if (d.db) {
        d.db[idx - nw] = tot;
    }
}

// The same reduction with 16-byte slab reads for slabs whose weight part and length are multiples of
// four floats (every 16 / 64 / 128-channel layer): a thread owns four consecutive elements, its
// slab group's loads are issued eight at a time, and the additions run in reduce_partials_kernel's
// order (slabs g, g+8, ... of a group one after the other, then the eight groups), so the results
// are bit-identical.  Blocks [0, wblocks) cover the weight part (128 elements each), the blocks
// behind them the bias elements with the scalar code.
__device__ __forceinline__ void reduce_partials4_body(const ReduceDesc &d, int wblocks)
{
    const long long nw = (long long)d.ntaps * d.K * d.Nout;
    const long long slab = nw + (long long)d.ntaps * d.Nout + d.tail;
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    if ((int)blockIdx.x < wblocks) {
        __shared__ f32x4 red4[8][33];
        const long long idx = ((long long)blockIdx.x * 32 + e) * 4;
        f32x4 s = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (idx < nw) {
            const float *src = d.partial + idx;
            int i = g;
            for (; i + 56 < d.nsplit; i += 64) {                // eight slabs of this group per trip
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + (long long)(i + 8 * u) * slab);
#pragma unroll
                for (int u = 0; u < 8; ++u) s = s + v[u];
            }
            for (; i < d.nsplit; i += 8) s = s + *reinterpret_cast<const f32x4 *>(src + (long long)i * slab);
        }
        red4[g][e] = s;
        __syncthreads();
        if (g != 0 || idx >= nw) return;
        f32x4 tot = red4[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot = tot + red4[k][e];
        const int j = (int)(idx % d.Nout);                      // Nout % 4 == 0: the four share k and t
        const int k = (int)((idx / d.Nout) % d.K);
        const int t = (int)(idx / ((long long)d.Nout * d.K));
        float *dst = d.dw + d.woff[t] + (long long)k * d.wsk + (long long)j * d.wsn;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[(long long)q * d.wsn] = tot[q];
        return;
    }
    __shared__ float red[8][33];
    const int nb = d.kside ? d.K : d.Nout;
    const int j = ((int)blockIdx.x - wblocks) * 32 + e;
    float s = 0.0f;
    if (j < nb) {
        if (d.kside) {
            const long long off = nw + (long long)d.ntaps * d.Nout + j;
            s = reduce_column8(d.partial, slab, off, g, d.nsplit, s);
        } else {
            for (int t = 0; t < d.ntaps; ++t) {
                if (!((d.bias_taps >> t) & 1u)) continue;
                s = reduce_column8(d.partial, slab, nw + (long long)t * d.Nout + j, g, d.nsplit, s);
            }
        }
    }
    red[g][e] = s;
    __syncthreads();
    if (g != 0 || j >= nb || !d.db) return;
    float tot = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) tot += red[k][e];
    d.db[j] = tot;
}

__global__ __launch_bounds__(256) void reduce_partials4_kernel(const ReduceBatch batch, int wblocks)
{
    reduce_partials4_body(batch.r[blockIdx.y], wblocks);
}

// The reductions a run of blocks deferred (erf_batch_end): up to 32 slabs sets of one geometry per launch.
#define ERF_REDUCE_WIDE 32
struct ReduceBatchWide {
    ReduceDesc r[ERF_REDUCE_WIDE];
};
__global__ __launch_bounds__(256) void reduce_partials4_wide_kernel(const ReduceBatchWide batch, int wblocks)
{
    reduce_partials4_body(batch.r[blockIdx.y], wblocks);
}
