// Implicit-GEMM convolution for bf16 tensors on v_mfma_f32_32x32x16_bf16 (fp32 accumulation): the
// stride-2 layers between the stages (3x3 s2 conv, 3x3 s2 transposed conv as four output-parity
// groups) and every other convolution the line / 16-channel kernels do not take.
//
// Same decomposition as igemm_f32_kernel (M = output pixels, N = output channels, K = taps x input
// channels; one K-step = one tap x 32 channels; A and B tiles loaded to registers one step ahead and
// written to a double-buffered LDS image), with three differences:
//   * a tile row is 32 bf16 = 64 bytes = four 16-byte chunks, XOR-swizzled by the row for A and B
//     alike; a lane's ds_read_b128 is a complete MFMA operand (8 consecutive k of its row);
//   * the product is formed transposed, D[channel][pixel] = W * X: a lane then holds groups of four
//     consecutive channels of ONE pixel, so residual / mask loads and the output store are 8-byte
//     accesses (bf16x4) instead of one element per lane;
//   * weights are packed as bf16 ([tap][k/32][j][32], pack_weights_kernel with bf16 = 1).
// Requirements (checked on the host, igemm_bf16_ok): K, in_C, in_coff multiples of 8 with a 16-byte
// aligned input; Nout, out_C, out_coff (and the residual / mask tensors' C and offset) multiples of 4
// with 8-byte aligned tensors.
#pragma once
#include "common.h"
#include "igemm.h"

__device__ __forceinline__ int bf_swz(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

template <int WM, int WN, int GM, int GN, int MINW>
__global__ __launch_bounds__(GM *GN * 64, MINW) void igemm_bf16_kernel(const IgemmDesc d)
{
    constexpr int BM = GM * WM * 32, BN = GN * WN * 32, NT = GM * GN * 64;
    constexpr int KC = 32, CPR = 4, ROWB = 64;                   // channels per K-step, chunks and bytes per tile row
    constexpr int A_SLOTS = BM * CPR, B_SLOTS = BN * CPR;
    constexpr int A_PASSES = (A_SLOTS + NT - 1) / NT, B_PASSES = (B_SLOTS + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char As[2][BM * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][BN * ROWB];
    __shared__ RowInfo rows[BM];
    __shared__ int taps[ERF_MAX_TAPS * 2];
    __shared__ __attribute__((aligned(16))) float cst[2][BN];   // per-channel scale / shift of the epilogue

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
        const unsigned m = (unsigned)m0 + (unsigned)r;      // (host: M < 2^31; 32-bit divisions)
        RowInfo ri;
        if (m < (unsigned)M) {
            const unsigned ghw = (unsigned)(d.GH * d.GW);
            const int n = (int)(m / ghw);
            const unsigned rem = m - (unsigned)n * ghw;
            const int gi = (int)(rem / (unsigned)d.GW), gj = (int)rem - gi * d.GW;
            ri.src_base = n * d.in_H * d.in_W;
            ri.hs_ws = (gi * d.ss) | ((gj * d.ss) << 16);
            ri.out_pix = (n * d.out_H + gi * d.os + d.goa[grp]) * d.out_W + gj * d.os + d.gob[grp];
            ri.n = n;
        } else {
            ri.src_base = -1; ri.hs_ws = 0; ri.out_pix = -1; ri.n = 0;
        }
        rows[r] = ri;
    }
    for (int c = tid; c < BN; c += NT) {
        float sc = 1.0f, sh = 0.0f;
        if (j0 + c < d.Nout) epi_channel_affine(d.epi, j0 + c, sc, sh);
        cst[0][c] = sc;
        cst[1][c] = sh;
    }
    __syncthreads();

    const int kchunks = (d.K + KC - 1) / KC;
    const int tap0 = __builtin_amdgcn_readfirstlane(d.gtap0[grp]);
    const int nsteps = __builtin_amdgcn_readfirstlane(d.gntaps[grp]) * kchunks;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    struct Stage {
        u32x4 a[A_PASSES], b[B_PASSES];
    };
    const __amdgpu_buffer_rsrc_t r_in = make_rsrc(d.in, (unsigned)d.N * d.in_H * d.in_W * d.in_C * 2u);

    int ld_t = tap0, ld_kc = 0, cp_kc = 0;         // (wave-uniform counters)
    auto load_step = [&](int s, Stage &R) __attribute__((always_inline)) {
        // (steps are loaded in ascending order: the tap and channel chunk of step s are counted, not divided out)
        const int t = ld_t, kc = ld_kc, kc0 = kc * KC;
        if (++ld_kc == kchunks) { ld_kc = 0; ++ld_t; }
        const int dh = taps[t * 2 + 0], dw = taps[t * 2 + 1];
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, ac = slot % CPR;
            const RowInfo ri = rows[r < BM ? r : 0];
            const int ih = (ri.hs_ws & 0xffff) + dh, iw = (ri.hs_ws >> 16) + dw;
            const int kk = kc0 + ac * 8;
            const bool ok = slot < A_SLOTS && ri.src_base >= 0 && (unsigned)ih < (unsigned)d.in_H &&
                            (unsigned)iw < (unsigned)d.in_W && kk < d.K;
            unsigned off = ((unsigned)(ri.src_base + ih * d.in_W + iw) * d.in_C + d.in_coff + kk) * 2u;
            off |= ok ? 0u : ERF_OOB;
            R.a[i] = __builtin_amdgcn_raw_buffer_load_b128(r_in, (int)off, 0, 0);
        }
        const bf16_t *wt = (const bf16_t *)d.wp + ((long long)(t * kchunks + kc) * d.Npad + j0) * KC;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int slot = tid + i * NT;
            R.b[i] = *reinterpret_cast<const u32x4 *>(wt + (slot < B_SLOTS ? slot : 0) * 8);
        }
    };
    auto store_step = [&](int buf, const Stage &R) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, ac = slot % CPR;
            if (A_SLOTS % NT == 0 || slot < A_SLOTS)
                *reinterpret_cast<u32x4 *>(&As[buf][r * ROWB + (bf_swz(r, ac) << 4)]) = R.a[i];
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int slot = tid + i * NT;
            const int r = slot / CPR, bc = slot % CPR;
            if (B_SLOTS % NT == 0 || slot < B_SLOTS)
                *reinterpret_cast<u32x4 *>(&Bs[buf][r * ROWB + (bf_swz(r, bc) << 4)]) = R.b[i];
        }
    };
    auto k_step = [&](int s, Stage &R) __attribute__((always_inline)) {
        const int buf = s & 1;
        if (s + 1 < nsteps) store_step(buf ^ 1, R);
        if (s + 2 < nsteps) load_step(s + 2, R);
        const int kc0 = cp_kc * KC;
        if (++cp_kc == kchunks) cp_kc = 0;
        const unsigned char *Ab = As[buf], *Bb = Bs[buf];
#pragma unroll
        for (int q = 0; q < KC / 16; ++q) {
            if (kc0 + q * 16 >= d.K) break;        // wave-uniform: short last chunk
            bf16x8 xf[WM], wf[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                const int row = (wm * WM + a) * 32 + l31;
                xf[a] = *reinterpret_cast<const bf16x8 *>(&Ab[row * ROWB + (bf_swz(row, 2 * q + h) << 4)]);
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                const int col = (wn * WN + b) * 32 + l31;
                wf[b] = *reinterpret_cast<const bf16x8 *>(&Bb[col * ROWB + (bf_swz(col, 2 * q + h) << 4)]);
            }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    };

    Stage st0;
    {
        Stage st1;
        load_step(0, st0);
        if (nsteps > 1) load_step(1, st1);
        store_step(0, st0);
        if (nsteps > 1) st0 = st1;
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) k_step(s, st0);

    // Epilogue.  D[channel][pixel]: column = pixel = lane & 31, row = channel = (r&3) + 8*(r>>2) + 4*(lane>>5):
    // register group g4 = r >> 2 holds channels 8*g4 + 4*h .. +3 of the lane's pixel.
    const unsigned Pout = (unsigned)d.N * d.out_H * d.out_W;
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(d.out, Pout * d.out_C * 2u);
    const __amdgpu_buffer_rsrc_t r_res = make_rsrc(d.epi.res, Pout * d.epi.res_C * 2u);
    const __amdgpu_buffer_rsrc_t r_mask = make_rsrc(d.epi.mask, Pout * d.epi.mask_C * 2u);
    const __amdgpu_buffer_rsrc_t r_ns = make_rsrc(d.epi.nscale, (unsigned)d.N * d.Nout * 4u);
    const bool has_mask = d.epi.mask != nullptr, has_ns = d.epi.nscale != nullptr, has_res = d.epi.res != nullptr;
    const bool relu = d.epi.relu != 0;
    const bool plain = !has_mask && !has_ns && !has_res;
#pragma unroll
    for (int a = 0; a < WM; ++a) {
        const RowInfo ri = rows[(wm * WM + a) * 32 + l31];
        const unsigned opx = (unsigned)ri.out_pix;
        const unsigned rbad = (unsigned)(ri.out_pix >> 31) & ERF_OOB;
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            if (plain) {        // (wave-uniform) no residual, mask or per-image scale: a third of the general path's instructions
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cl = (wn * WN + b) * 32 + 8 * g4 + 4 * h;
                    const unsigned bad = (j0 + cl < d.Nout ? 0u : ERF_OOB) | rbad;
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(&cst[0][cl]);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(&cst[1][cl]);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[a][b][4 * g4 + e] * sc[e] + sh[e];
                        v[e] = relu ? fmaxf(x, 0.0f) : x;
                    }
                    Act<bf16_t>::store4(r_out, (opx * d.out_C + d.out_coff + (unsigned)(j0 + cl)) | bad, v);
                }
                continue;
            }
            f32x4 rres[4], rmask[4], rns[4];
            unsigned cbad[4], cc[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cl = (wn * WN + b) * 32 + 8 * g4 + 4 * h;      // channel within the block tile
                cc[g4] = (unsigned)(j0 + cl);
                cbad[g4] = (j0 + cl < d.Nout ? 0u : ERF_OOB) | rbad;
                rres[g4] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                rmask[g4] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                rns[g4] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                if (has_res) rres[g4] = Act<bf16_t>::load4(r_res, (opx * d.epi.res_C + d.epi.res_coff + cc[g4]) | cbad[g4]);
                if (has_mask) rmask[g4] = Act<bf16_t>::load4(r_mask, (opx * d.epi.mask_C + d.epi.mask_coff + cc[g4]) | cbad[g4]);
                if (has_ns) rns[g4] = buf_load4(r_ns, (((unsigned)ri.n * d.Nout + cc[g4]) * 4u) | cbad[g4]);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cl = (wn * WN + b) * 32 + 8 * g4 + 4 * h;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(&cst[0][cl]);
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(&cst[1][cl]);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[a][b][4 * g4 + e] * sc[e] + sh[e];
                    x = x * rns[g4][e] + rres[g4][e];
                    if (relu) x = fmaxf(x, 0.0f);
                    v[e] = rmask[g4][e] > 0.0f ? x : 0.0f;
                }
                Act<bf16_t>::store4(r_out, (opx * d.out_C + d.out_coff + cc[g4]) | cbad[g4], v);
            }
        }
    }
}
