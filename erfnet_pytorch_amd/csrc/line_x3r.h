// line_x3_kernel's arithmetic (fp32 tensors on the bf16 matrix pipe, three bf16 terms per operand,
// six MFMAs per product) with the operand tiles ROLLING through LDS: a block walks consecutive
// output lines of one chain  q, q + dil, q + 2 dil, ...  so an input line, once split into its three
// bf16 planes, serves as the upper tap of one output line, the centre tap of the next and the lower
// tap of the one after.  line_x3_kernel splits a line tile once per tap (three times); here it is
// split once, which removes two thirds of the global loads, of the conversion arithmetic and of the
// LDS plane writes — the 64-channel kernel was bound by instruction issue, not by the matrix pipe.
//
//   * outputs of a launch: u = (chain, j), chain = (image, segment, residue r of q mod dil),
//     q = r + j dil, j < Lc = lines_per_img / dil (host: dil divides lines_per_img); a block owns a
//     contiguous range of u, cut into runs at chain ends;
//   * an output line runs its taps in the fixed order  centre (line j), lo (j-1), hi (j+1)  — whatever
//     the block partition, so results do not depend on the grid;
//   * three LDS slots hold lines j-1, j, j+1 of the current output; the line the NEXT step sequence
//     needs (line j+1, or at a chain's end the first line of the next run) is split into the third
//     slot during the lo step's MFMAs — one piece of the conversion behind each MFMA, as in
//     line_x3_kernel — while the line after it is loaded into the staging registers.  That slot last
//     held line j-2, read for the last time before the previous output's barrier.  In the steady
//     state: one conversion and ONE barrier per output line.  A block starts with its first centre
//     line split in the open; that first centre step hosts the split of line j-1 (and fetches the
//     weight fragments of the other two taps behind its MFMAs);
//   * epilogue straight from the accumulators (a lane holds 4 consecutive channels of a pixel per
//     accumulator quad: 16-byte buffer stores; residual and mask are read the same way), so the ring
//     is all the LDS the kernel needs and an output line costs no further barrier.
#pragma once
#include "line_x3.h"

// A per-thread constant handed through an empty asm statement: the compiler can no longer prove that
// values derived from it are loop-invariant, so it recomputes them (one or two instructions) where they
// are used instead of keeping dozens of derived addresses in registers for the whole kernel — which
// pushed weight fragments out to scratch memory.
__device__ __forceinline__ unsigned x3r_opaque(unsigned v)
{
    asm volatile("" : "+v"(v));
    return v;
}

#ifndef X3R_PFA64
#define X3R_PFA64 0
#endif
#ifndef X3R_PFA128
#define X3R_PFA128 1
#endif
template <int NC, int BM>
__global__ __launch_bounds__(256, (NC == 64 ? 2 : 1)) void line_x3r_kernel(const LineDesc d)
{
    constexpr int GC = NC / 32, GP = 4 / GC, WM = BM / (32 * GP);
    static_assert((WM == 1 || WM == 2) && WM * GP * 32 == BM, "tile shape: one or two 32-pixel MFMA tiles per wave");
    constexpr int KS = NC / 16;                 // MFMA k-steps per tap
    constexpr int PITCH = NC * 2;               // bytes per pixel row of a bf16 plane
    constexpr int TP = BM * PITCH, T3 = 3 * TP; // one plane, one line tile (three planes)
    constexpr int F4R = NC / 4;                 // 16-byte chunks per fp32 pixel row
    constexpr int RPP = 256 / F4R;              // rows per copy pass of the block
    constexpr int PASSES = BM / RPP;            // float4 per thread and tile
    static_assert(KS % PASSES == 0 && (WM == 1 || PASSES == KS), "conversion schedule");
    constexpr int CSTRIDE = KS / PASSES;
    constexpr int OFF_CST = 3 * T3;
    constexpr int LDS_BYTES = OFF_CST + 2 * NC * 4;
    static_assert(LDS_BYTES * (NC == 64 ? 2 : 1) <= 160 * 1024, "LDS budget");
    constexpr size_t PL = (size_t)3 * NC * NC;  // elements per weight plane
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    float *cst = reinterpret_cast<float *>(lds + OFF_CST);          // sc[NC] sh[NC]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % GC, wp = wave / GC;
    const int l31 = lane & 31, h = lane >> 5;

    // ---- the block's share of the outputs (all of this is wave-uniform scalar arithmetic)
    const int dil = d.s1[0] < 0 ? -d.s1[0] : d.s1[0];
    const bool lo_is_tap0 = d.s1[0] < 0;        // weight tap 0 reads line q - dil (forward) or q + dil (data gradient)
    const int Lc = d.lines_per_img / dil;       // lines per chain
    const int cpi = d.segs * dil;               // chains per image
    const unsigned U = (unsigned)(d.nlines * d.segs);       // (host: below 2^31)
    const unsigned G = gridDim.x;
    const unsigned bix = xcd_tile(blockIdx.x, G);
    // contiguous shares of U / G outputs, the first U % G blocks one more (one 32-bit division)
    const unsigned uq = U / G, ur = U - uq * G;
    const int u0 = (int)(bix * uq + (bix < ur ? bix : ur)), u1 = u0 + (int)uq + (bix < ur ? 1 : 0);
    if (u0 >= u1) return;
    // The epilogue constants are requested before anything else (buffer loads: an absent vector has no
    // records and reads as zero; no branch near them, so nothing waits here) and land during the
    // scalar set-up below.
    float c_b, c_g, c_be, c_mu, c_va;
    {
        const unsigned cb = NC * 4u, vo = tid < NC ? (unsigned)tid * 4u : ERF_OOB;
        const BnAffine &bn = d.epi1.bn;
        c_b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(d.epi1.bias, d.epi1.bias ? cb : 0u), (int)vo, 0, 0));
        c_g = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(bn.gamma, bn.on && bn.gamma ? cb : 0u), (int)vo, 0, 0));
        c_be = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(bn.beta, bn.on && bn.beta ? cb : 0u), (int)vo, 0, 0));
        c_mu = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(bn.mean, bn.on ? cb : 0u), (int)vo, 0, 0));
        c_va = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(bn.var, bn.on ? cb : 0u), (int)vo, 0, 0));
    }
#ifdef ERF_LINE_STAMPS
    int stamp_i = 0;
#define X3R_STAMP()                                                                                \
    do {                                                                                           \
        if (d.stamps && tid == 0 && stamp_i < 48)                                                  \
            d.stamps[(size_t)blockIdx.x * 48 + stamp_i++] = __builtin_amdgcn_s_memtime();          \
    } while (0)
#else
#define X3R_STAMP() do {} while (0)
#endif
    X3R_STAMP();    // 0: start
    const unsigned lstep = (unsigned)(dil * d.qstride) * (unsigned)(NC * 4);    // bytes between lines of a chain

    struct Chain { unsigned base; int len; };       // byte offset of line 0's first pixel, pixels in the tile
    auto chain_at = [&](int c) __attribute__((always_inline)) {
        Chain k;
        const int n = c / cpi, rem = c - n * cpi;
        const int seg = rem / dil, r = rem - seg * dil;
        k.base = ((unsigned)n * d.img_pixels + (unsigned)r * d.qstride + (unsigned)(seg * BM) * d.pstride) * (unsigned)(NC * 4);
        k.len = d.L - seg * BM < BM ? d.L - seg * BM : BM;
        return k;
    };
    // stream of input lines: a run's lines x = j0 .. b of chain c, then the next run's (line j0 - 1 of
    // the block's first run is slipped in behind line j0 by the prologue)
    struct Str { int c, x, b, len; unsigned soff; };      // len == 0: past the end (loads nothing)
    auto str_run = [&](int c, int j0) __attribute__((always_inline)) {
        Str s;
        const Chain k = chain_at(c);
        const int rest = u1 - c * Lc;                       // outputs from line 0 of this chain to the block's end
        const int j1 = (rest < Lc ? rest : Lc) - 1;
        s.c = c; s.x = j0; s.b = j1 + 1 < Lc ? j1 + 1 : Lc - 1;
        s.len = k.len; s.soff = k.base + (unsigned)s.x * lstep;
        return s;
    };
    auto str_next = [&](const Str &s) __attribute__((always_inline)) {
        if (s.len == 0) return s;
        Str n = s;
        if (s.x < s.b) { n.x = s.x + 1; n.soff = s.soff + lstep; return n; }
        if ((long long)(s.c + 1) * Lc >= u1) { n.len = 0; n.soff = 0; return n; }
        return str_run(s.c + 1, 0);
    };

    // this thread's share of a line tile: float4 number c4 of rows r0 + p * RPP
    const float *in = reinterpret_cast<const float *>(d.in);
    const unsigned tbytes = (unsigned)((size_t)d.N * d.img_pixels * NC * 4);
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, tbytes);
    const int r0 = tid / F4R, c4 = tid % F4R;
    const unsigned vo0 = (unsigned)(r0 * d.pstride * NC + c4 * 4) * 4u;
    const unsigned vstep = (unsigned)(RPP * d.pstride * NC) * 4u;       // (scalar)
    f32x4 stg[PASSES];
    auto load_pass = [&](int p, const Str &s) __attribute__((always_inline)) {
        const unsigned vo = ((int)x3r_opaque((unsigned)r0) < s.len - p * RPP) ? x3r_opaque(vo0) + (unsigned)p * vstep : ERF_OOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, (int)s.soff, 0);
        stg[p] = __builtin_bit_cast(f32x4, v);
    };
    // byte offset of this thread's 8 bytes in a bf16 plane, pass p: rows p * RPP + r0; the swizzle term
    // has a period of 16 rows, so it takes one value (RPP = 16) or two (RPP = 8)
    constexpr int NPO = RPP >= 16 ? 1 : 2;
    unsigned po[NPO];
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
        const int r = i * RPP + r0;
        po[i] = (unsigned)(r0 * PITCH + (c4 & 1) * 8 + (((c4 >> 1) ^ line_swz<NC>(r)) << 4));
    }
    auto plane_off = [&](int p) __attribute__((always_inline)) {
        return x3r_opaque(po[p % NPO]) + (unsigned)(p * RPP * PITCH);
    };

    // ---- prologue: first centre line into the staging registers, constants, the centre tap's weights,
    // first line split
    int oc = u0 / Lc, j = u0 - oc * Lc;                 // output cursor: chain, line of the chain
    Str cv = str_run(oc, j);                            // the line in the staging registers (next to be split)
#pragma unroll
    for (int p = 0; p < PASSES; ++p) load_pass(p, cv);
    Str after = str_next(cv);                           // the stream behind the first centre line
    Str nx = after;                                     // the line loaded while cv is split
    if (j > 0) { nx = cv; nx.x = j - 1; nx.soff = cv.soff - lstep; }
    X3R_STAMP();    // first loads issued
    {   // epi_channel_affine's arithmetic on the values requested at the top
        float sc = 1.0f, sh = c_b;
        if (d.epi1.bn.on) {
            const BnAffine &bn = d.epi1.bn;
            const float inv = bn.var_is_invstd ? c_va : rsqrtf(c_va + bn.eps);
            const float g = (bn.gamma ? c_g : 1.0f) * inv;
            sh = (sh - c_mu) * g + c_be;
            sc = g;
        }
        if (tid < NC) { cst[tid] = sc; cst[NC + tid] = sh; }
    }

    X3R_STAMP();    // constants done
    // A CU's weight fragments (144 / 288 KB through a 64-byte-per-clock path) would hold every wave at
    // its load instructions for thousands of cycles: only the centre tap's are loaded here, the other
    // two taps' piecewise behind the MFMAs of the block's first step (run_step, FIRST).
    bf16x8 w[3][3][KS];         // [plane][slot: 0 = line j-1, 1 = line j, 2 = line j+1][k-step]
    auto w_frag = [&](int P, int sl, int s) __attribute__((always_inline)) {
        const int t = lo_is_tap0 ? sl : 2 - sl;
        return *reinterpret_cast<const bf16x8 *>(d.wp1 + P * PL + ((size_t)((t * KS + s) * GC + cg) * 64 + lane) * 8);
    };
#pragma unroll
    for (int P = 0; P < 3; ++P)
#pragma unroll
        for (int s = 0; s < KS; ++s) w[P][1][s] = w_frag(P, 1, s);
    X3R_STAMP();    // weights issued
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const f32x4 x = stg[p];
        const f32x2 xa = f32x2{x[0], x[1]}, xb = f32x2{x[2], x[3]};
        const unsigned ha = x3_cvt2(xa), hb = x3_cvt2(xb);
        const f32x2 ra = x3_sub_bf16x2(xa, ha), rb = x3_sub_bf16x2(xb, hb);
        const unsigned ma = x3_cvt2(ra), mb = x3_cvt2(rb);
        const unsigned la = x3_cvt2(x3_sub_bf16x2(ra, ma)), lb = x3_cvt2(x3_sub_bf16x2(rb, mb));
        unsigned char *dst = lds + plane_off(p);
        *reinterpret_cast<u32x2 *>(dst) = u32x2{ha, hb};
        *reinterpret_cast<u32x2 *>(dst + TP) = u32x2{ma, mb};
        *reinterpret_cast<u32x2 *>(dst + 2 * TP) = u32x2{la, lb};
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) load_pass(p, nx);
    cv = nx;
    nx = j > 0 ? after : str_next(nx);
    // ring slots (byte offsets) of lines j-1, j, j+1
    int o_c = 0, o_lo = T3, o_hi = 2 * T3;
    X3R_STAMP();    // 1: prologue issued
    lds_barrier();
    X3R_STAMP();    // 2: prologue done

    f32x16 acc[WM];
    unsigned rowb[WM], x4b[WM];         // this lane's pixel row of a plane (bytes), its swizzle term
#pragma unroll
    for (int a = 0; a < WM; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
        const int px = (wp * WM + a) * 32 + l31;
        rowb[a] = (unsigned)(px * PITCH);
        x4b[a] = (unsigned)((line_swz<NC>(px) ^ h) << 4);
    }

    // One step: the MFMAs of one tap over the line tile at rd (6 x WM per k-step: terms lh hl mm mh hm
    // hh), and with JOB the split of the staged line into the slot at wr_off plus the loads of the
    // line after it, one piece behind each MFMA.
    auto run_step = [&](auto slot_c, auto job_c, auto first_c, int rd, int wr_off, const Str &ldn) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_c)::value;
        constexpr bool JOB = decltype(job_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;    // the block's first step: fetches the other slots' weights
        constexpr int OT1 = SL == 0 ? 1 : 0, OT2 = SL == 2 ? 1 : 2;
        const unsigned char *tb = lds + rd;
        unsigned char *nb = lds + wr_off;
        const unsigned char *rowp[WM];
        unsigned x4[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            rowp[a] = tb + x3r_opaque(rowb[a]);
            x4[a] = x3r_opaque(x4b[a]);
        }
        // operand fragments of a k-step: planes mid and lo are read again (for the next k-step) right
        // behind their last MFMA of this one, plane hi — used by the last MFMA — alternates between two sets
        bf16x8 xh[2][WM], xm[WM], xl[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            xh[0][a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + x4[a]);
            xl[a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + 2 * TP + x4[a]);
            xm[a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + TP + x4[a]);
        }
        f32x2 ra, rb;
        unsigned ha = 0, hb = 0, ma = 0, mb = 0;
        // the conversion of staged float4 p in six pieces (few values live between pieces)
        auto piece = [&](int p, int q) __attribute__((always_inline)) {
            unsigned char *dst = nb + plane_off(p);
            if (q == 0) {
                const f32x2 xa = f32x2{stg[p][0], stg[p][1]};
                ha = x3_cvt2(xa);
                ra = x3_sub_bf16x2(xa, ha);
            } else if (q == 1) {
                const f32x2 xb = f32x2{stg[p][2], stg[p][3]};
                hb = x3_cvt2(xb);
                rb = x3_sub_bf16x2(xb, hb);
                *reinterpret_cast<u32x2 *>(dst) = u32x2{ha, hb};
            } else if (q == 2) {
                ma = x3_cvt2(ra);
                ra = x3_sub_bf16x2(ra, ma);
            } else if (q == 3) {
                mb = x3_cvt2(rb);
                rb = x3_sub_bf16x2(rb, mb);
                *reinterpret_cast<u32x2 *>(dst + TP) = u32x2{ma, mb};
            } else if (q == 4) {
                *reinterpret_cast<u32x2 *>(dst + 2 * TP) = u32x2{x3_cvt2(ra), x3_cvt2(rb)};
            } else {
                load_pass(p, ldn);
            }
        };
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int jj = 0; jj < 6 * WM; ++jj) {
                const int a = jj % WM, term = jj / WM;
                // (weight plane, operand plane): small terms first — lh hl mm mh hm hh
                const int pw = term == 0 ? 2 : (term == 1 || term == 4 || term == 5 ? 0 : 1);
                const int px = term == 1 ? 2 : (term == 0 || term == 3 || term == 5 ? 0 : 1);
                const bf16x8 xo = px == 0 ? xh[s & 1][a] : (px == 1 ? xm[a] : xl[a]);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[pw][SL][s], xo, acc[a], 0, 0, 0);
                if (s + 1 < KS) {
                    const unsigned o = (unsigned)((s + 1) << 5) ^ x4[a];
                    if (term == 0) xh[(s + 1) & 1][a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + o);
                    if (term == 1) xl[a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + 2 * TP + o);
                    if (term == 4) xm[a] = *reinterpret_cast<const bf16x8 *>(rowp[a] + TP + o);
                }
                if (FIRST && jj < 6) w[jj % 3][jj < 3 ? OT1 : OT2][s] = w_frag(jj % 3, jj < 3 ? OT1 : OT2, s);
                if (JOB) {
                    if (WM == 1) {
                        if (s % CSTRIDE == 0) piece(s / CSTRIDE, jj);
                    } else if (jj >= 6) {
                        piece(s, jj - 6);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using J1 = std::true_type;
    using J0 = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // after a step that hosted a split: the ring advances, the staged line becomes the next one
    auto job_done = [&]() __attribute__((always_inline)) {
        cv = nx;
        nx = str_next(nx);
        X3R_STAMP();
        lds_barrier();          // the new line is complete in LDS before the step that reads it first
        X3R_STAMP();
    };

    // ---- epilogue from the accumulators: v = acc*sc + sh [+ res] -> [relu] -> [mask] -> global
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(d.out, tbytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(d.epi1.res, tbytes);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(d.epi1.mask, tbytes);
    unsigned vb[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
        const int px = (wp * WM + a) * 32 + l31;
        vb[a] = (unsigned)(px * d.pstride * NC + 32 * cg + 4 * h) * 4u;
    }
    // The residual (or, without one, the mask) of an output line is requested when the line's steps
    // begin and has landed when the epilogue needs it; with both tensors the mask is read in the
    // epilogue.  PFA pixel tiles of a wave are covered (register budget).
    constexpr int PFA0 = NC == 64 ? X3R_PFA64 : X3R_PFA128, PFA = PFA0 < WM ? PFA0 : WM;
    // (all wave-uniform) a bit tensor, when given, stands for the mask tensor
    const bool has_res = d.epi1.res != nullptr, bits_in = d.epi1.mask_bits != nullptr, bits_out = d.epi1.bits_out != nullptr;
    const bool masked = !bits_in && d.epi1.mask != nullptr;
    const unsigned bbytes = (unsigned)((size_t)d.N * d.img_pixels * GC * 4);
    const __amdgpu_buffer_rsrc_t rbi = make_rsrc(d.epi1.mask_bits, bbytes);
    const __amdgpu_buffer_rsrc_t rbo = make_rsrc(d.epi1.bits_out, bbytes);
    unsigned bo[WM], mbits[WM];         // this lane's word of a pixel tile (byte offset), the mask bits of the current line
#pragma unroll
    for (int a = 0; a < WM; ++a) {
        const int px = (wp * WM + a) * 32 + l31;
        bo[a] = (unsigned)(px * d.pstride * GC + cg) * 4u;
        mbits[a] = 0u;
    }
    const __amdgpu_buffer_rsrc_t rpf = make_rsrc(has_res ? d.epi1.res : d.epi1.mask, tbytes);
    f32x4 pf[PFA > 0 ? PFA : 1][4];
    auto prefetch = [&](unsigned soff, int len) __attribute__((always_inline)) {
        if (bits_in) {          // (a pixel's words sit at 1/32 of its activations' byte offset)
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                const int px = (wp * WM + a) * 32 + (int)x3r_opaque((unsigned)l31);
                const unsigned vo = px < len ? x3r_opaque(bo[a]) : ERF_OOB;
                mbits[a] = __builtin_amdgcn_raw_buffer_load_b32(rbi, (int)vo, (int)(soff >> 5), 0);
            }
        }
        if (PFA == 0 || (!has_res && !masked)) return;
#pragma unroll
        for (int a = 0; a < PFA; ++a) {
            const int px = (wp * WM + a) * 32 + (int)x3r_opaque((unsigned)l31);
            const unsigned vo = px < len ? x3r_opaque(vb[a]) : ERF_OOB;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                pf[a][gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpf, (int)vo, (int)(soff + 32u * gq), 0));
        }
    };
    auto finish_t = [&](auto relu_t, auto scale_t, auto res_t, unsigned soff, int len) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_t)::value, SCALE = decltype(scale_t)::value, RES = decltype(res_t)::value;
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            const int px = (wp * WM + a) * 32 + (int)x3r_opaque((unsigned)l31);
            const unsigned vo = px < len ? x3r_opaque(vb[a]) : ERF_OOB;
            f32x4 rr[4];
            if (RES) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    rr[gq] = a < PFA ? pf[a < PFA ? a : 0][gq]
                                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (int)vo, (int)(soff + 32u * gq), 0));
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ch = 32 * cg + 8 * gq + 4 * (int)x3r_opaque((unsigned)h);
                f32x4 v = f32x4{acc[a][4 * gq], acc[a][4 * gq + 1], acc[a][4 * gq + 2], acc[a][4 * gq + 3]};
                const f32x4 sh = *reinterpret_cast<const f32x4 *>(cst + NC + ch);
                if (SCALE) v = v * *reinterpret_cast<const f32x4 *>(cst + ch) + sh;
                else v = v + sh;
                if (RES) v = v + rr[gq];
                if (RELU) v = __builtin_elementwise_max(v, f32x4{0.0f, 0.0f, 0.0f, 0.0f});
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][4 * gq + e] = v[e];
            }
            if (masked) {       // (the residual's registers are free again)
                f32x4 mm[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    mm[gq] = (!RES && a < PFA) ? pf[a < PFA ? a : 0][gq]
                                               : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rmk, (int)vo, (int)(soff + 32u * gq), 0));
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[a][4 * gq + e] = mm[gq][e] > 0.0f ? acc[a][4 * gq + e] : 0.0f;
            }
            if (bits_in) {
                // bit 8 gq + 4 h + e of the word belongs to this lane's element (gq, e)
                const unsigned m = mbits[a] >> (4u * x3r_opaque((unsigned)h));
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (a bit cast applied to a vector's element reads element 0: go through a scalar)
                        const float v = acc[a][4 * gq + e];
                        acc[a][4 * gq + e] = ((m >> (8 * gq + e)) & 1u) ? v : 0.0f;
                    }
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 v = f32x4{acc[a][4 * gq], acc[a][4 * gq + 1], acc[a][4 * gq + 2], acc[a][4 * gq + 3]};
                buf_store4_soff(__builtin_bit_cast(u32x4, v), rout, (int)vo, (int)(soff + 32u * gq));
            }
            if (bits_out) {
                unsigned wbits = 0u;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) wbits |= (acc[a][4 * gq + e] > 0.0f ? 1u : 0u) << (8 * gq + e);
                wbits <<= 4u * x3r_opaque((unsigned)h);
                wbits |= (unsigned)__shfl_xor((int)wbits, 32);          // the other half of the 32 channels
                const unsigned vw = (px < len && x3r_opaque((unsigned)h) == 0u) ? x3r_opaque(bo[a]) : ERF_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(wbits, rbo, (int)vw, (int)(soff >> 5), 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
        }
    };
    const int ekey = (d.epi1.relu ? 4 : 0) | (d.epi1.bn.on ? 2 : 0) | (d.epi1.res != nullptr ? 1 : 0);
    auto finish = [&](unsigned soff, int len) __attribute__((always_inline)) {
        using T1 = std::true_type;
        using T0 = std::false_type;
        switch (ekey) {         // wave-uniform
        case 0: finish_t(T0{}, T0{}, T0{}, soff, len); break;
        case 1: finish_t(T0{}, T0{}, T1{}, soff, len); break;
        case 2: finish_t(T0{}, T1{}, T0{}, soff, len); break;
        case 3: finish_t(T0{}, T1{}, T1{}, soff, len); break;
        case 4: finish_t(T1{}, T0{}, T0{}, soff, len); break;
        case 5: finish_t(T1{}, T0{}, T1{}, soff, len); break;
        case 6: finish_t(T1{}, T1{}, T0{}, soff, len); break;
        default: finish_t(T1{}, T1{}, T1{}, soff, len); break;
        }
    };

    // ---- the block's outputs
    Chain ok = chain_at(oc);
    // one output line: its steps, the epilogue, the cursor's advance.  FIRST: the block's first output
    auto output_line = [&](auto first_c, bool more) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        const bool lo_valid = j > 0, hi_valid = j + 1 < Lc;
        prefetch(ok.base + (unsigned)j * lstep, ok.len);
        if (FIRST) {
            // the first centre step fetches the other taps' weights; it hosts the split of line j-1, or
            // (no line below) of the line the steps after it need
            if (cv.len != 0) {
                run_step(S1{}, J1{}, J1{}, o_c, lo_valid ? o_lo : o_hi, nx);
                job_done();
            } else {
                run_step(S1{}, J0{}, J1{}, o_c, o_hi, nx);
            }
        } else if (!lo_valid && cv.len != 0) {      // a run's first output: the next line is split here
            run_step(S1{}, J1{}, J0{}, o_c, o_hi, nx);
            job_done();
        } else {
            run_step(S1{}, J0{}, J0{}, o_c, o_hi, nx);
        }
        if (lo_valid) {
            if (cv.len != 0) {          // line j+1, or (chain end) the first line of the next run
                run_step(S0{}, J1{}, J0{}, o_lo, o_hi, nx);
                job_done();
            } else {
                run_step(S0{}, J0{}, J0{}, o_lo, o_hi, nx);
            }
        }
        if (hi_valid) run_step(S2{}, J0{}, J0{}, o_hi, o_hi, nx);
        X3R_STAMP();
        finish(ok.base + (unsigned)j * lstep, ok.len);
        X3R_STAMP();
        const int t = o_lo;
        o_lo = o_c; o_c = o_hi; o_hi = t;
        if (hi_valid) {
            ++j;
        } else {
            ++oc; j = 0;
            if (more) ok = chain_at(oc);
        }
    };
    output_line(std::true_type{}, u0 + 1 < u1);
    for (int u = u0 + 1; u < u1; ++u) output_line(std::false_type{}, u + 1 < u1);
}
