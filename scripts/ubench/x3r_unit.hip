// line_x3r_kernel against line_x3_kernel: same results (to fp32 rounding: the taps are summed in a
// different order) on a set of geometries, and the time per launch of both.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I erfnet_pytorch_amd/csrc scripts/ubench/x3r_unit.hip \
//        -o scripts/ubench/x3r_unit
#include "line_x3r.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

static int g_fail = 0;

// axis 0: a (3,1) convolution (lines = image rows), axis 1: a (1,3) one (lines = image columns)
template <int NC>
void run(int N, int H, int W, int axis, int dil, bool flip, bool relu, bool res, bool mask, bool bn, int reps)
{
    constexpr int BM = 64;
    const size_t P = (size_t)N * H * W, E = P * NC;
    std::vector<float> hx(E), hw((size_t)3 * NC * NC), hr(E), hm(E), hb(NC), hg(NC), hbe(NC), hmu(NC), hva(NC);
    for (auto &v : hx) v = frand() > -0.2f ? frand() : 0.0f;
    for (auto &v : hw) v = frand() * 0.05f;
    for (auto &v : hr) v = frand();
    for (auto &v : hm) v = frand();
    for (int c = 0; c < NC; ++c) { hb[c] = frand() * 0.1f; hg[c] = 1.0f + 0.2f * frand(); hbe[c] = 0.1f * frand(); hmu[c] = 0.1f * frand(); hva[c] = 1.0f + 0.5f * frand(); }
    float *x, *y0, *y1, *r, *m, *w, *b, *g, *be, *mu, *va;
    bf16_t *wp;
    CK(hipMalloc(&x, E * 4)); CK(hipMalloc(&y0, E * 4)); CK(hipMalloc(&y1, E * 4)); CK(hipMalloc(&r, E * 4)); CK(hipMalloc(&m, E * 4));
    CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&wp, 3 * 3 * NC * NC * 2));
    CK(hipMalloc(&b, NC * 4)); CK(hipMalloc(&g, NC * 4)); CK(hipMalloc(&be, NC * 4)); CK(hipMalloc(&mu, NC * 4)); CK(hipMalloc(&va, NC * 4));
    CK(hipMemcpy(x, hx.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(r, hr.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(m, hm.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), NC * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), NC * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(be, hbe.data(), NC * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(mu, hmu.data(), NC * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(va, hva.data(), NC * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y0, 0xff, E * 4)); CK(hipMemset(y1, 0xff, E * 4));
    // weights W[tap][k][j] = w[tap*NC*NC + k*NC + j]
    LinePackBatch pb;
    memset(&pb, 0, sizeof(pb));
    pb.p[0].w = w; pb.p[0].wp = wp; pb.p[0].NC = NC; pb.p[0].wsk = NC; pb.p[0].wsn = 1; pb.p[0].x3 = 1;
    for (int t = 0; t < 3; ++t) pb.p[0].woff[t] = t * NC * NC;
    line_pack_kernel<<<dim3(3 * NC * NC / 256 / 4, 1, 1), 256>>>(pb);
    LineDesc d;
    memset(&d, 0, sizeof(d));
    d.in = (const bf16_t *)x; d.wp1 = wp;
    d.epi1.bias = b; d.epi1.relu = relu ? 1 : 0;
    if (res) d.epi1.res = r;
    if (mask) d.epi1.mask = m;
    if (bn) { d.epi1.bn.on = 1; d.epi1.bn.gamma = g; d.epi1.bn.beta = be; d.epi1.bn.mean = mu; d.epi1.bn.var = va; d.epi1.bn.eps = 1e-3f; }
    d.N = N; d.img_pixels = H * W;
    if (axis == 0) { d.lines_per_img = H; d.L = W; d.pstride = 1; d.qstride = W; }
    else { d.lines_per_img = W; d.L = H; d.pstride = W; d.qstride = 1; }
    d.segs = (d.L + BM - 1) / BM;
    d.nlines = N * d.lines_per_img;
    for (int t = 0; t < 3; ++t) d.s1[t] = (t - 1) * dil * (flip ? -1 : 1);
    const int ntiles = d.nlines * d.segs;
    const int slots = 256 * (NC == 64 ? 2 : 1);
    const int grid = ntiles < slots ? ntiles : slots;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms0 = 0, ms1 = 0;
    d.out = (bf16_t *)y0;
    line_x3_kernel<NC, BM><<<grid, 256>>>(d);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) line_x3_kernel<NC, BM><<<grid, 256>>>(d);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms0, e0, e1));
    d.out = (bf16_t *)y1;
    line_x3r_kernel<NC, BM><<<grid, 256>>>(d);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) line_x3r_kernel<NC, BM><<<grid, 256>>>(d);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
    CK(hipDeviceSynchronize());
    // the ReLU masks as bit tensors: the mask read from bits must give the very same output, and the
    // bits written for the output must say (output > 0)
    {
        const int GCW = NC / 32;
        std::vector<unsigned> hbits(P * GCW, 0u), hbo(P * GCW, 0u);
        for (size_t px = 0; px < P; ++px)
            for (int c = 0; c < NC; ++c)
                if (hm[px * NC + c] > 0.0f) hbits[px * GCW + c / 32] |= 1u << (c % 32);
        unsigned *bi, *bo_;
        float *y2;
        CK(hipMalloc(&bi, hbits.size() * 4)); CK(hipMalloc(&bo_, hbits.size() * 4)); CK(hipMalloc(&y2, E * 4));
        CK(hipMemcpy(bi, hbits.data(), hbits.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(bo_, 0xff, hbits.size() * 4)); CK(hipMemset(y2, 0xff, E * 4));
        LineDesc d2 = d;
        d2.out = (bf16_t *)y2;
        d2.epi1.bits_out = bo_;
        if (mask) d2.epi1.mask_bits = bi;
        line_x3r_kernel<NC, BM><<<grid, 256>>>(d2);
        CK(hipDeviceSynchronize());
        std::vector<float> h2(E), h1b(E);
        CK(hipMemcpy(h2.data(), y2, E * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1b.data(), y1, E * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hbo.data(), bo_, hbo.size() * 4, hipMemcpyDeviceToHost));
        size_t badv = 0, badb = 0;
        for (size_t i = 0; i < E; ++i) if (memcmp(&h2[i], &h1b[i], 4)) {
            if (badv < 6) { unsigned u; memcpy(&u, &h2[i], 4); printf("  elem %zu (pixel %zu c %zu): mask %g fp32-mask run %g bits run %g (%08x) word %08x\n", i, i / NC, i % NC, hm[i], h1b[i], h2[i], u, hbits[(i / NC) * GCW + (i % NC) / 32]); }
            ++badv;
        }
        for (size_t px = 0; px < P; ++px)
            for (int gw = 0; gw < GCW; ++gw) {
                unsigned want = 0u;
                for (int c = 0; c < 32; ++c) if (h2[px * NC + gw * 32 + c] > 0.0f) want |= 1u << c;
                if (want != hbo[px * GCW + gw]) ++badb;
            }
        if (badv || badb) { ++g_fail; printf("FAIL bits: %zu values differ with the mask read from bits, %zu bit words wrong\n", badv, badb); }
        hipFree(bi); hipFree(bo_); hipFree(y2);
    }
    std::vector<float> h0(E), h1(E);
    CK(hipMemcpy(h0.data(), y0, E * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, E * 4, hipMemcpyDeviceToHost));
    double maxd = 0, scale = 0;
    size_t bad = 0, firstbad = 0;
    for (size_t i = 0; i < E; ++i) {
        const double dd = std::fabs((double)h0[i] - (double)h1[i]);
        if (!(dd <= 1e30)) { if (!bad) firstbad = i; ++bad; continue; }
        scale = std::max(scale, (double)std::fabs(h0[i]));
        if (dd > maxd) maxd = dd;
    }
    for (size_t i = 0; i < E; ++i)
        if (std::fabs((double)h0[i] - (double)h1[i]) > 2e-6 * scale + 1e-7) { if (!bad) firstbad = i; ++bad; }
    const bool ok = bad == 0;
    if (!ok) ++g_fail;
    printf("%s NC %3d N %d %3dx%3d axis %d dil %2d flip %d relu %d res %d mask %d bn %d | grid %3d tiles %4d | old %6.1f us  new %6.1f us | max diff %.3g (scale %.3g) bad %zu",
           ok ? "ok  " : "FAIL", NC, N, H, W, axis, dil, (int)flip, (int)relu, (int)res, (int)mask, (int)bn, grid, ntiles,
           ms0 * 1e3 / reps, ms1 * 1e3 / reps, maxd, scale, bad);
    if (!ok) {
        const size_t pix = firstbad / NC;
        printf(" first bad: elem %zu (n %zu y %zu x %zu c %zu) old %g new %g", firstbad, pix / ((size_t)H * W), (pix / W) % H, pix % W,
               firstbad % NC, h0[firstbad], h1[firstbad]);
    }
    printf("\n");
    hipFree(x); hipFree(y0); hipFree(y1); hipFree(r); hipFree(m); hipFree(w); hipFree(wp);
    hipFree(b); hipFree(g); hipFree(be); hipFree(mu); hipFree(va);
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    // training shapes
    run<64>(6, 128, 256, 0, 1, false, true, false, false, false, 20);
    run<64>(6, 128, 256, 1, 1, false, true, false, false, false, 20);
    run<64>(6, 128, 256, 1, 1, false, true, true, false, false, 20);
    run<64>(6, 128, 256, 0, 1, true, false, false, true, false, 20);
    run<128>(6, 64, 128, 0, 2, false, true, false, false, false, 20);
    run<128>(6, 64, 128, 1, 2, false, true, true, false, false, 20);
    run<128>(6, 64, 128, 0, 16, false, true, false, false, false, 20);
    run<128>(6, 64, 128, 1, 16, true, false, false, true, false, 20);
    if (quick) return g_fail ? 1 : 0;
    // odd shapes: ragged tiles, few lines, chains of one line, every epilogue switch
    run<64>(1, 7, 40, 0, 1, false, true, true, true, true, 2);
    run<64>(2, 24, 40, 1, 2, false, false, true, false, false, 2);
    run<64>(2, 24, 100, 0, 4, true, true, false, true, false, 2);
    run<64>(3, 16, 200, 0, 16, false, true, false, false, true, 2);
    run<64>(1, 128, 256, 0, 1, false, true, false, false, false, 5);
    run<64>(1, 256, 512, 1, 1, false, true, true, false, false, 5);
    run<128>(1, 5, 33, 1, 1, false, true, true, true, true, 2);
    run<128>(2, 12, 70, 0, 3, false, true, false, false, false, 2);
    run<128>(2, 40, 56, 0, 8, true, false, true, true, false, 2);
    run<128>(2, 40, 56, 1, 8, false, true, false, false, false, 2);
    run<128>(1, 64, 128, 0, 4, false, true, false, false, false, 5);
    run<128>(1, 128, 256, 1, 2, false, true, true, false, false, 5);
    run<128>(8, 28, 28, 0, 2, false, true, false, false, true, 5);
    printf(g_fail ? "FAILED %d\n" : "all ok\n", g_fail);
    return g_fail ? 1 : 0;
}
