// Diagnostic: where a block of line_x3r_kernel spends its cycles (s_memtime stamps of thread 0).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DERF_LINE_STAMPS -I erfnet_pytorch_amd/csrc \
//        scripts/ubench/x3r_stamps.hip -o scripts/ubench/x3r_stamps
#include "line_x3r.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NC, int BM>
void run(int N, int H, int W, int dil, bool with_res, int blocks_per_cu)
{
    const size_t P = (size_t)N * H * W;
    float *x, *y, *res;
    bf16_t *wp;
    unsigned long long *st;
    CK(hipMalloc(&x, P * NC * 4)); CK(hipMalloc(&y, P * NC * 4)); CK(hipMalloc(&res, P * NC * 4));
    CK(hipMalloc(&wp, 3 * 3 * NC * NC * 2));
    CK(hipMemset(x, 0x3c, P * NC * 4)); CK(hipMemset(res, 0x3c, P * NC * 4)); CK(hipMemset(wp, 0x3c, 3 * 3 * NC * NC * 2));
    LineDesc d;
    memset(&d, 0, sizeof(d));
    d.in = (const bf16_t *)x; d.wp1 = wp; d.out = (bf16_t *)y;
    d.epi1.relu = 1;
    if (with_res) d.epi1.res = res;
    d.N = N; d.img_pixels = H * W; d.lines_per_img = H; d.L = W; d.pstride = 1; d.qstride = W;
    d.segs = (W + BM - 1) / BM;
    d.nlines = N * H;
    for (int t = 0; t < 3; ++t) d.s1[t] = (t - 1) * dil;
    const int ntiles = d.nlines * d.segs;
    const int slots = 256 * blocks_per_cu;
    const int grid = ntiles < slots ? ntiles : slots;
    CK(hipMalloc(&st, (size_t)grid * 48 * 8));
    CK(hipMemset(st, 0, (size_t)grid * 48 * 8));
    d.stamps = nullptr;
    for (int i = 0; i < 3; ++i) line_x3r_kernel<NC, BM><<<grid, 256>>>(d);
    d.stamps = st;
    line_x3r_kernel<NC, BM><<<grid, 256>>>(d);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * 48);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    printf("NC %d BM %d res %d dil %d: N %d H %d W %d, %d tiles on %d blocks; s_memtime ticks since block start (median block), delta\n",
           NC, BM, (int)with_res, dil, N, H, W, ntiles, grid);
    unsigned long long prev = 0;
    for (int i = 1; i < 48; ++i) {
        std::vector<unsigned long long> v;
        for (int b = 0; b < grid; ++b) if (h[(size_t)b * 48 + i]) v.push_back(h[(size_t)b * 48 + i] - h[(size_t)b * 48]);
        if (v.size() < (size_t)grid / 2) break;
        std::sort(v.begin(), v.end());
        printf("  stamp %2d: median %8llu (+%6lld)  min %8llu  max %8llu  (n=%zu)\n", i, v[v.size() / 2],
               (long long)v[v.size() / 2] - (long long)prev, v.front(), v.back(), v.size());
        prev = v[v.size() / 2];
    }
    unsigned long long lo = ~0ull, hi = 0, endhi = 0;
    for (int b = 0; b < grid; ++b) {
        lo = std::min(lo, h[(size_t)b * 48]); hi = std::max(hi, h[(size_t)b * 48]);
        for (int i = 0; i < 48; ++i) endhi = std::max(endhi, h[(size_t)b * 48 + i]);
    }
    printf("  block starts spread %llu ticks; first start -> last stamp %llu ticks\n", hi - lo, endhi - lo);
}

int main()
{
    run<128, 64>(6, 64, 128, 2, false, 1);
    run<64, 64>(6, 128, 256, 1, false, 2);
    run<64, 64>(6, 128, 256, 1, true, 2);
    return 0;
}
