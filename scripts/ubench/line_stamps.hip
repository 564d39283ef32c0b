// Diagnostic: where a block of the bf16 line kernel spends its cycles (s_memtime stamps of wave 0).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DERF_LINE_STAMPS -I erfnet_pytorch_amd/csrc \
//        scripts/ubench/line_stamps.hip -o scripts/ubench/line_stamps
#include "line_bf16.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NC, int BM, bool PAIR>
void run(int N, int H, int W, int dil2)
{
    const size_t P = (size_t)N * H * W;
    bf16_t *x, *y, *res, *wp;
    float *w;
    unsigned long long *st;
    CK(hipMalloc(&x, P * NC * 2)); CK(hipMalloc(&y, P * NC * 2)); CK(hipMalloc(&res, P * NC * 2));
    CK(hipMalloc(&wp, 2 * 3 * NC * NC * 2)); CK(hipMalloc(&w, 2 * 3 * NC * NC * 4));
    CK(hipMemset(x, 0x3c, P * NC * 2)); CK(hipMemset(res, 0x3c, P * NC * 2)); CK(hipMemset(w, 0, 2 * 3 * NC * NC * 4));
    CK(hipMemset(wp, 0x3c, 2 * 3 * NC * NC * 2));
    LineDesc d;
    memset(&d, 0, sizeof(d));
    d.in = x; d.wp1 = wp; d.wp2 = wp + 3 * NC * NC; d.out = y;
    d.epi1.relu = 1; d.epi2.relu = 1;
    (PAIR ? d.epi2 : d.epi1).res = res;
    d.N = N; d.img_pixels = H * W; d.lines_per_img = W; d.L = H; d.pstride = W; d.qstride = 1; d.segs = 1;
    d.nlines = N * W;
    for (int t = 0; t < 3; ++t) { d.s1[t] = t - 1; d.s2[t] = (t - 1) * dil2; }
    const int grid = d.nlines < 256 ? d.nlines : 256;
    CK(hipMalloc(&st, (size_t)grid * 32 * 8));
    CK(hipMemset(st, 0, (size_t)grid * 32 * 8));
    d.stamps = nullptr;
    for (int i = 0; i < 3; ++i) line_bf16_kernel<NC, BM, PAIR><<<grid, 256>>>(d);
    d.stamps = st;
    line_bf16_kernel<NC, BM, PAIR><<<grid, 256>>>(d);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * 32);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    printf("NC %d BM %d pair %d: N %d H %d W %d, %d blocks; cycles since block start (median block)\n", NC, BM, (int)PAIR, N, H, W, grid);
    // per stamp index: median over blocks of (stamp - stamp0)
    for (int i = 1; i < 32; ++i) {
        std::vector<unsigned long long> v;
        for (int b = 0; b < grid; ++b) if (h[(size_t)b * 32 + i]) v.push_back(h[(size_t)b * 32 + i] - h[(size_t)b * 32]);
        if (v.empty()) break;
        std::sort(v.begin(), v.end());
        printf("  stamp %2d: median %8llu  min %8llu  max %8llu  (n=%zu)\n", i, v[v.size() / 2], v.front(), v.back(), v.size());
    }
    // spread of block start times
    unsigned long long lo = ~0ull, hi = 0, endhi = 0;
    for (int b = 0; b < grid; ++b) {
        lo = std::min(lo, h[(size_t)b * 32]); hi = std::max(hi, h[(size_t)b * 32]);
        for (int i = 0; i < 32; ++i) endhi = std::max(endhi, h[(size_t)b * 32 + i]);
    }
    printf("  block starts spread %llu cycles; first start -> last stamp %llu cycles\n", hi - lo, endhi - lo);
}

int main()
{
    run<128, 128, true>(1, 128, 256, 2);
    run<128, 64, true>(6, 64, 128, 2);
    run<64, 256, true>(1, 256, 512, 1);
    run<128, 128, false>(1, 128, 256, 0);
    return 0;
}
