// Diagnostic: where the two roles of line2_bf16_kernel spend a period (s_memtime stamps of each
// role's first wave).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DERF_LINE_STAMPS -I erfnet_pytorch_amd/csrc \
//        scripts/ubench/line2_stamps.hip -o scripts/ubench/line2_stamps
#include "line2_bf16.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NC, int BM>
void run(int N, int H, int W, int dil2, bool with_mid, bool with_res, bool res_first)
{
    const size_t P = (size_t)N * H * W;
    bf16_t *x, *y, *res, *wp, *mid;
    unsigned long long *st;
    CK(hipMalloc(&x, P * NC * 2)); CK(hipMalloc(&y, P * NC * 2)); CK(hipMalloc(&res, P * NC * 2)); CK(hipMalloc(&mid, P * NC * 2));
    CK(hipMalloc(&wp, 2 * 3 * NC * NC * 2));
    CK(hipMemset(x, 0x3c, P * NC * 2)); CK(hipMemset(res, 0x3c, P * NC * 2));
    CK(hipMemset(wp, 0x3c, 2 * 3 * NC * NC * 2));
    LineDesc d;
    memset(&d, 0, sizeof(d));
    d.in = x; d.wp1 = wp; d.wp2 = wp + 3 * NC * NC; d.out = y; d.mid = with_mid ? mid : nullptr;
    d.epi1.relu = 1; d.epi2.relu = 1;
    if (with_res) (res_first ? d.epi1 : d.epi2).res = res;
    d.N = N; d.img_pixels = H * W; d.lines_per_img = W; d.L = H; d.pstride = W; d.qstride = 1; d.segs = 1;
    d.nlines = N * W;
    for (int t = 0; t < 3; ++t) { d.s1[t] = t - 1; d.s2[t] = (t - 1) * dil2; }
    const int grid = d.nlines < 256 ? d.nlines : 256;
    CK(hipMalloc(&st, (size_t)grid * 128 * 8));
    CK(hipMemset(st, 0, (size_t)grid * 128 * 8));
    d.stamps = nullptr;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) line2_bf16_kernel<NC, BM><<<grid, 512>>>(d);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) line2_bf16_kernel<NC, BM><<<grid, 512>>>(d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    d.stamps = st;
    line2_bf16_kernel<NC, BM><<<grid, 512>>>(d);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * 128);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    printf("NC %d BM %d: N %d H %d W %d mid %d res %d (first %d), %d blocks, %.1f us per launch without stamps\n", NC, BM, N, H, W,
           (int)with_mid, (int)with_res, (int)res_first, grid, ms * 1e3 / 20);
    for (int role = 0; role < 2; ++role) {
        printf(" role %c: ticks since the role's first stamp (median block), delta\n", role ? 'B' : 'A');
        unsigned long long prev = 0;
        for (int i = 1; i < 64; ++i) {
            std::vector<unsigned long long> v;
            for (int b = 0; b < grid; ++b) {
                const unsigned long long *q = &h[((size_t)b * 2 + role) * 64];
                if (q[i]) v.push_back(q[i] - q[0]);
            }
            if (v.size() < (size_t)grid / 2) break;
            std::sort(v.begin(), v.end());
            const unsigned long long m = v[v.size() / 2];
            printf("  %2d: %7llu (+%6llu)\n", i, m, m - prev);
            prev = m;
        }
    }
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(res)); CK(hipFree(mid)); CK(hipFree(wp)); CK(hipFree(st));
}

int main()
{
    run<64, 128>(6, 128, 256, 1, true, false, false);      // forward pair, intermediate kept
    run<64, 128>(6, 128, 256, 1, true, true, true);        // conv1x3_2 (+ residual) with the next block's conv3x1_1
    run<128, 64>(6, 64, 128, 2, true, false, false);
    return 0;
}
