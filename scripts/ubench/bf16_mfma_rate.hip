// Calibration: what a wave can sustain on v_mfma_f32_32x32x16_bf16 in the shapes the line kernels use.
// One block of 256 threads per CU (WPS = 1: one wave per SIMD) or two (WPS = 2), every wave runs
// ITER x 96 MFMAs on two alternating accumulators with operands that never leave registers, and
//   mode 0: nothing else
//   mode 1: one 16-byte LDS read per two MFMAs (the operand-fragment traffic of line_x3r_kernel), issued
//           DIST MFMA pairs ahead of its first use (the kernel: six pairs)
//   mode 2: mode 1 + four independent vector instructions behind every MFMA
//   mode 4: mode 0 + NV independent fp32 multiply-adds behind every MFMA (NV = DIST argument)
//   mode 5: mode 0 + NV scalar instructions behind every MFMA
//   mode 10..24 (second argument given): mode 0 + NV copies of one vector instruction kind behind every MFMA
//   mode 8: mode 0 + one 8-byte LDS write per two MFMAs
//   mode 3: mode 0 with the weight operand taken from a different register set every MFMA (24 sets)
// `bf16_mfma_rate calib` prints one JSON line for bench.py (bare loop, constant against random operands).
// Prints wall time per MFMA and SIMD, the s_memtime ticks per MFMA, and the clock both imply.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/bf16_mfma_rate.hip -o scripts/ubench/bf16_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WPS, int DIST>
__global__ __launch_bounds__(256, WPS) void rate_kernel(const u32x4 *src, float *out, unsigned long long *ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[(WPS == 1 ? 128 : 72) * 1024];    // (one or two blocks fit a CU)
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 1024 / 16; i += 256) reinterpret_cast<u32x4 *>(lds)[i] = src[i & 1023];
    __syncthreads();
    constexpr int NW = MODE == 3 ? 24 : 2;
    constexpr int NX = 8;
    bf16x8 w[NW], x[NX];
    for (int i = 0; i < NW; ++i) w[i] = __builtin_bit_cast(bf16x8, src[(tid + 64 * i) & 1023]);
    for (int i = 0; i < NX; ++i) x[i] = __builtin_bit_cast(bf16x8, src[(tid + 7 + 4 * i) & 1023]);
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    float ff[8]; unsigned ii[8]; unsigned sc[4] = {1, 2, 3, 4};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[8]; const f32x2 pk1 = {1.0f, 2.0f};
    for (int v = 0; v < 8; ++v) { ff[v] = (float)(tid + v); ii[v] = tid * 3 + v; pk[v] = f32x2{(float)tid, (float)v}; }
    float f0 = (float)tid, f1 = 1.0f, f2 = 2.0f, f3 = 3.0f;
    const unsigned char *rp = lds + (tid & 63) * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 96; ++m) {
            acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m % NW], x[(m >> 1) % NX], acc[m & 1], 0, 0, 0);
            if (MODE == 1 || MODE == 2) {
                if ((m & 1) == 1) x[((m >> 1) + DIST) % NX] = *reinterpret_cast<const bf16x8 *>(rp + ((m * 1024) & 0xfc00));
            }
            if (MODE == 2) {
                f0 = f0 * 1.0001f + f1; f1 = f1 * 0.9999f + f2; f2 = f2 * 1.0002f + f3; f3 = f3 * 0.9998f + 1.0f;
            }
            if (MODE == 4) {
#pragma unroll
                for (int v = 0; v < DIST; ++v) ff[v] = ff[v] * 1.0001f + 0.5f;
            }
            if (MODE == 5) {
#pragma unroll
                for (int v = 0; v < DIST; ++v) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc[v % 4]));
            }
            if (MODE >= 10) {       // NV copies of one instruction kind, each on its own register
#pragma unroll
                for (int v = 0; v < DIST; ++v) {
                    if (MODE == 10) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ii[v]) : "v"(ii[7]));
                    if (MODE == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ii[v]) : "v"(ii[7]));
                    if (MODE == 12) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(ii[v]));
                    if (MODE == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(ii[v]) : "v"(ii[7]));
                    if (MODE == 14) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(ff[v]) : "v"(ff[7]));
                    if (MODE == 15) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(ff[v]) : "v"(ff[7]));
                    if (MODE == 16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ii[v]) : "v"(ff[v]), "v"(ff[7]));
                    if (MODE == 17) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[v]) : "v"(pk1));
                    if (MODE == 18) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[v]) : "v"(pk1));
                    if (MODE == 19) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(ii[v]) : "v"(ii[7]), "v"(ii[6]));
                    if (MODE == 20) asm volatile("v_max_f32 %0, %0, %1" : "+v"(ff[v]) : "v"(ff[7]));
                    if (MODE == 21) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ii[v]) : "v"(ii[7]), "v"(ii[6]));
                    if (MODE == 22) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ii[v]) : "v"(ii[7]));
                    if (MODE == 23) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(ii[v]) : "v"(ii[7]));
                    if (MODE == 24) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(ii[v]) : "v"(ii[7]));
                }
            }
            if (MODE == 8) {
                if ((m & 1) == 1) *reinterpret_cast<unsigned long long *>(lds + 65536 + tid * 8 + ((m * 2048) & 0x7800)) = ((unsigned long long)tid << 32) | (unsigned)m;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = f0 + f1 + f2 + f3;
    for (int v = 0; v < 8; ++v) s += ff[v] + (float)ii[v] + pk[v][0] + pk[v][1];
    s += (float)(sc[0] + sc[1] + sc[2] + sc[3]);
    if (MODE == 8) s += (float)lds[65536 + tid];
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

struct Rate { double tflops, ticks_per_mfma, ghz; };
static bool g_quiet = false;

template <int MODE, int WPS, int DIST = 6>
Rate run(const u32x4 *src, float *out, unsigned long long *tk, int iters)
{
    const int grid = 256 * WPS;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) rate_kernel<MODE, WPS, DIST><<<grid, 256>>>(src, out, tk, iters);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) rate_kernel<MODE, WPS, DIST><<<grid, 256>>>(src, out, tk, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), tk, grid * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double n_per_wave = 96.0 * iters;
    const double us = ms * 1e3 / reps;
    const double ns_per_mfma_simd = us * 1e3 / (n_per_wave * WPS);
    const double ticks_per_mfma_simd = (double)h[grid / 2] / (n_per_wave * WPS);
    const double tflops = 256.0 * 4 * WPS * n_per_wave * 32768.0 / (us * 1e-6) / 1e12;
    if (!g_quiet) printf("mode %d dist %d, %d wave(s) per SIMD, %d x 96 MFMAs per wave: %.1f us per launch, %.2f ns and %.1f ticks per MFMA and SIMD "
           "(tick = %.3f ns), %.0f TFLOP/s\n", MODE, DIST, WPS, iters, us, ns_per_mfma_simd, ticks_per_mfma_simd,
           ns_per_mfma_simd / ticks_per_mfma_simd, tflops);
    return Rate{tflops, ticks_per_mfma_simd, ticks_per_mfma_simd / ns_per_mfma_simd};
}

static void fill_random(u32x4 *src)
{   // operands of random sign, exponent around 1, random mantissa (toggling data costs power)
    std::vector<unsigned short> hsrc(1024 * 8);
    unsigned long long z = 88172645463325252ull;
    for (auto &v : hsrc) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (unsigned short)(((z >> 20) & 0x80ff) | ((0x7c + ((z >> 40) & 7)) << 7)); }
    CK(hipMemcpy(src, hsrc.data(), 1024 * 16, hipMemcpyHostToDevice));
}

template <int MODE>
void kinds(const char *name, const u32x4 *src, float *out, unsigned long long *tk, int iters)
{
    printf("%s\n", name);
    run<MODE, 1, 1>(src, out, tk, iters);
    run<MODE, 1, 2>(src, out, tk, iters);
    run<MODE, 1, 4>(src, out, tk, iters);
    run<MODE, 2, 4>(src, out, tk, iters);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 && atoi(argv[1]) > 0 ? atoi(argv[1]) : 9;    // 9 x 96 = 864 MFMAs per wave: a line_x3r launch at 128 channels
    const bool all = argc > 2;
    u32x4 *src; float *out; unsigned long long *tk;
    CK(hipMalloc(&src, 1024 * 16)); CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&tk, 512 * 8));
    CK(hipMemset(src, 0x3c, 1024 * 16));
    if (argc > 1 && !strcmp(argv[1], "calib")) {
        // one JSON line: the bare MFMA loop (one wave per SIMD, 17280 MFMAs per wave) on constant and on
        // random operands — same cycles per MFMA, but the clock the part sustains differs
        g_quiet = true;
        const Rate c = run<0, 1>(src, out, tk, 180);
        fill_random(src);
        const Rate r = run<0, 1>(src, out, tk, 180);
        printf("{\"kernel\": \"v_mfma_f32_32x32x16_bf16 loop, 256 blocks x 4 waves, 17280 MFMAs per wave\", "
               "\"constant_operands_tflops\": %.0f, \"constant_operands_clock_ghz\": %.2f, "
               "\"random_operands_tflops\": %.0f, \"random_operands_clock_ghz\": %.2f, \"cycles_per_mfma\": %.1f}\n",
               c.tflops, c.ghz, r.tflops, r.ghz, r.ticks_per_mfma);
        return 0;
    }
    if (getenv("RATE_RANDOM")) { fill_random(src); printf("random operands\n"); }
    run<0, 1>(src, out, tk, iters);
    run<3, 1>(src, out, tk, iters);
    run<1, 1, 1>(src, out, tk, iters);
    run<1, 1, 2>(src, out, tk, iters);
    run<1, 1>(src, out, tk, iters);
    run<2, 1>(src, out, tk, iters);
    run<4, 1, 1>(src, out, tk, iters);
    run<4, 1, 2>(src, out, tk, iters);
    run<4, 1, 4>(src, out, tk, iters);
    run<4, 1, 8>(src, out, tk, iters);
    run<5, 1, 4>(src, out, tk, iters);
    run<5, 1, 8>(src, out, tk, iters);
    run<8, 1>(src, out, tk, iters);
    run<0, 2>(src, out, tk, iters);
    run<1, 2>(src, out, tk, iters);
    run<2, 2>(src, out, tk, iters);
    run<4, 2, 4>(src, out, tk, iters);
    run<4, 2, 8>(src, out, tk, iters);
    run<8, 2>(src, out, tk, iters);
    if (all) {
        kinds<10>("v_xor_b32", src, out, tk, iters);
        kinds<11>("v_add_u32", src, out, tk, iters);
        kinds<12>("v_lshlrev_b32", src, out, tk, iters);
        kinds<13>("v_mov_b32", src, out, tk, iters);
        kinds<14>("v_sub_f32", src, out, tk, iters);
        kinds<15>("v_fma_f32", src, out, tk, iters);
        kinds<16>("v_cvt_pk_bf16_f32", src, out, tk, iters);
        kinds<17>("v_pk_add_f32", src, out, tk, iters);
        kinds<18>("v_pk_mul_f32", src, out, tk, iters);
        kinds<19>("v_perm_b32", src, out, tk, iters);
        kinds<20>("v_max_f32", src, out, tk, iters);
        kinds<21>("v_and_or_b32", src, out, tk, iters);
        kinds<22>("v_cndmask_b32", src, out, tk, iters);
        kinds<23>("v_pk_add_u16", src, out, tk, iters);
        kinds<24>("v_lshl_add_u32", src, out, tk, iters);
    }
    run<0, 1>(src, out, tk, iters * 20);
    run<2, 1>(src, out, tk, iters * 20);
    run<2, 2>(src, out, tk, iters * 20);
    return 0;
}
