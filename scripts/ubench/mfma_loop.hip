// Micro-benchmark: what limits a v_mfma_f32_32x32x2_f32 loop shaped like the igemm K-step?
// Variants: 0 = bare MFMAs; 1 = + 4 ds_read_b128 per 16 MFMAs; 2 = + barrier per step;
//           3 = + ds_write staging per step (double buffer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR, int WM, int WN>
__global__ __launch_bounds__(256, 2) void k(float *out, int steps)
{
    __shared__ __attribute__((aligned(16))) float As[2][128 * 32];
    __shared__ __attribute__((aligned(16))) float Bs[2][128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
    for (int i = tid; i < 128 * 32; i += 256) { As[0][i] = i * 1e-4f; As[1][i] = i * 2e-4f; }
    for (int i = tid; i < 128 * 36; i += 256) { Bs[0][i] = i * 1e-4f; Bs[1][i] = i * 3e-4f; }
    __syncthreads();
    f32x16 acc[WM][WN];
    for (int a = 0; a < WM; ++a) for (int b = 0; b < WN; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
    f32x4 af[WM], bf[WN];
    for (int a = 0; a < WM; ++a) af[a] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + a);
    for (int b = 0; b < WN; ++b) bf[b] = f32x4{1.f, .5f, .25f, 4.f} * (float)(lane + b);
    f32x4 stage[4];
    for (int i = 0; i < 4; ++i) stage[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(tid + i);
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (VAR >= 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int slot = tid + i * 256, r = slot >> 3, ac = slot & 7;
                *reinterpret_cast<f32x4 *>(&As[buf ^ 1][r * 32 + ((ac ^ ((r >> 1) & 7)) << 2)]) = stage[i];
                *reinterpret_cast<f32x4 *>(&Bs[buf ^ 1][(slot >> 3) * 36 + (slot & 7) * 4]) = stage[i];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (VAR >= 1) {
#pragma unroll
                for (int a = 0; a < WM; ++a) {
                    const int row = ((wave >> 1) * WM + a) * 32 + l31;
                    af[a] = *reinterpret_cast<const f32x4 *>(&As[buf][(row & 127) * 32 + (((2 * q + h) ^ ((row >> 1) & 7)) << 2)]);
                }
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    const int col = ((wave & 1) * WN + b) * 32 + l31;
                    bf[b] = *reinterpret_cast<const f32x4 *>(&Bs[buf][(col & 127) * 36 + q * 8 + h * 4]);
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][m], bf[b][m], acc[a][b], 0, 0, 0);
        }
        if (VAR >= 2) __syncthreads();
    }
    float sum = 0;
    for (int a = 0; a < WM; ++a) for (int b = 0; b < WN; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int VAR, int WM, int WN>
void run(const char *name, int grid, int steps, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<VAR, WM, WN><<<grid, 256>>>(out, steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<VAR, WM, WN><<<grid, 256>>>(out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid * 4 * steps * 16 * WM * WN * 4096.0;
    printf("%-28s grid %4d steps %3d: %8.1f us  %6.1f TFLOP/s\n", name, grid, steps, ms * 1e3, flops / ms / 1e9);
}


// Shaped like pair_f32_kernel<64,128>: 64 x 128 tile, KC = 16 (two fragment groups and one barrier
// per step), wave tile 64 x 32, three blocks per CU (LDS padded to the pair kernel's 52.5 KB).
template <int VAR>
__global__ __launch_bounds__(256, 3) void kp(float *out, int steps)
{
    __shared__ __attribute__((aligned(16))) float Ts[65 * 128];
    __shared__ __attribute__((aligned(16))) float Bs[2][128 * 20];
    float *As = Ts;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
    for (int i = tid; i < 65 * 128; i += 256) Ts[i] = i * 1e-4f;
    for (int i = tid; i < 128 * 20; i += 256) { Bs[0][i] = i * 1e-4f; Bs[1][i] = i * 3e-4f; }
    __syncthreads();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    f32x4 af[2], bf;
    for (int a = 0; a < 2; ++a) af[a] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + a);
    bf = f32x4{1.f, .5f, .25f, 4.f} * (float)(lane);
    f32x4 stage[3];
    for (int i = 0; i < 3; ++i) stage[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(tid + i);
#pragma unroll 2
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (VAR >= 3) {
            const int r = tid >> 2, ac = tid & 3;
            *reinterpret_cast<f32x4 *>(&As[(buf ^ 1) * 1024 + r * 16 + ((ac ^ ((r >> 2) & 3)) << 2)]) = stage[0];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int slot = tid + i * 256;
                *reinterpret_cast<f32x4 *>(&Bs[buf ^ 1][(slot >> 2) * 20 + (slot & 3) * 4]) = stage[1 + i];
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (VAR >= 1) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int row = a * 32 + l31;
                    af[a] = *reinterpret_cast<const f32x4 *>(&As[buf * 1024 + row * 16 + (((2 * q + h) ^ ((row >> 2) & 3)) << 2)]);
                }
                bf = *reinterpret_cast<const f32x4 *>(&Bs[buf][(wave * 32 + l31) * 20 + q * 8 + h * 4]);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][m], bf[m], acc[a], 0, 0, 0);
        }
        if (VAR >= 2) __syncthreads();
    }
    float sum = 0;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int VAR>
void runp(const char *name, int grid, int steps, int reps, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kp<VAR><<<grid, 256>>>(out, steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) kp<VAR><<<grid, 256>>>(out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = (double)grid * 4 * steps * 16 * 4096.0;
    printf("%-28s grid %4d steps %3d reps %3d: %8.1f us  %6.1f TFLOP/s\n", name, grid, steps, reps, ms * 1e3, flops / ms / 1e9);
}

int main(int argc, char **argv)
{
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    const int steps = 48;
    for (int reps : {5, 100}) {
        runp<0>("pair-shape bare", 768, 48, reps, out);
        runp<1>("pair-shape +ds_read", 768, 48, reps, out);
        runp<2>("pair-shape +barrier", 768, 48, reps, out);
        runp<3>("pair-shape +ds_write", 768, 48, reps, out);
        runp<3>("pair-shape +all, 10x longer", 768, 480, reps, out);
    }
    if (argc > 1) return 0;
    for (int grid : {256, 512, 768, 1024}) {
        run<0, 2, 2>("bare 2x2", grid, steps, out);
        run<1, 2, 2>("+ds_read 2x2", grid, steps, out);
        run<2, 2, 2>("+barrier 2x2", grid, steps, out);
        run<3, 2, 2>("+ds_write 2x2", grid, steps, out);
        run<0, 3, 1>("bare 3x1", grid, steps, out);
        run<3, 3, 1>("+all 3x1", grid, steps, out);
        run<0, 1, 1>("bare 1x1", grid, steps, out);
        run<3, 1, 1>("+all 1x1", grid, steps, out);
    }
    return 0;
}
