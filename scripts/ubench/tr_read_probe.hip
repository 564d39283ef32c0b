// Probe of ds_read_b64_tr_b16: which element does each lane receive?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long *out)
{
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) tile[i] = (unsigned short)i;      // value = row*64 + col
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // group g reads the 4 x 16 block at rows 4g.., cols 16g..: lane supplies row q, columns 4p..4p+3
    auto v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3))) *)(tile + (4 * g + q) * 64 + 16 * g + 4 * p));
    out[lane] = __builtin_bit_cast(unsigned long long, v);
}
int main()
{
    unsigned long long *d, h[64];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) { unsigned v = (h[l] >> (16 * e)) & 0xffff; printf(" (r%u,c%u)", v / 64, v % 64); }
        printf("\n");
    }
    return 0;
}
