// Dev probe: is s_memtime the shader clock?  ticks of s_memtime per microsecond of s_memrealtime (100 MHz) for a light
// kernel (one wave spinning on SALU), a VALU-heavy and an MFMA-heavy kernel on all CUs.
// hipcc --offload-arch=gfx950 -O3 -o tools/clock_probe.bin tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.37f + i); b[i] = (_Float16)(threadIdx.x * 0.11f - i); }
    float v = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v = fmaf(v, 1.0001f, 0.5f);
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    float s = v;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) sink[0] = s;
}
template <int MODE> void run(const char* tag, int blocks, int iters)
{
    unsigned long long* d; float* sink;
    hipMalloc(&d, 8 * 2 * 4096); hipMalloc(&sink, 4);
    k<MODE><<<blocks, 512>>>(d, sink, iters); hipDeviceSynchronize();
    k<MODE><<<blocks, 512>>>(d, sink, iters); hipDeviceSynchronize();
    unsigned long long h[2 * 4096];
    hipMemcpy(h, d, 8 * 2 * blocks, hipMemcpyDeviceToHost);
    double t = 0, r = 0;
    for (int i = 0; i < blocks; ++i) { t += h[2 * i]; r += h[2 * i + 1]; }
    printf("%-28s blocks %4d: s_memtime ticks / us of s_memrealtime = %.1f  (kernel ~%.0f us)\n", tag, blocks, t / (r / 100.0), r / blocks / 100.0);
}
int main()
{
    run<0>("sleep, 1 block", 1, 20000);
    run<0>("sleep, 512 blocks", 512, 20000);
    run<1>("VALU fma, 512 blocks", 512, 20000);
    run<2>("MFMA 16x16x32, 512 blocks", 512, 40000);
    run<2>("MFMA 16x16x32, 1 block", 1, 40000);
    return 0;
}
