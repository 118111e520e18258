// Dev probe: pure v_mfma_f32_32x32x16_f16 rate with 8 accumulators per wave, 8 waves per CU (the register shape of the round-2 large-problem kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(seed + threadIdx.x * 0.001f + e); y[e] = (_Float16)(seed * 0.5f + e * 0.25f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 6; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
        asm volatile("" : "+v"(x), "+v"(y));
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC> void run(float* out, int blocks, int threads)
{
    const int iters = 128;   // 128 * 6 * NACC MFMAs per wave
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    probe<NACC><<<blocks, threads>>>(out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) probe<NACC><<<blocks, threads>>>(out, iters, 1.0f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 5;
    const double fl = (double)blocks * (threads / 64) * iters * 6 * NACC * 2.0 * 32 * 32 * 16;
    printf("NACC=%d blocks=%d threads=%d: %.3f ms  %.0f TFLOP/s executed\n", NACC, blocks, threads, ms, fl / ms / 1e9);
}
int main()
{
    float* out; hipMalloc(&out, 4 * 512 * 2048);
    run<8>(out, 256, 512); run<8>(out, 252, 512); run<8>(out, 512, 512); run<4>(out, 256, 512); run<8>(out, 256, 256); run<4>(out, 1024, 256);
    return 0;
}
