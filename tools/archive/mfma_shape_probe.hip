// Dev probe: sustained rate of v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 on register operands, with operand
// values that do (pseudo-random mantissas) or do not (zeros) toggle the multiplier arrays - separates the issue rate of the
// two shapes from the power-limited clock.  hipcc --offload-arch=gfx950 -O3 -o x.bin tools/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f16x8 fill(unsigned s, int zero)
{
    f16x8 v;
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u;
        v[e] = zero ? (_Float16)0.f : (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 4096.0f));   // 12 random bits in [-0.5, 0.5)
    }
    return v;
}

template <int SHAPE>   // 32: 8 x 32x32x16 accumulators;  16: 32 x 16x16x32 accumulators (same 128 registers, same flops / round)
__global__ __launch_bounds__(512) void probe(float* out, int iters, int zero)
{
    f16x8 x[4], y[4];
    for (int i = 0; i < 4; ++i) { x[i] = fill(threadIdx.x * 7 + i, zero); y[i] = fill(threadIdx.x * 13 + 5 + i, zero); }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[a & 3], y[(a >> 1) & 3], acc[a], 0, 0, 0);
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(x[i]), "+v"(y[i]));
        }
        for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    } else {
        f32x4 acc[32];
        for (int a = 0; a < 32; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)      // 3 x 32 x 16384 flop = 6 x 8 x 32768 flop
#pragma unroll
                for (int a = 0; a < 32; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[a & 3], y[(a >> 2) & 3], acc[a], 0, 0, 0);
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(x[i]), "+v"(y[i]));
        }
        for (int a = 0; a < 32; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE> void run(float* out, int zero)
{
    const int iters = 256, blocks = 256, threads = 512;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int r = 0; r < 3; ++r) probe<SHAPE><<<blocks, threads>>>(out, iters, zero);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 10; ++r) probe<SHAPE><<<blocks, threads>>>(out, iters, zero);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
    const double fl = (double)blocks * (threads / 64) * iters * 6 * 8 * 2.0 * 32 * 32 * 16;
    printf("mfma %s  %s operands: %.3f ms  %.0f TFLOP/s\n", SHAPE == 32 ? "32x32x16" : "16x16x32", zero ? "zero  " : "random", ms, fl / ms / 1e9);
}
int main()
{
    float* out; hipMalloc(&out, 4 * 512 * 2048);
    for (int rep = 0; rep < 2; ++rep) { run<32>(out, 0); run<16>(out, 0); run<32>(out, 1); run<16>(out, 1); }
    return 0;
}
