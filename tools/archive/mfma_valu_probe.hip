// Dev probe: how do MFMA and ordinary VALU work share a SIMD on gfx950?
// Each wave runs `iters` rounds of { NM dependent-free MFMAs (32x32x16 f16), NV fp32 FMAs on private registers };
// W waves per SIMD.  Reports cycles per round per SIMD so that overlap (max) vs serialisation (sum) is visible.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_valu_probe.bin tools/mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, int CHAIN>
__global__ void probe(float* out, int iters, float seed)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(seed + threadIdx.x * 0.001f + e); y[e] = (_Float16)(seed * 0.5f + e * 0.25f); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed + e + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int a = CHAIN ? 0 : (m & 3);                       // CHAIN: every MFMA depends on the previous one
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / (NM > 0 ? NM : 1); ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
        }
        if (NM == 0) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
        }
        asm volatile("" : "+v"(x), "+v"(y));
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int CHAIN> void run(float* out, int waves_per_simd)
{
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;       // one block per CU, waves_per_simd per SIMD
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    probe<NM, NV, CHAIN><<<blocks, threads>>>(out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    probe<NM, NV, CHAIN><<<blocks, threads>>>(out, iters, 1.0f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double us_round_simd = ms * 1e3 / iters;                               // time of one round of ALL waves of a SIMD
    printf("MFMA %2d%s + VALU %3d per wave-round, %d wave(s)/SIMD: %7.3f us per SIMD-round = %6.0f cyc @2.1GHz  (MFMA alone would need %4d cyc)\n",
           NM, CHAIN ? " (chain)" : "        ", NV, waves_per_simd, us_round_simd, us_round_simd * 2100.0, NM * 32 * waves_per_simd);
}

int main()
{
    float* out; hipMalloc(&out, 4 * 256 * 1024);
    for (int w = 1; w <= 3; ++w) {
        run<24, 0, 0>(out, w); run<24, 0, 1>(out, w); run<0, 144, 0>(out, w);
        run<24, 72, 0>(out, w); run<24, 144, 0>(out, w); run<24, 144, 1>(out, w); run<24, 288, 0>(out, w);
    }
    return 0;
}
