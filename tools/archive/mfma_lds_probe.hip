// Dev probe: do ds_read_b128 fragment reads overlap with fp16 MFMAs on gfx950?  Per round a wave issues 24
// v_mfma_f32_32x32x16_f16 (8 accumulators x 3, the GEMM's mix) and R ds_read_b128 whose results feed the NEXT round's
// MFMAs (so the reads are real operands but never on the critical path).  8 waves per CU (2 per SIMD), 256 blocks.
// hipcc --offload-arch=gfx950 -O3 -o x.bin tools/mfma_lds_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int DO_MFMA>
__global__ __launch_bounds__(512) void probe(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768; i += 512) lds[i] = (_Float16)(((i * 2654435761u) >> 20 & 4095) * (1.0f / 4096.0f) - 0.5f);
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 f[12], g[12];
    for (int i = 0; i < 12; ++i) { f[i] = *reinterpret_cast<f16x8*>(lds + ((lane * 8 + i * 512) & 32767)); g[i] = f[i]; }
    int off = lane * 8;
    for (int it = 0; it < iters; ++it) {
        // reads for the next round (conflict-free: consecutive lanes read consecutive 16-byte chunks)
#pragma unroll
        for (int i = 0; i < R; ++i) g[i % 12] = *reinterpret_cast<f16x8*>(lds + ((off + i * 512) & 32767));
        if (DO_MFMA) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[(a & 3) + 4 * (t & 1)], f[8 + (a >> 2) + 2 * (t >> 1)], acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) f[i] = g[i];
        off = (off + 4096) & 32767;
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 12; ++i) s += (float)f[i][0];
    out[blockIdx.x * 512 + tid] = s;
}

template <int R, int DO_MFMA> void run(float* out)
{
    const int iters = 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<R, DO_MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int r = 0; r < 3; ++r) probe<R, DO_MFMA><<<256, 512, 65536>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 10; ++r) probe<R, DO_MFMA><<<256, 512, 65536>>>(out, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
    printf("reads/round=%2d mfma=%d: %.3f ms  -> %.2f us per round (24 MFMAs/wave, 2 waves/SIMD)\n", R, DO_MFMA, ms, ms * 1e3 / iters);
}
int main()
{
    float* out; hipMalloc(&out, 4 * 512 * 256);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1>(out); run<6, 1>(out); run<12, 1>(out); run<24, 1>(out);
        run<12, 0>(out); run<24, 0>(out);
    }
    return 0;
}
