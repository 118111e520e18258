// Dev probe: what does each ingredient of the fp32 GEMM main loop cost on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_probe.hip ; run: /tmp/mfma_probe
// MODE bits: 1 = ds_read_b128 fragments from LDS, 2 = barrier per 64 MFMAs, 4 = global loads + ds_write per 64 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 36;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ g, float* __restrict__ out, int iters, size_t gstride)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2 * 256 * LD; i += 256) smem[i] = 0.001f * (i % 97);
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int a_off = ((wid >> 1) * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
    const int b_off = (128 + (wid & 1) * 64 + (lane & 31)) * LD + (lane >> 5) * 4;
    const float* gp = g + (size_t)blockIdx.x * gstride + (tid >> 3) * 1024 + (tid & 7) * 4;
    f32x4 regs[8];
    f32x4 af0 = {1.f, 2.f, 3.f, 4.f}, af1 = af0, bf0 = af0, bf1 = af0;
    for (int it = 0; it < iters; ++it) {
        const float* A = smem + (it & 1) * 256 * LD;
        if ((MODE & 4) && !(MODE & 32)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) regs[i] = *reinterpret_cast<const f32x4*>(gp + i * 32 * 1024);
            gp += 32;
        }
        if (MODE & 32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { regs[i] = af0; asm volatile("" : "+v"(regs[i])); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE & 1) {
                af0 = *reinterpret_cast<const f32x4*>(A + a_off + q * 8);
                af1 = *reinterpret_cast<const f32x4*>(A + a_off + 32 * LD + q * 8);
                bf0 = *reinterpret_cast<const f32x4*>(A + b_off + q * 8);
                bf1 = *reinterpret_cast<const f32x4*>(A + b_off + 32 * LD + q * 8);
            }
            if ((MODE & 4) && q == 3) {
                if (MODE & 16) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(regs[i]));
                } else {
                    float* W = smem + ((it & 1) ^ 1) * 256 * LD + (tid >> 3) * LD + (tid & 7) * 4;
#pragma unroll
                    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(W + i * 32 * LD) = regs[i];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[t], bf0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[t], bf1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[t], bf0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[t], bf1[t], acc[1][1], 0, 0, 0);
            }
            if (!(MODE & 1)) { asm volatile("" : "+v"(af0), "+v"(bf0)); }
        }
        if (MODE & 8) {
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
            SGB(0x100, 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1); SGB(0x020, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { SGB(0x008, 2); SGB(0x100, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { SGB(0x008, 4); SGB(0x100, 1); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1); SGB(0x200, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { SGB(0x008, 2); SGB(0x100, 1); }
            SGB(0x008, 16);
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, const float* g, float* out, int blocks, int iters, size_t gstride)
{
    const size_t lds = 2 * 256 * LD * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    probe<MODE><<<blocks, 256, lds>>>(g, out, iters, gstride);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) probe<MODE><<<blocks, 256, lds>>>(g, out, iters, gstride);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    ms /= 5;
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-44s blocks=%5d iters=%4d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks, iters, ms, flops / ms / 1e9);
}

int main()
{
    const int iters = 128;                       // = K 4096 worth of k-steps
    const size_t gstride = 1024 * 256 + 64;      // floats per block region (>= 256 rows * 1024)
    const int maxb = 4096;
    float *g, *out;
    hipMalloc(&g, sizeof(float) * (gstride * maxb + 32 * iters + 1024 * 256));
    hipMemset(g, 0, sizeof(float) * (gstride * maxb + 32 * iters + 1024 * 256));
    hipMalloc(&out, sizeof(float) * maxb * 256);
    for (int blocks : {256, 512, 1024, 4096}) {
        run<0>("pure MFMA", g, out, blocks, iters, gstride);
        run<1>("+ ds_read_b128 fragments", g, out, blocks, iters, gstride);
        run<3>("+ ds_read + barrier", g, out, blocks, iters, gstride);
        run<7>("+ds_read+barrier+global(HBM stream)/ds_write", g, out, blocks, iters, gstride);
        run<7>("+ds_read+barrier+global(L2 shared)/ds_write", g, out, blocks, iters, 0);
        run<6>("barrier+global(L2 shared)/ds_write, no ds_read", g, out, blocks, iters, 0);
        run<15>("mode 7 (L2 shared) + sched_group_barrier interleave", g, out, blocks, iters, 0);
        run<23>("mode 7 (L2 shared) loads only, no ds_write", g, out, blocks, iters, 0);
        run<39>("mode 7 ds_write only, no global loads", g, out, blocks, iters, 0);
        run<15>("mode 7 (HBM stream) + sched_group_barrier interleave", g, out, blocks, iters, gstride);
    }
    return 0;
}
