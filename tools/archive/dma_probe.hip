// Dev probe: LDS-DMA (global_load_lds_dwordx4) streaming rate per CU vs tiles in flight, GEMM-A access pattern.
// hipcc --offload-arch=gfx950 -O3 -o tools/dma_probe.bin tools/dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16;
__device__ __forceinline__ void glds16(const void* g, void* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// block = 512 threads, owns 256 rows of A_hi and A_lo [M][K] (f16); per step a 256 x 32 tile of each (32 KiB total)
template <int S>
__global__ __launch_bounds__(512) void probe(const f16* __restrict__ Ah, const f16* __restrict__ Al, int K, float* out, int mod)
{
    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    constexpr int STAGE = 2 * 256 * 32;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t m0 = (int64_t)(blockIdx.x % mod) * 256;
    const f16* ph[2]; const f16* pl[2];
    for (int j = 0; j < 2; ++j) {
        const int r = 32 * wid + 16 * j + (lane >> 2);
        ph[j] = Ah + (m0 + r) * K + 8 * (lane & 3);
        pl[j] = Al + (m0 + r) * K + 8 * (lane & 3);
    }
    const int nk = K / 32;
    auto issue = [&](int t) {
        f16* Sx = smem + (t % S) * STAGE + 32 * wid * 32;
        for (int j = 0; j < 2; ++j) {
            glds16(ph[j], Sx + 16 * j * 32);
            glds16(pl[j], Sx + 256 * 32 + 16 * j * 32);
            ph[j] += 32; pl[j] += 32;
        }
    };
    for (int t = 0; t < S - 1; ++t) issue(t);
    float acc = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
        if (S == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (S == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (S == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nk) issue(kt + S - 1);
        else { asm volatile("" ::: "memory"); }
        acc += (float)smem[(kt % S) * STAGE + threadIdx.x];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int S> void run(const f16* a, const f16* b, int M, int K, float* out, int mod = 1 << 30)
{
    const size_t lds = (size_t)S * 2 * 256 * 32 * sizeof(f16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    probe<S><<<M / 256, 512, lds>>>(a, b, K, out, mod);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) probe<S><<<M / 256, 512, lds>>>(a, b, K, out, mod);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 5;
    const double bytes = 2.0 * M * K * 2;
    printf("mod=%d stages=%d (in flight %d x 32 KiB) M=%d K=%d: %.3f ms  %.2f TB/s  %.1f GB/s/CU  %.1f B/clk/CU@2.1GHz\n", mod, S, S - 1, M, K, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
}
int main()
{
    const int M = 65536, K = 4096;               // 256 blocks... M/256 = 256 blocks = 1 per CU; 1 GiB total
    f16 *a, *b; float* out;
    hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&b, (size_t)M * K * 2); hipMalloc(&out, 4 * 512 * 4096);
    hipMemset(a, 0, (size_t)M * K * 2); hipMemset(b, 0, (size_t)M * K * 2);
    run<2>(a, b, M, K, out); run<3>(a, b, M, K, out); run<4>(a, b, M, K, out); run<5>(a, b, M, K, out);
    const int M2 = 16384;                        // 64 blocks only: lightly loaded chip
    run<2>(a, b, M2, K, out); run<4>(a, b, M2, K, out);
    run<2>(a, b, M, K, out, 1); run<4>(a, b, M, K, out, 1); run<2>(a, b, M, K, out, 8); run<4>(a, b, M, K, out, 8);
    return 0;
}
