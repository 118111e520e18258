// Dev probe (round 4): LDS-DMA delivery rate per CU for the SMALL-M GEMM pattern: every block streams a 128-row A panel and a
// 128-row W panel ([rows][K/32][hi 32 | lo 32] = 128-byte lines per K-tile), pieces of 8 rows x 128 B per wave-instruction,
// ring of DEPTH pieces per wave in flight (counted vmcnt), no MFMA.  Sweeps: waves per block, depth, panel sharing (how many
// blocks read the same W panel / A panel: L2-hit vs fabric), clock.
// hipcc --offload-arch=gfx950 -O3 -o tools/dma_probe2.bin tools/dma_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ __forceinline__ void dma1(uint32_t voff, uint32_t m0v, const void* sbase)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(sbase) : "memory");
}
// block b: A panel (b % n_ap), W panel (b / n_ap) % n_wp.  Per K-tile the block moves 256 rows x 128 B = 32 KiB = 32 pieces.
template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void probe(const char* __restrict__ A, const char* __restrict__ W, int nk, int n_ap, int n_wp,
                                                    unsigned long long* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    int ap = blockIdx.x % n_ap, wp = (blockIdx.x / n_ap) % n_wp;
    if (n_ap < 0) {          // XCD-aware map of the small-M GEMM: an XCD (block & 7) owns whole W panels, all A panels of one back to back
        const int na = -n_ap, xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        ap = q % na; wp = ((q / na) * 8 + xcd) % n_wp;
    }
    const int64_t ld = (int64_t)nk * 128;
    const char* abase = A + (int64_t)ap * 128 * ld;
    const char* wbase = W + (int64_t)wp * 128 * ld;
    constexpr int PPW = 32 / WAVES;          // pieces per wave per K-tile
    uint32_t off[PPW];
    for (int j = 0; j < PPW; ++j) {
        const int piece = wid * PPW + j;     // 0..31: pieces 0..15 A rows 8p.., 16..31 W
        const int r = (piece & 15) * 8 + (lane >> 3);
        off[j] = (uint32_t)(r * ld + (lane & 7) * 16);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0));
    int issued = 0;
    for (int t = 0; t < nk; ++t) {
        for (int j = 0; j < PPW; ++j) {
            const int piece = wid * PPW + j;
            const char* base = (piece < 16 ? abase : wbase) + (int64_t)t * 128;
            dma1(off[j], lds0 + (uint32_t)(((t & 3) * 32 + piece) * 1024), base);
            ++issued;
            if (issued > DEPTH) {
                if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (DEPTH == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int WAVES, int DEPTH>
void run(const char* a, const char* w, int nk, int blocks, int n_ap, int n_wp, unsigned long long* out, const char* tag)
{
    const size_t lds = 128 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<WAVES, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int r = 0; r < 2; ++r) probe<WAVES, DEPTH><<<blocks, WAVES * 64, lds>>>(a, w, nk, n_ap, n_wp, out);
    hipDeviceSynchronize();
    hipEventRecord(s);
    const int R = 10;
    for (int r = 0; r < R; ++r) probe<WAVES, DEPTH><<<blocks, WAVES * 64, lds>>>(a, w, nk, n_ap, n_wp, out);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= R;
    unsigned long long h[2 * 1024]; hipMemcpy(h, out, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; } cyc /= blocks; rt /= blocks;
    const double bytes = (double)blocks * nk * 32768.0;
    printf("%-32s waves=%d depth=%2d blocks=%3d nk=%4d A-panels=%3d W-panels=%3d: launch %7.1f us  in-kernel %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU (clk %.2f GHz)  chip %.2f TB/s\n",
           tag, WAVES, DEPTH, blocks, nk, n_ap, n_wp, ms * 1e3, rt / 100.0, (double)nk * 32768.0 / (rt / 100.0) / 1e3,
           (double)nk * 32768.0 / cyc, cyc / (rt / 100.0) / 1e3, bytes / (rt / 100.0) / 1e6);
}
int main()
{
    const int nk = 32;                                  // K = 1024
    const size_t panel = (size_t)128 * nk * 128;        // 512 KiB
    char *a, *w; unsigned long long* out;
    hipMalloc(&a, panel * 512); hipMalloc(&w, panel * 512); hipMalloc(&out, 16 * 2048);
    hipMemset(a, 1, panel * 512); hipMemset(w, 2, panel * 512);
    // (1) the small-M GEMM: 8 A panels, 24 / 32 W panels (to_qkv / ff1), blocks = product
    run<8, 8>(a, w, nk, 192, 8, 24, out, "qkv-like");
    run<8, 16>(a, w, nk, 192, 8, 24, out, "qkv-like");
    run<8, 32>(a, w, nk, 192, 8, 24, out, "qkv-like");
    run<4, 16>(a, w, nk, 192, 8, 24, out, "qkv-like");
    run<4, 32>(a, w, nk, 192, 8, 24, out, "qkv-like");
    run<8, 16>(a, w, nk, 256, 8, 32, out, "ff1-like");
    run<8, 32>(a, w, nk, 256, 8, 32, out, "ff1-like");
    run<8, 16>(a, w, nk, 192, -8, 24, out, "qkv-like, XCD owns W panels");
    run<4, 16>(a, w, nk, 192, -8, 24, out, "qkv-like, XCD owns W panels");
    run<8, 16>(a, w, nk, 256, -8, 32, out, "ff1-like, XCD owns W panels");
    run<8, 16>(a, w, nk, 64, -8, 8, out, "out-like, XCD owns W panels");
    run<8, 16>(a, w, 8, 256, -8, 32, out, "splitK4 K=256 slices");
    // (2) everything L2-resident: one A panel, one W panel for all blocks
    run<8, 16>(a, w, nk, 256, 1, 1, out, "all-shared (L2 hits)");
    run<8, 32>(a, w, nk, 256, 1, 1, out, "all-shared (L2 hits)");
    run<4, 32>(a, w, nk, 256, 1, 1, out, "all-shared (L2 hits)");
    // (3) nothing shared: every block its own panels (HBM / MALL stream)
    run<8, 16>(a, w, nk, 256, 256, 256, out, "private panels");
    run<8, 32>(a, w, nk, 256, 256, 256, out, "private panels");
    // (4) few blocks: a lightly loaded chip (64 tiles: to_out)
    run<8, 16>(a, w, nk, 64, 8, 8, out, "out-like 64 blocks");
    run<8, 32>(a, w, nk, 64, 8, 8, out, "out-like 64 blocks");
    // (5) long K (ff2: nk = 128), 64 blocks and 256 blocks (split-K 4)
    run<8, 16>(a, w, 128, 64, 8, 8, out, "ff2-like");
    return 0;
}
