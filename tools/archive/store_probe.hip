// Dev probe: how fast can 256 blocks x 8 waves write [16000 x N] fp32 tiles of 256 x 256 with different per-instruction
// footprints (what a GEMM epilogue does)?  P0: 8 rows x 128 B per instruction (the quad-transposed MFMA layout),
// P1: 4 rows x 256 B, P2: 1 row x 1 KiB, each also with non-temporal stores; FILL: a flat streaming fill of the same bytes.
// hipcc --offload-arch=gfx950 -O3 -o tools/store_probe.bin tools/store_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int P, bool NT>
__global__ __launch_bounds__(512) void probe(float* __restrict__ C, int M, int N, int tiles_n, int tiles_total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        const int xcd = tile & 7, slot = tile >> 3;
        const int tm = xcd + 8 * (slot / tiles_n), tn = slot % tiles_n;
        const int m0 = tm * 256, n0 = tn * 256;
        if (m0 >= M) continue;
        const int wr = wid >> 2, wc = wid & 3;
        f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
        if (P == 0) {
            // wave tile 128 x 64: per (mi, rg, half): 8 rows x 128 B
            for (int mi = 0; mi < 4; ++mi)
                for (int rg = 0; rg < 4; ++rg)
                    for (int hf = 0; hf < 2; ++hf) {
                        const int row = m0 + wr * 128 + mi * 32 + 8 * rg + 4 * (lane >> 5) + (lane & 3);
                        const int col = n0 + wc * 64 + hf * 32 + 4 * ((lane & 31) >> 2);
                        if (row < M) {
                            f32x4* p = reinterpret_cast<f32x4*>(C + (int64_t)row * N + col);
                            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                        }
                    }
        } else if (P == 1) {
            // wave tile 128 x 64: per instruction 4 rows x 256 B
            for (int i = 0; i < 32; ++i) {
                const int row = m0 + wr * 128 + 4 * i + (lane >> 4);
                const int col = n0 + wc * 64 + 4 * (lane & 15);
                if (row < M) {
                    f32x4* p = reinterpret_cast<f32x4*>(C + (int64_t)row * N + col);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
            }
        } else {
            // wave owns 32 full rows of the 256-wide tile: 1 row x 1 KiB per instruction
            for (int i = 0; i < 32; ++i) {
                const int row = m0 + wid * 32 + i;
                const int col = n0 + 4 * lane;
                if (row < M) {
                    f32x4* p = reinterpret_cast<f32x4*>(C + (int64_t)row * N + col);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
            }
        }
    }
}
__global__ void fill(f32x4* p, int64_t n4) { for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) p[i] = f32x4{1, 2, 3, 4}; }
template <int P, bool NT> void run(float* C, int M, int N, int grid, const char* tag)
{
    const int tn = N / 256, tm = (M + 255) / 256, gm = (tm + 7) / 8 * 8, total = gm * tn;
    if (grid == 0) grid = total;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    probe<P, NT><<<grid, 512>>>(C, M, N, tn, total); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 10; ++r) probe<P, NT><<<grid, 512>>>(C, M, N, tn, total);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
    printf("%-34s N=%4d grid=%4d: %7.1f us  %.2f TB/s\n", tag, N, grid, ms * 1e3, (double)M * N * 4 / ms / 1e9);
}
int main()
{
    const int M = 16000;
    float* C; hipMalloc(&C, (size_t)16384 * 4096 * 4);
    for (int N : {1024, 4096}) {
        run<0, false>(C, M, N, 0, "8 rows x 128 B"); run<0, true>(C, M, N, 0, "8 rows x 128 B nt");
        run<1, false>(C, M, N, 0, "4 rows x 256 B"); run<1, true>(C, M, N, 0, "4 rows x 256 B nt");
        run<2, false>(C, M, N, 0, "1 row x 1 KiB"); run<2, true>(C, M, N, 0, "1 row x 1 KiB nt");
        run<0, false>(C, M, N, 256, "8 rows x 128 B persistent(256)"); run<2, false>(C, M, N, 256, "1 row x 1 KiB persistent(256)");
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        const int64_t n4 = (int64_t)M * N / 4;
        fill<<<2048, 256>>>((f32x4*)C, n4); hipDeviceSynchronize();
        hipEventRecord(s); for (int r = 0; r < 10; ++r) fill<<<2048, 256>>>((f32x4*)C, n4); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
        printf("%-34s N=%4d           : %7.1f us  %.2f TB/s\n", "flat fill", N, ms * 1e3, (double)M * N * 4 / ms / 1e9);
    }
    return 0;
}
