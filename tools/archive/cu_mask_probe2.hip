// Dev probe (round 5): co-residency of one-block-per-CU kernels on a CU-masked stream: start time and place of every block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(512) void hog_kernel(unsigned long long ticks, unsigned long long* out)
{
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = ((unsigned long long)(xcc & 0xf) << 16) | (hw & 0xff00); }
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) acc += lds[threadIdx.x & 1023];
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv)
{
    const int ncu = 256;
    unsigned long long* dev;
    CK(hipMalloc(&dev, 1 << 20));
    CK(hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // masks: "drop k indices per XCC from the top" for k = 1, 2, 4 and "drop the bottom 2 / 4"
    struct M { const char* name; int lo, hi; } masks[] = {{"all 256", 0, 256}, {"bits [0,248)", 0, 248}, {"bits [0,240)", 0, 240}, {"bits [0,224)", 0, 224},
                                                          {"bits [16,256)", 16, 256}, {"bits [32,256)", 32, 256}};
    for (auto& mk : masks) {
        std::vector<uint32_t> m(8, 0u);
        for (int b = mk.lo; b < mk.hi; ++b) m[b / 32] |= 1u << (b % 32);
        hipStream_t st;
        CK(hipExtStreamCreateWithCUMask(&st, 8, m.data()));
        const int avail = mk.hi - mk.lo;
        for (int lds : {160 * 1024, 64 * 1024}) {
            for (int blocks : {avail, avail - 8, avail - 16}) {
                std::vector<unsigned long long> h(2 * blocks);
                hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(512), lds, st, 2000000ull, dev);      // 20 ms
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(h.data(), dev, blocks * 16, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull;
                for (int b = 0; b < blocks; ++b) t0 = std::min(t0, h[2 * b]);
                int late = 0; std::map<unsigned long long, int> per_cu; std::map<unsigned, int> late_se;
                for (int b = 0; b < blocks; ++b) {
                    if (h[2 * b] - t0 > 100000) { ++late; late_se[(unsigned)((h[2 * b + 1] >> 16) * 8 + ((h[2 * b + 1] >> 13) & 7))]++; }
                    per_cu[h[2 * b + 1]]++;
                }
                printf("%-14s lds %3d KiB  %3d blocks: %3zu distinct CUs, %3d blocks started late", mk.name, lds / 1024, blocks, per_cu.size(), late);
                if (late) { printf("  (xcc*8+se: count)"); for (auto& kv : late_se) printf(" %u:%d", kv.first, kv.second); }
                printf("\n");
            }
        }
        CK(hipStreamDestroy(st));
    }
    // SE layout of the low / high indices of XCC 0 (which SE the last mask indices of an XCC name)
    return 0;
}
