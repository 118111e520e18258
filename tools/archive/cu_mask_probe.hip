// Dev probe (round 5): CU-masked streams on MI355X.  hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe.bin
//  1. which physical CUs (XCC, SE, CU) a stream created with hipExtStreamCreateWithCUMask runs on, for a few masks;
//  2. do two masked streams with disjoint masks run concurrently without slowing each other down (a long persistent "solve"
//     kernel on 240 CUs, a chain of short dependent kernels on 16);
//  3. does a HIP graph captured from / launched into a masked stream keep the mask.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>
#include <map>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void where_kernel(unsigned* out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xff00);
    // hold the CU for a while so that the blocks spread over everything the stream may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
}

// "solve": every block spins for `cycles` (a persistent kernel that owns its CU: 160 KiB of LDS requested -> one block per CU)
__global__ void hog_kernel(unsigned long long cycles, unsigned* sink)
{
    extern __shared__ char lds[];
    unsigned long long t0 = wall_clock64();                       // 100 MHz
    unsigned acc = 0;
    while (wall_clock64() - t0 < cycles) acc += lds[threadIdx.x & 1023];
    if (acc == 0x12345678u) sink[0] = acc;
}

// one link of a latency chain: 64 blocks stream 4 MB
__global__ void link_kernel(const float4* __restrict__ src, float* __restrict__ dst, int n4)
{
    float s = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) { float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 1.2345f) dst[0] = s;
}

static int census(hipStream_t st, const char* name, unsigned* dev, int blocks)
{
    std::vector<unsigned> h(blocks);
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), 0, st, dev, 200000);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), dev, blocks * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per;
    for (unsigned v : h) per[v >> 16].insert(v & 0xffff);
    int tot = 0;
    printf("%-34s", name);
    for (auto& kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total %d CUs\n", tot);
    return 0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int ncu = prop.multiProcessorCount;
    unsigned* dev;
    CK(hipMalloc(&dev, 1 << 20));
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    census(plain, "plain stream", dev, 4096);

    auto mk = [&](std::vector<int> bits, hipStream_t* st) -> hipError_t {
        std::vector<uint32_t> m((ncu + 31) / 32, 0u);
        for (int b : bits) m[b / 32] |= 1u << (b % 32);
        return hipExtStreamCreateWithCUMask(st, (uint32_t)m.size(), m.data());
    };
    std::vector<int> lo240, hi16, first32, evens, per_xcd2, per_xcd30;
    for (int i = 0; i < ncu; ++i) {
        if (i < ncu - 16) lo240.push_back(i); else hi16.push_back(i);
        if (i < 32) first32.push_back(i);
        if (i % 2 == 0) evens.push_back(i);
    }
    hipStream_t sA, sB, sC, sD;
    hipError_t e = mk(lo240, &sA);
    printf("hipExtStreamCreateWithCUMask(low %d bits): %s\n", ncu - 16, hipGetErrorString(e));
    if (e != hipSuccess) return 2;
    CK(mk(hi16, &sB)); CK(mk(first32, &sC)); CK(mk(evens, &sD));
    census(sA, "mask bits [0, ncu-16)", dev, 4096);
    census(sB, "mask bits [ncu-16, ncu)", dev, 4096);
    census(sC, "mask bits [0, 32)", dev, 4096);
    census(sD, "mask even bits", dev, 4096);
    std::vector<uint32_t> got(8, 0u);
    if (hipExtStreamGetCUMask(sB, 8, got.data()) == hipSuccess) {
        printf("hipExtStreamGetCUMask(sB):");
        for (uint32_t w : got) printf(" %08x", w);
        printf("\n");
    }

    // ---- 2. concurrency
    float4* src; float* dst; unsigned* sink;
    const int n4 = 1 << 18;                                   // 4 MB
    CK(hipMalloc(&src, (size_t)n4 * 16)); CK(hipMalloc(&dst, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 0, (size_t)n4 * 16));
    CK(hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        auto chain = [&](hipStream_t st, int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(link_kernel, dim3(64), dim3(256), 0, st, src, dst, n4); };
    auto timed_chain = [&](hipStream_t st, int n, const char* what) {
        hipStreamSynchronize(st);
        double t0 = now(); chain(st, n); hipStreamSynchronize(st); double t1 = now();
        printf("  %-58s %8.1f us per link\n", what, (t1 - t0) / n * 1e6);
    };
    for (int rep = 0; rep < 2; ++rep) { chain(plain, 200); chain(sB, 200); }
    CK(hipDeviceSynchronize());
    timed_chain(plain, 2000, "chain alone, plain stream");
    timed_chain(sB, 2000, "chain alone, 16-CU stream");
    // measure the hog alone
    {
        double t0 = now();
        hipLaunchKernelGGL(hog_kernel, dim3(ncu - 16), dim3(512), 160 * 1024, sA, 20ull * 1000 * 1000, sink);
        CK(hipStreamSynchronize(sA));
        printf("  hog alone on the 240-CU stream: %.1f ms\n", (now() - t0) * 1e3);
    }
    // hog on A (masked) + chain on B (masked)
    {
        double t0 = now();
        hipLaunchKernelGGL(hog_kernel, dim3(ncu - 16), dim3(512), 160 * 1024, sA, 20ull * 1000 * 1000, sink);
        double c0 = now(); chain(sB, 2000); CK(hipStreamSynchronize(sB)); double c1 = now();
        CK(hipStreamSynchronize(sA));
        printf("  hog(240-CU stream) + chain(16-CU stream): chain %.1f us per link, hog done after %.1f ms\n", (c1 - c0) / 2000 * 1e6, (now() - t0) * 1e3);
    }
    // hog on plain with a full-chip grid + chain on a plain stream: the chain has to wait
    {
        hipStream_t plain2; CK(hipStreamCreate(&plain2));
        double t0 = now();
        hipLaunchKernelGGL(hog_kernel, dim3(ncu), dim3(512), 160 * 1024, plain, 20ull * 1000 * 1000, sink);
        double c0 = now(); chain(plain2, 2000); CK(hipStreamSynchronize(plain2)); double c1 = now();
        CK(hipStreamSynchronize(plain));
        printf("  hog(plain, all CUs) + chain(plain): chain %.1f us per link, hog done after %.1f ms\n", (c1 - c0) / 2000 * 1e6, (now() - t0) * 1e3);
    }
    // hog with 240 blocks on a plain stream + chain on the 16-CU stream
    {
        double t0 = now();
        hipLaunchKernelGGL(hog_kernel, dim3(ncu - 16), dim3(512), 160 * 1024, plain, 20ull * 1000 * 1000, sink);
        double c0 = now(); chain(sB, 2000); CK(hipStreamSynchronize(sB)); double c1 = now();
        CK(hipStreamSynchronize(plain));
        printf("  hog(plain, 240 blocks) + chain(16-CU stream): chain %.1f us per link, hog done after %.1f ms\n", (c1 - c0) / 2000 * 1e6, (now() - t0) * 1e3);
    }

    // ---- 3. graphs
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(sB, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, sB, dev, 200000);
        CK(hipStreamEndCapture(sB, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<unsigned> h(4096);
        for (int which = 0; which < 2; ++which) {
            hipStream_t st = which ? plain : sB;
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), dev, 4096 * 4, hipMemcpyDeviceToHost));
            std::set<unsigned> cus; for (unsigned v : h) cus.insert(v);
            printf("graph captured on the 16-CU stream, launched into %s: %zu CUs\n", which ? "a plain stream" : "the 16-CU stream", cus.size());
        }
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(plain, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, plain, dev, 200000);
        CK(hipStreamEndCapture(plain, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge2, sB)); CK(hipStreamSynchronize(sB));
        CK(hipMemcpy(h.data(), dev, 4096 * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus; for (unsigned v : h) cus.insert(v);
        printf("graph captured on a plain stream, launched into the 16-CU stream: %zu CUs\n", cus.size());
    }
    printf("done\n");
    return 0;
}
