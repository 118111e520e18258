// Dev probe: issue cost (shader cycles per wave64 instruction) of the VALU / transcendental / MFMA instructions of the attention
// softmax, alone and mixed, at 1 and 2 waves per SIMD.  A wave runs ITERS x 16 independent instructions of a kind;
// cycles per instruction = s_memtime delta / (ITERS * 16) (per wave; at 2 waves per SIMD the SIMD serves two such streams).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_probe.bin tools/valu_rate_probe.hip && tools/valu_rate_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 512
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
// MODE: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_fma_mix_f32, 4 v_cvt_pk_f16_f32, 5 v_max3_f32, 6 mfma 32x32x16 (2 accumulators),
//       7 mfma + 4 v_fma per mfma (same wave), 8 waves 0-3 mfma / waves 4-7 v_fma (different waves of a SIMD), 9 same with v_exp
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int dummy)
{
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    f32x16 a0, a1;
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * threadIdx.x + i); fb[i] = (_Float16)(0.02f * threadIdx.x - i); }
    const float c = 1.0001f + dummy, d = 0.5f;
    const int wid = threadIdx.x >> 6;
    const bool mf = (MODE == 6 || MODE == 7) || ((MODE == 8 || MODE == 9) && wid < 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
            REP16(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(f32x2*)&v[2 * (i & 7)]) : "v"(f32x2{c, c}));
            REP16(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            REP16(X)
#undef X
        } else if (MODE == 3) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(dummy));
            REP16(X)
#undef X
        } else if (MODE == 4) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (MODE == 5) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
            REP16(X)
#undef X
        } else if (mf && MODE != 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a1, 0, 0, 0);
            }
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a0, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v[0]), "+v"(v[1]) : "v"(c), "v"(d));
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v[2]), "+v"(v[3]) : "v"(c), "v"(d));
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a1, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v[4]), "+v"(v[5]) : "v"(c), "v"(d));
                asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v[6]), "+v"(v[7]) : "v"(c), "v"(d));
            }
        } else if (MODE == 8) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            REP16(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + a0[i] + a1[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads, double per_iter, unsigned long long* d_out, float* d_sink)
{
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d_out, d_sink, 0);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double lo = 0, hi = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
    lo /= 256.0 * (nw < 4 ? nw : 4); if (nw > 4) hi /= 256.0 * (nw - 4);
    printf("%-44s %d waves/SIMD: %7.2f cycles per instruction (waves 0-3)", name, nw / 4, lo / (ITERS * per_iter));
    if (nw > 4) printf("   %7.2f (waves 4-7)", hi / (ITERS * (MODE == 8 ? 64.0 : per_iter)));
    printf("\n");
}
int main()
{
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_sink, 4);
    for (int t : {256, 512}) {
        run<0>("v_fma_f32", t, 16, d_out, d_sink);
        run<1>("v_pk_fma_f32", t, 16, d_out, d_sink);
        run<2>("v_exp_f32", t, 16, d_out, d_sink);
        run<3>("v_fma_mix_f32", t, 16, d_out, d_sink);
        run<4>("v_cvt_pk_f16_f32", t, 16, d_out, d_sink);
        run<5>("v_max3_f32", t, 16, d_out, d_sink);
        run<6>("v_mfma_f32_32x32x16_f16", t, 16, d_out, d_sink);
        run<7>("mfma + 4 v_fma each, one wave (per mfma)", t, 16, d_out, d_sink);
    }
    run<8>("waves 0-3 mfma | waves 4-7 v_fma", 512, 16, d_out, d_sink);
    run<9>("waves 0-3 mfma | waves 4-7 v_exp", 512, 16, d_out, d_sink);
    return 0;
}
