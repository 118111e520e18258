// Shared helpers for the covomix HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/covomix_hip.h"

typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void cvx_set_error(const char* fmt, ...);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) per (device, kernel), raised whenever a launch asks for more than was
// granted so far: function attributes are per device, so a once-per-process flag is wrong for the second GPU a process
// uses.  Thread-safe; a map lookup after the first call.
void cvx_allow_dynamic_lds(const void* kernel, int bytes);
// compute units of the current device (cached per device): what a launch context with n_cus == 0 stands for
int cvx_device_cus();
// The launch context every entry point takes (covomix_hip.h: cvx_ctx / cvx_stream_t): the stream, the CALLER-OWNED sticky saturation
// flag of that stream and the CUs the stream owns.  Nothing about a stream lives in the library (until version 106 a process-wide
// (device, stream) table behind a mutex did: a hipGetDevice + lock + scan per launch, and a re-issued stream handle inherited it).
static inline hipStream_t cvx_hip_stream(cvx_stream_t s) { return s ? reinterpret_cast<hipStream_t>(s->stream) : nullptr; }
static inline int cvx_ctx_cus(cvx_stream_t s) { return (s && s->n_cus > 0) ? s->n_cus : cvx_device_cus(); }
// Every kernel that writes (fp16 hi, fp16 lo) split pairs clamps to +-65504 and ORs bit 0 into the context's flag when a value it
// stored was larger than that - the pair then no longer represents the fp32 value and the caller must not trust the result.  A call
// that writes pairs REFUSES a context without a flag (CVX_REQUIRE_SAT) unless the caller waived the bookkeeping explicitly.
static inline uint32_t* cvx_sat_flag_for(cvx_stream_t s) { return s ? s->sat_flag : nullptr; }
#define CVX_REQUIRE_SAT(s)                                                                                                            \
    CVX_REQUIRE((s) && ((s)->sat_flag || ((s)->flags & CVX_CTX_NO_SATURATION_FLAG)),                                                 \
                "%s: this call writes split (fp16 hi, fp16 lo) pairs and its launch context carries no saturation flag: set "      \
                "cvx_ctx.sat_flag (caller-owned device word), or CVX_CTX_NO_SATURATION_FLAG to run without the bookkeeping", __func__)

#define CVX_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) {                               \
            cvx_set_error(__VA_ARGS__);              \
            return CVX_EINVAL;                       \
        }                                            \
    } while (0)

#define CVX_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            cvx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return CVX_EHIP;                                                     \
        }                                                                        \
    } while (0)

// MFMA accumulator element r of a 32x32 tile lives at (row, col) =
// ((r&3) + 8*(r>>2) + 4*(lane>>5), lane&31)   [cdna_hip_programming.md section 3]
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// 16-byte load through the GLOBAL address space.  Pointers that went through select/offset arithmetic lose
// their provenance and hipcc falls back to flat_load (counted on lgkmcnt too, so every LDS wait would also
// wait for HBM); this keeps them global_load_dwordx4.
typedef const f32x4 __attribute__((address_space(1)))* cvx_gptr4;
__device__ __forceinline__ f32x4 gload4(const float* p) { return *reinterpret_cast<cvx_gptr4>(reinterpret_cast<uintptr_t>(p)); }

// valid positions of item b in a ragged vocoder batch (cvx_item_lengths in the header), at most the common length L
__device__ __forceinline__ int cvx_item_len(const cvx_item_lengths& it, int b, int L)
{
    return it.item_len_dev ? min(L, max(0, it.item_len_dev[b] * it.mul + it.add)) : L;
}

// Saturation bookkeeping of the values a lane stores as split pairs: the running max |v| and a NaN bit.  The clamps
// (v_med3 / fmin(fmax)) turn a NaN into -65504 and v_max3 skips NaN operands, so NaN needs its own predicate: one v_cmp_u_f32
// per two values into a scalar mask (no vector register).  -DCVX_NO_NAN_TRACK: dev A/B of what that costs.
struct CvxSat {
    float m; bool bad;
    __device__ __forceinline__ CvxSat() : m(0.f), bad(false) {}
};
// max(m, |a|, |b|) in ONE instruction (fmaxf would add a canonicalising v_max per operand in IEEE mode and, in the big unrolled
// GEMM epilogues, enough live values to spill)
__device__ __forceinline__ void cvx_amax3(CvxSat& s, const float a, const float b)
{
#ifndef CVX_NO_SAT_TRACK                      // (dev A/B: what the bookkeeping costs)
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(s.m) : "v"(a), "v"(b));
#ifndef CVX_NO_NAN_TRACK
    s.bad |= __builtin_isunordered(a, b);
#endif
#endif
}
__device__ __forceinline__ void cvx_amax4(CvxSat& s, const f32x4 v)
{
    cvx_amax3(s, v[0], v[1]);
    cvx_amax3(s, v[2], v[3]);
}
// the same in plain C on one float for the 32 x 32 GEMM epilogues of gemm_common.h: there the inline asm (opaque to the
// optimiser) made the compiler keep a 576-byte copy of the accumulator block in scratch in every kernel that carries the generic
// epilogue (round 3: the opt-in f16 mode lost 22 % to it before this was found).  A value that is not <= 65504 in magnitude
// (too large, infinite or NaN) makes the running maximum infinite.
#ifndef CVX_NO_SAT_TRACK
__device__ __forceinline__ float cvx_amax3_c(float m, const float a, const float b)
{
    return (fabsf(a) <= 65504.f && fabsf(b) <= 65504.f) ? m : __builtin_inff();
}
#else
__device__ __forceinline__ float cvx_amax3_c(float m, const float, const float) { return m; }
#endif
// the commit: one atomic, only when saturated (a NaN maximum counts: "not <=", so a direct cvx_sat_commit(flag, |v|) is covered too)
__device__ __forceinline__ void cvx_sat_commit(uint32_t* flag, float amax)
{
    if (flag && !(amax <= 65504.f)) atomicOr(flag, 1u);
}
__device__ __forceinline__ void cvx_sat_commit(uint32_t* flag, const CvxSat& s)
{
    if (flag && (s.m > 65504.f || s.bad)) atomicOr(flag, 1u);
}

// the same, NON-TEMPORAL (global_load_dwordx4 ... nt): for weight rows that ONE CU reads once per pass (batch-1 decode GEMVs) -
// MI355X_MICROARCH.md `nt-weights`: issued -> landed -18 %, a decode layer -5...10 %
__device__ __forceinline__ f32x4 gload4_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<cvx_gptr4>(reinterpret_cast<uintptr_t>(p))); }

// packed-pair fp32 helpers (v_pk_fma_f32 / v_pk_mul_f32)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(const f32x2 a, const f32x2 b, const f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(const float x) { return f32x2{x, x}; }
// erf to ~1 ulp without branches (both polynomial pieces, one select): |x| <= 0.927734375: x + x * P(x^2); beyond:
// 1 - exp(Q(|x|)).  The library erff is several times longer and branchy; GELU runs on 4096 columns of every row.
__device__ __forceinline__ f32x2 erf_fast2(const f32x2 a)
{
    const f32x2 t = __builtin_elementwise_abs(a), s = a * a;
    f32x2 r = fma2(splat2(-1.72853470e-5f), t, splat2(3.83197126e-4f));
    const f32x2 u = fma2(splat2(-3.88396438e-3f), t, splat2(2.42546219e-2f));
    r = fma2(r, s, u);
    r = fma2(r, t, splat2(-1.06777877e-1f));
    r = fma2(r, t, splat2(-6.34846687e-1f));
    r = fma2(r, t, splat2(-1.28717512e-1f));
    r = fma2(r, t, -t);
    const f32x2 e = r * splat2(1.44269504088896340736f);
    r = splat2(1.0f) - f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    r = f32x2{copysignf(r[0], a[0]), copysignf(r[1], a[1])};
    f32x2 q = splat2(-5.96761703e-4f);
    q = fma2(q, s, splat2(4.99119423e-3f));
    q = fma2(q, s, splat2(-2.67681349e-2f));
    q = fma2(q, s, splat2(1.12819925e-1f));
    q = fma2(q, s, splat2(-3.76125336e-1f));
    q = fma2(q, s, splat2(1.28379166e-1f));
    q = fma2(q, a, a);
    return f32x2{t[0] > 0.927734375f ? r[0] : q[0], t[1] > 0.927734375f ? r[1] : q[1]};
}
__device__ __forceinline__ f32x2 gelu_fast2(const f32x2 v)
{
    return (splat2(0.5f) * v) * (splat2(1.0f) + erf_fast2(v * splat2(0.70710678118654752440f)));
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }
