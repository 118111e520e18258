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
// compute units of the current device (cached per device)
int cvx_device_cus();

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

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }
