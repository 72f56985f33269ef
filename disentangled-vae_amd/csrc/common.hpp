// Shared helpers for the gfx950 kernels of libdvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/dvae.h"

namespace dvae {

void set_error(const char* fmt, ...);

#define DVAE_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            dvae::set_error(__VA_ARGS__);         \
            return DVAE_E_BADARG;                 \
        }                                         \
    } while (0)

#define DVAE_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            dvae::set_error("%s failed: %s", #expr, hipGetErrorString(e_));        \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

// Launch check: hipGetLastError after a <<<>>> launch (no sync).
#define DVAE_LAUNCH_OK(name)                                                        \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            dvae::set_error("launch of %s failed: %s", name, hipGetErrorString(e_));\
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;

// -DDVAE_DIAG (python disentangled-vae_amd/build.py --diag -> libdvae_hip_diag.so): the default library holds PRODUCT kernels only; the
// measured-slower alternates that earlier rounds built and kept for A/B runs and equality tests -- the 4-wave rows kernel under the bf16
// policies, the LDS-staged weight-gradient kernel, the optimizer step in the weight-gradient kernel's tail or in the next rows kernel's
// opening, the unit-mapped apply kernel, the staged ISTFT on frame-major input -- exist in the diagnostic build alone.
#ifdef DVAE_DIAG
constexpr bool kDiagBuild = true;
#else
constexpr bool kDiagBuild = false;
#endif

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case DVAE_ACT_TANH: return tanhf(v);
        case DVAE_ACT_RELU: return v > 0.f ? v : 0.f;
        case DVAE_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case DVAE_ACT_EXP: return expf(v);
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT o
__device__ __forceinline__ float act_grad_from_out(float o, int act) {
    switch (act) {
        case DVAE_ACT_TANH: return 1.f - o * o;
        case DVAE_ACT_RELU: return o > 0.f ? 1.f : 0.f;
        case DVAE_ACT_SIGMOID: return o * (1.f - o);
        case DVAE_ACT_EXP: return o;
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace dvae
