// Label makers of the training-set builders on the GPU (reference packages/processing/target.py:5-70):
// time-domain VAD (frame energy against the quietest frame) and the ideal binary mask (bins within
// `ibm_threshold` dB of the loudest bin).  Outputs are 0/1 labels and must match the reference bit for bit,
// so every float32 operation of the numpy code is reproduced as "exact value rounded once to float32"
// (double arithmetic, then one rounding): that is what a correctly rounded float32 libm returns.
#include <math.h>
#include "common.hpp"

namespace dvae {

// energy[t] = sum_{i < nfft} y[t*hop + i]^2 in double (samples past n count as the zero end-pad)
template <typename T>
__global__ __launch_bounds__(256) void frame_energy_kernel(const T* __restrict__ y, int64_t n, int nfft, int hop, int64_t frames,
                                                           double* __restrict__ energy) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= frames) return;
    const int64_t s0 = t * hop;
    double a = 0.0;
    for (int i = lane; i < nfft; i += 64) {
        const int64_t s = s0 + i;
        const double v = s < n ? (double)y[s] : 0.0;
        a = fma(v, v, a);
    }
    a = wave_sum(a);
    if (lane == 0) energy[t] = a;
}

// one workgroup: min over frames, then vad[t] = energy[t] > factor * min   (target.py:52-54)
__global__ __launch_bounds__(1024) void vad_threshold_kernel(const double* __restrict__ energy, int64_t frames, double factor, float* __restrict__ vad) {
    __shared__ double red[16];
    __shared__ double mn_s;
    double mn = INFINITY;
    for (int64_t t = threadIdx.x; t < frames; t += 1024) mn = fmin(mn, energy[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = fmin(mn, __shfl_xor(mn, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = red[0];
        for (int w = 1; w < 16; ++w) m = fmin(m, red[w]);
        mn_s = m;
    }
    __syncthreads();
    const double thr = factor * mn_s;
    for (int64_t t = threadIdx.x; t < frames; t += 1024) vad[t] = energy[t] > thr ? 1.f : 0.f;
}

// float32 |S| as numpy computes it (npy_cabsf = hypotf, correctly rounded in current glibc)
__device__ __forceinline__ float mag_f32(float re, float im) {
    return (float)sqrt((double)re * (double)re + (double)im * (double)im);
}
// float32 20 * log10(mag + eps): float32 add, correctly rounded float32 log10, float32 multiply
__device__ __forceinline__ float db_f32(float mag, float eps) {
    const float t = mag + eps;
    return 20.f * (float)log10((double)t);
}

__global__ __launch_bounds__(256) void ibm_max_kernel(const float2* __restrict__ S, int64_t count, float* __restrict__ partial) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float2 v = S[i];
        m = fmaxf(m, mag_f32(v.x, v.y));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// mask = 20 log10(|S| + eps) > max_db - threshold  (target.py:65-68); log10 is monotone, so max_db comes from the max magnitude
__global__ __launch_bounds__(256) void ibm_mask_kernel(const float2* __restrict__ S, int64_t count, const float* __restrict__ partial, int nparts,
                                                       float eps, float threshold, const float* __restrict__ gate, int64_t gate_cols,
                                                       float* __restrict__ mask) {
    float m = 0.f;
    for (int i = 0; i < nparts; ++i) m = fmaxf(m, partial[i]);       // nparts <= 1024, cached
    const float thr = db_f32(m, eps) - threshold;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float2 v = S[i];
        float o = db_f32(mag_f32(v.x, v.y), eps) > thr ? 1.f : 0.f;
        if (gate) o *= gate[i % gate_cols];                            // noise_robust_clean_speech_IBM: ibm * vad (target.py:103)
        mask[i] = o;
    }
}

}  // namespace dvae

using namespace dvae;

extern "C" size_t dvae_vad_workspace_bytes(int64_t frames) { return (size_t)frames * sizeof(double) + 256; }

extern "C" int dvae_vad_labels(const void* y, int in_f64, int64_t n, int nfft, int hop, int64_t frames, double vad_threshold,
                               float* vad, void* workspace, void* stream) {
    DVAE_CHECK_ARG(y && vad && workspace, "vad_labels: null argument");
    DVAE_CHECK_ARG(n > 0 && nfft > 0 && hop > 0 && frames > 0, "vad_labels: bad sizes");
    DVAE_CHECK_ARG((frames - 1) * hop + nfft <= n + hop, "vad_labels: %lld frames need more samples than n + hop = %lld",
                   (long long)frames, (long long)(n + hop));
    hipStream_t s = (hipStream_t)stream;
    double* energy = (double*)workspace;
    const unsigned blocks = (unsigned)cdiv(frames, 4);
    if (in_f64) hipLaunchKernelGGL(frame_energy_kernel<double>, dim3(blocks), dim3(256), 0, s, (const double*)y, n, nfft, hop, frames, energy);
    else hipLaunchKernelGGL(frame_energy_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)y, n, nfft, hop, frames, energy);
    DVAE_LAUNCH_OK("frame_energy_kernel");
    hipLaunchKernelGGL(vad_threshold_kernel, dim3(1), dim3(1024), 0, s, energy, frames, pow(10.0, vad_threshold), vad);
    DVAE_LAUNCH_OK("vad_threshold_kernel");
    return 0;
}

extern "C" size_t dvae_ibm_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int dvae_ibm_labels(const void* S, int64_t rows, int64_t cols, float eps, float ibm_threshold, const float* vad_gate,
                               float* mask, void* workspace, void* stream) {
    DVAE_CHECK_ARG(S && mask && workspace, "ibm_labels: null argument");
    DVAE_CHECK_ARG(rows > 0 && cols > 0, "ibm_labels: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int64_t count = rows * cols;
    const int nparts = (int)(cdiv(count, 256) < 1024 ? cdiv(count, 256) : 1024);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(ibm_max_kernel, dim3(nparts), dim3(256), 0, s, (const float2*)S, count, partial);
    DVAE_LAUNCH_OK("ibm_max_kernel");
    const unsigned blocks = (unsigned)(cdiv(count, 256) < 65536 ? cdiv(count, 256) : 65536);
    hipLaunchKernelGGL(ibm_mask_kernel, dim3(blocks), dim3(256), 0, s, (const float2*)S, count, partial, nparts, eps, ibm_threshold, vad_gate, cols, mask);
    DVAE_LAUNCH_OK("ibm_mask_kernel");
    return 0;
}
