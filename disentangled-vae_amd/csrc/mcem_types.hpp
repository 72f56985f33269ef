// Arguments of the Metropolis-Hastings chain kernels (mcem.hip: streamed weights; mcem_resident.hip: weight-stationary).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "common.hpp"

namespace dvae {
namespace fused {

// cost of one utterance = mean over (R, F, N_u) of its tiles' partial sums (mcem.py:69-71): ONE form for mstep_finish_kernel (mcem.hip) and for
// the W update that forms the previous iteration's cost (dvae_mcem_em_iteration_lazy) -- the same additions in the same order, the same bits.
// 256 threads; red: four doubles of LDS.
__device__ __forceinline__ void mstep_cost_of_partials(const double* __restrict__ partial, int t0, int t1, double denom, float* out, double* red) {
    double c = 0.0;
    for (int i = t0 + (int)threadIdx.x; i < t1; i += 256) c += partial[i];
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) *out = (float)((red[0] + red[1] + red[2] + red[3]) / denom);
}


struct MhArgs {
    const float* Z0;      // (16, N)            initial latents                       [MH mode]
    const float* y;       // (ydim, N) or null
    const float* g;       // (N)
    const float* Vb;      // (513, N)
    const float* X2;      // (513, N)
    const float* noise;   // (nit, 16, N)
    const float* logu;    // (nit, N)
    float* Zs;            // (N, R, 16)  MH mode: written (R = nit - burnin); decode mode: read
    float* Zlast;         // (16, N) or null: the chain's final state = its last kept sample (EM.run's `self.Z = Z_sampled_t[:, -1, :].T`); may be Z0 itself
    float* Vs;            // (R, 513, N) or null
    float* accp;          // (nit, N) log acceptance ratios, optional
    unsigned char* accd;  // (nit, N) decisions, optional
    int ydim, nit, burnin, R, ntiles;
    int64_t N;
    float sd;
    const void* wcopy; int64_t wcopy_bytes;
    int64_t oW3, oW4, oW5;     // element offsets of the fragment-major copies
    unsigned wpl;              // bytes between the hi and lo planes of the copies (split-bf16 policy)
    const float* bias;         // b3[128] b4[128] b5[544]
    unsigned long long* dbg;   // diagnostic: per (workgroup, wave) sums of shader clocks per chain phase (dvae_mcem_debug_stamps), null in production
};

// Sums across the lane groups of a wave on gfx950's VALU half / quarter exchanges instead of ds_bpermute_b32 (an LDS round trip each, and the
// second waits for the first: two of them were 250 of wave 0's 1 240 serial clocks per chain step).  v_permlane32_swap vdst, src: lanes 32-63
// of vdst <-> lanes 0-31 of src; v_permlane16_swap: the odd rows of 16 of vdst <-> the even rows of src.  With both operands = x the two
// results are (lower | lower) and (upper | upper): their sum is x[lane] + x[lane ^ 32] (^ 16) in every lane -- the operands of each addition
// are those of `x += __shfl_xor(x, 32)`, so the results are bit-identical to the shuffle form.
// (Written out: with the builtins hipcc / ROCm 7.2 emitted `v_add_f32 v, v, v` behind a float swap -- it added the first result to itself, also
// with the second operand made a value of its own -- while the double form came out right.  The s_nop covers the two wait states a VALU write
// of an operand needs before a permlane swap reads it.)
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float xsum32(float x) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    swap32(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xsum16(float x) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    swap16(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ double xsum32(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    unsigned alo = (unsigned)u, ahi = (unsigned)(u >> 32), blo = alo, bhi = ahi;
    swap32(alo, blo);
    swap32(ahi, bhi);
    return __builtin_bit_cast(double, (unsigned long long)alo | ((unsigned long long)ahi << 32)) +
           __builtin_bit_cast(double, (unsigned long long)blo | ((unsigned long long)bhi << 32));
}
__device__ __forceinline__ double xsum16(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    unsigned alo = (unsigned)u, ahi = (unsigned)(u >> 32), blo = alo, bhi = ahi;
    swap16(alo, blo);
    swap16(ahi, bhi);
    return __builtin_bit_cast(double, (unsigned long long)alo | ((unsigned long long)ahi << 32)) +
           __builtin_bit_cast(double, (unsigned long long)blo | ((unsigned long long)bhi << 32));
}

// weight-stationary chain (mcem_resident.hip): every operand policy, label rows 0 / 1..16
bool resident_chain_supported(int precision, int yp);
int launch_resident_chain(int precision, int yp, const MhArgs& a, hipStream_t s);
// the same chain on 16-frame tiles (mcem_resident16.hip; a.ntiles counts 16-frame tiles): launch_resident_chain takes it when
// the 16-frame tiles of the call fit the chip in one round (DVAE_MCEM_TILE=16 / 32 forces either)
bool resident16_chain_supported(int precision, int yp);
int launch_resident16_chain(int precision, int yp, const MhArgs& a, hipStream_t s);
// the exact-fp32 chain on 4-frame tiles (mcem_resident4.hip: v_mfma_f32_4x4x1_16b_f32; a.ntiles counts 4-frame tiles): taken while those fit the
// chip in two rounds (DVAE_MCEM_TILE=4 forces it)
bool resident4_chain_supported(int precision, int yp);
int launch_resident4_chain(int yp, const MhArgs& a, hipStream_t s);

}  // namespace fused
}  // namespace dvae
