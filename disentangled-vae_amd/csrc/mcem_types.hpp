// Arguments of the Metropolis-Hastings chain kernels (mcem.hip: streamed weights; mcem_resident.hip: weight-stationary).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace dvae {
namespace fused {

struct MhArgs {
    const float* Z0;      // (16, N)            initial latents                       [MH mode]
    const float* y;       // (ydim, N) or null
    const float* g;       // (N)
    const float* Vb;      // (513, N)
    const float* X2;      // (513, N)
    const float* noise;   // (nit, 16, N)
    const float* logu;    // (nit, N)
    float* Zs;            // (N, R, 16)  MH mode: written (R = nit - burnin); decode mode: read
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

// weight-stationary chain (mcem_resident.hip): every operand policy, label rows 0 / 1..16
bool resident_chain_supported(int precision, int yp);
int launch_resident_chain(int precision, int yp, const MhArgs& a, hipStream_t s);
// the same chain on 16-frame tiles (mcem_resident16.hip; a.ntiles counts 16-frame tiles): launch_resident_chain takes it when
// the 16-frame tiles of the call fit the chip in one round (DVAE_MCEM_TILE=16 / 32 forces either)
bool resident16_chain_supported(int precision, int yp);
int launch_resident16_chain(int precision, int yp, const MhArgs& a, hipStream_t s);

}  // namespace fused
}  // namespace dvae
