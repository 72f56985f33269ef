// MCEM speech enhancement on the GPU (reference: packages/models/mcem.py): the Metropolis-Hastings
// chain over the VAE latents as ONE persistent launch per E-step, the NMF multiplicative updates as
// two HBM-streaming launches, the Wiener gains as one.  See include/dvae_mcem.h.
//
// MH kernels: since round 4 the chain runs in csrc/mcem_resident.hip (the decoder resident in the CU's registers) for every policy and label
// variant; the STREAMING kernel below stays for (F, N) matrices of 2 GB and more and as the A/B reference (DVAE_MCEM_CHAIN=stream).
// Streaming MH kernel: one 256-thread workgroup owns a tile of 32 frames for the whole chain (frames are
// independent given g and Vb).  The decoder (tanh 128, tanh 128, exp 513) runs on the MFMA tile
// machinery of the train step (transposed orientation: the lane is the frame), so the
// log-likelihood terms, the accept test and the masked update never leave registers / LDS:
//   * the label part of decoder layer 1 (W3[:, 16:] y + b3) does not change along the chain: computed
//     once per tile and kept in 16 registers;
//   * the reference evaluates the decoder twice per iteration (proposal, then the updated state,
//     mcem.py:247,268); the decoder acts frame by frame, so keeping the per-frame likelihood of the
//     current state is equivalent and halves the work;
//   * the per-frame likelihood sum over 513 bins is accumulated in double.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fused_tiles.hpp"
#include "mcem_types.hpp"
#include "../../include/dvae_mcem.h"

namespace dvae {
namespace fused {

template <typename T> struct MhLds {
    static constexpr int per16 = 16 / (int)sizeof(T);
    static constexpr int hh = HD + per16, z = 32 + per16;
    static constexpr int nbias = 2 * HD + NO;
    static constexpr int plane(int yp) { return TB * (2 * hh + z + (yp ? yp + per16 : 0)); }      // elements of one operand plane
    static constexpr size_t bytes(int yp, int np = 1) {
        return (size_t)plane(yp) * np * sizeof(T) + (size_t)nbias * sizeof(float) + 64;
    }
};

// The chain reuses one small weight set thousands of times per launch.  RES = 0 streams every layer from L2 through
// the fragment ring (lean: 168 VGPRs, three workgroups per CU); RES = 1 keeps decoder layers 1-2 in registers and
// streams the output layer through a 16-deep ring; RES = 2 keeps all of it (bf16: 196 registers -- hipcc spills
// instead of using AGPRs for the fragments, so it is not selected).  A chain step is issue-bound (MFMA, then the
// VALU-heavy likelihood epilogue, in the same waves), which is why co-resident workgroups pay off.
// fp32 chain policy: exact fp32 products; the 513-bin likelihood epilogue (exp, log, divide per bin and per chain
// step) is VALU-bound at one wave per SIMD, so it uses the hardware exp2 / log2 / rcp units (1-2 ulp) instead of
// the ~60-instruction libm expansions: 8 us of a 23 us chain step on MI355X.
struct PolF32Deep : PolF32 {
    static constexpr int PD = 16; static constexpr int PRE = 8;
    static __device__ __forceinline__ float exp_(float v) { return __expf(v); }
    static __device__ __forceinline__ float log_(float v) { return __logf(v); }
    static __device__ __forceinline__ float div_(float a, float b) { return __fdividef(a, b); }
    static __device__ __forceinline__ float tanh_(float v) {
        const float e = __expf(2.f * v);                       // tanh = 1 - 2 / (e^{2v} + 1); saturates cleanly at +-1
        return 1.f - __fdividef(2.f, e + 1.f);
    }
};

struct PolF32Lean : PolF32Deep { static constexpr int PD = 6; static constexpr int PRE = 2; };

// Split-bf16 chain policy (DVAE_PREC_BF16X3): every operand a (hi, lo) pair of bf16 planes, three MFMAs per product -- 16 mantissa
// bits at 3/16 of the exact-fp32 matrix cost (the train step's parity-grade policy, fused_tiles.hpp).  Streaming variant (RES 0); the
// lo planes of the LDS activation buffers sit one MhLds plane above the hi planes, those of the weight copies MhArgs::wpl bytes up.
template <int YP> struct PolX3M : PolX3 { static constexpr int PD = 6; static constexpr int PRE = 2; };
template <int YP> struct Pl<PolX3M<YP>> { static constexpr int lds = MhLds<__bf16>::plane(YP); };

// Workgroups per CU the fp32 chain kernel is compiled for.  3 caps it at 168 registers (29 - 37 of them spill to scratch); 2 gives it 256
// and no spill.  Measured (round 3, same box, alternating; tools/bench_mcem.py): E-step of one utterance 1027 us either way, 25 utterances
// side by side 161 - 162 (3) against 154 - 161 (2) utterances / s: the spilled values are touched once per chain step, the third
// workgroup hides more latency than they cost.
#ifndef MCEM_OCC
#define MCEM_OCC 3
#endif
template <typename P, int YP, int RES>
__global__ __launch_bounds__(256, RES == 0 ? (P::NP == 2 ? 2 : MCEM_OCC) : 1) void mcem_mh_kernel(const MhArgs g) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    constexpr int E = P::E, KS = P::KSTEP;
    constexpr int LDH = MhLds<T>::hh, LDZ = MhLds<T>::z, LDY = YP + MhLds<T>::per16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(P::NP == 1 || RES == 0, "the split-bf16 chain streams its weights");
    T* Ha = reinterpret_cast<T*>(smem);
    T* Hb = Ha + TB * LDH;
    T* Zb = Hb + TB * LDH;
    T* Yb = Zb + TB * LDZ;
    float* Bias = reinterpret_cast<float*>(Ha + MhLds<T>::plane(YP) * P::NP);       // behind the operand plane(s)
    constexpr int OB3 = 0, OB4 = HD, OB5 = 2 * HD;
    __shared__ double red[4][32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int fb = 32 * wave;
    constexpr int FB = 64 * E;
    constexpr unsigned SZ = sizeof(T);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned S4 = 4 * FB * SZ, S17 = NT_OUT * FB * SZ, TSTEP = FB * SZ;
    constexpr unsigned KB3 = (ZD / KS) * 4 * FB * SZ;
    const WRef W3r{lane * 16, (unsigned)(g.oW3 * SZ) + (unsigned)wave_u * TSTEP, g.wpl};
    const WRef W4r{lane * 16, (unsigned)(g.oW4 * SZ) + (unsigned)wave_u * TSTEP, g.wpl};
    const WRef W5r{lane * 16, (unsigned)(g.oW5 * SZ), g.wpl};
    auto woff = [](WRef r, unsigned bytes) { return WRef{r.voff, r.soff + bytes, r.pl}; };
    const T* const Har = Ha + l31 * LDH + h * E;
    const T* const Hbr = Hb + l31 * LDH + h * E;
    const T* const Zbr = Zb + l31 * LDZ + h * E;
    const T* const Ybr = Yb + l31 * LDY + h * E;

    for (int i = tid; i < MhLds<T>::nbias; i += 256) Bias[i] = g.bias[i];
    // resident weight fragments of this wave
    constexpr int NT5 = (NT_OUT + 3) / 4;                      // output tiles per wave (wave 0: 5, others 4)
    Frag w3zR[ZD / KS][P::NP], w4R[RES >= 1 ? HD / KS : 1], w5R[RES == 2 ? NT5 : 1][HD / KS];
#pragma unroll
    for (int i = 0; i < ZD / KS; ++i) wloadp<P>(w3zR[i], wrs, W3r, i * S4);
    if constexpr (RES >= 1) {
#pragma unroll
        for (int i = 0; i < HD / KS; ++i) w4R[i] = wload<P>(wrs, W4r, i * S4);
    }
    if constexpr (RES == 2) {
#pragma unroll
        for (int q = 0; q < NT5; ++q) {
            const int t = wave_u + 4 * q;
            const WRef wr = woff(W5r, (unsigned)(t < NT_OUT ? t : wave_u) * TSTEP);
#pragma unroll
            for (int i = 0; i < HD / KS; ++i) w5R[q][i] = wload<P>(wrs, wr, i * S17);
        }
    }
    __syncthreads();

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int64_t n0 = (int64_t)tile * TB;
        const bool live = n0 + l31 < g.N;
        const int64_t n = live ? n0 + l31 : g.N - 1;          // clamped frame index of this lane
        const float g_n = g.g ? g.g[n] : 1.f;

        // ---- label part of decoder layer 1, once per tile: c1 = W3[:, 16:] y + b3 ----
        float c1[16];
        bias16(Bias + OB3, fb, h, c1);
        if (YP > 0) {
            for (int idx = tid; idx < TB * YP; idx += 256) {
                const int f = idx >> 5, fr = idx & 31;          // consecutive threads: consecutive frames of one label row
                float v = 0.f;
                if (f < g.ydim && n0 + fr < g.N) v = g.y[(int64_t)f * g.N + n0 + fr];
                const T hi = P::cvt(v);
                Yb[fr * LDY + f] = hi;
                if constexpr (P::NP == 2) Yb[Pl<P>::lds + fr * LDY + f] = P::cvt(v - (float)hi);
            }
            __syncthreads();
            f32x16 acc;
            zero_acc<P>(acc);
            WPre<P, YP / KS> w3y;
            wprefetch<P, YP / KS>(w3y, wrs, woff(W3r, KB3), S4);
            gemm_block<P, YP / KS>(acc, w3y, wrs, woff(W3r, KB3), Ybr, S4);
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[r] += acc[r];
        }

        // latent state of this lane's frame (wave 0): features 4h..4h+3 and 8+4h..8+4h+3
        float z[8], zp[8];
        float prior_cur = 0.f;
        double ll_cur = 0.0;
        const int mstart = g.nit > 0 ? -1 : 0;
        const int mend = g.nit > 0 ? g.nit : 0;
        if (wave == 0 && g.nit > 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) z[r] = g.Z0[(int64_t)feat_of(r, h) * g.N + n];
        }

        // one decoder pass over the latents in Zb; EPI(t, f32x16 acc) handles output tile t of this wave
        auto decode = [&](auto&& epi_pre, auto&& epi) {
            f32x16 acc;
            zero_acc<P>(acc);
            gemm_resident_p<P, ZD / KS>(acc, w3zR, Zbr);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = P::tanh_(acc[r] + c1[r]);
            put_lds<P>(v, Ha, LDH, fb, l31, h);
            __syncthreads();
            zero_acc<P>(acc);
            WPre<P, HD / KS> w5;
            if constexpr (RES >= 1) {
                if constexpr (RES != 2) wprefetch<P, HD / KS>(w5, wrs, woff(W5r, wave_u * TSTEP), S17);
                gemm_resident<P, HD / KS>(acc, w4R, Har);
            } else {
                WPre<P, HD / KS> w4;
                wprefetch<P, HD / KS>(w4, wrs, W4r, S4);
                gemm_block<P, HD / KS>(acc, w4, wrs, W4r, Har, S4);
                wprefetch<P, HD / KS>(w5, wrs, woff(W5r, wave_u * TSTEP), S17);
            }
            float bv[16];
            bias16(Bias + OB4, fb, h, bv);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = P::tanh_(acc[r] + bv[r]);
            put_lds<P>(v, Hb, LDH, fb, l31, h);
            __syncthreads();
            if constexpr (RES == 2) {
#pragma unroll
                for (int q = 0; q < NT5; ++q) {
                    const int t = wave_u + 4 * q;
                    __builtin_amdgcn_sched_barrier(0);      // keep one tile's operand loads in flight at a time
                    if (t < NT_OUT) {
                        epi_pre(t);
                        zero_acc<P>(acc);
                        gemm_resident<P, HD / KS>(acc, w5R[q], Hbr);
                        bias16(Bias + OB5, 32 * t, h, bv);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += bv[r];
                        epi(t, acc);
                    }
                }
            } else {
#pragma unroll 1
                for (int t = wave_u; t < NT_OUT; t += 4) {
                    zero_acc<P>(acc);
                    const WRef wr = woff(W5r, (unsigned)t * TSTEP);
                    gemm_block<P, HD / KS>(acc, w5, wrs, wr, Hbr, S17, [&]() { epi_pre(t); });
                    // the next tile's first fragments (the last tile wraps to the first tile of the next decoder pass)
                    wprefetch<P, HD / KS>(w5, wrs, woff(W5r, (unsigned)(t + 4 < NT_OUT ? t + 4 : wave_u) * TSTEP), S17);
                    bias16(Bias + OB5, 32 * t, h, bv);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += bv[r];
                    epi(t, acc);
                }
            }
        };

        for (int m = mstart; m < mend; ++m) {
            float prior_p = 0.f;
            if (wave == 0) {
                if (m >= 0) {
                    const float* nz = g.noise + ((int64_t)m * ZD) * g.N + n;
#pragma unroll
                    for (int r = 0; r < 8; ++r) zp[r] = z[r] + g.sd * nz[(int64_t)feat_of(r, h) * g.N];   // mcem.py:244
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) zp[r] = z[r];
                }
                float zv[16];
#pragma unroll
                for (int r = 0; r < 8; ++r) { zv[r] = zp[r]; zv[r + 8] = 0.f; prior_p += zp[r] * zp[r]; }
                put_lds<P>(zv, Zb, LDZ, 0, l31, h);
                prior_p += __shfl_xor(prior_p, 32, 64);
            }
            __syncthreads();
            double ll = 0.0;
            float xs[16], vbs[16];
            decode(
                [&](int t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int f = 32 * t + feat_of(r, h); f = f < XD ? f : XD - 1;
                        xs[r] = g.X2[(int64_t)f * g.N + n];
                        vbs[r] = g.Vb[(int64_t)f * g.N + n];
                    }
                },
                [&](int t, const f32x16& a) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool ok = 32 * t + feat_of(r, h) < XD;
                        const float vx = fmaf(g_n, P::exp_(a[r]), vbs[r]);            // mcem.py:248-249
                        const float term = P::log_(vx) + P::div_(xs[r], vx);           // mcem.py:252-253
                        s += ok ? term : 0.f;
                    }
                    ll += (double)s;
                });
            ll += __shfl_xor(ll, 32, 64);
            if (h == 0) red[wave][l31] = ll;
            __syncthreads();
            if (wave == 0) {
                const double ll_p = red[0][l31] + red[1][l31] + red[2][l31] + red[3][l31];
                if (m < 0) {
                    ll_cur = ll_p; prior_cur = prior_p;
                } else {
                    const float acc_prob = (float)(ll_cur - ll_p) + 0.5f * (prior_cur - prior_p);   // mcem.py:252-254
                    const bool is_acc = g.logu[(int64_t)m * g.N + n] < acc_prob;                    // mcem.py:257
                    if (is_acc) {
                        ll_cur = ll_p; prior_cur = prior_p;
#pragma unroll
                        for (int r = 0; r < 8; ++r) z[r] = zp[r];
                    }
                    if (live && h == 0) {
                        if (g.accp) g.accp[(int64_t)m * g.N + n] = acc_prob;
                        if (g.accd) g.accd[(int64_t)m * g.N + n] = is_acc ? 1 : 0;
                    }
                    if (m >= g.burnin && live) {                                                    // mcem.py:271-273
                        float* dst = g.Zs + ((int64_t)n * g.R + (m - g.burnin)) * ZD;
                        *reinterpret_cast<f32x4*>(dst + 4 * h) = f32x4{z[0], z[1], z[2], z[3]};
                        *reinterpret_cast<f32x4*>(dst + 8 + 4 * h) = f32x4{z[4], z[5], z[6], z[7]};
                    }
                }
            }
            // red[] and Zb are next written after the barriers of the following decoder pass
        }

        // ---- speech variances of the sampled latents: Vs[r] = decoder([Zs[:, r, :] | y])  (mcem.py:280-290) ----
        if (g.Vs != nullptr) {
            for (int r_s = 0; r_s < g.R; ++r_s) {
                __syncthreads();
                if (wave == 0) {
                    const float* src = g.Zs + ((int64_t)n * g.R + r_s) * ZD;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + 4 * h);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + 8 + 4 * h);
                    float zv[16];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { zv[r] = a0[r]; zv[4 + r] = a1[r]; zv[8 + r] = 0.f; zv[12 + r] = 0.f; }
                    put_lds<P>(zv, Zb, LDZ, 0, l31, h);
                }
                __syncthreads();
                float* const vs_r = g.Vs + (int64_t)r_s * XD * g.N;
                decode([&](int) {},
                       [&](int t, const f32x16& a) {
#pragma unroll
                           for (int r = 0; r < 16; ++r) {
                               const int f = 32 * t + feat_of(r, h);
                               if (live && f < XD) vs_r[(int64_t)f * g.N + n] = P::exp_(a[r]);
                           }
                       });
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// weight pack: nn.Linear [rows][ld] fp32 -> fragment-major [k-step][row tile][lane][E] copy
// dst_lo (optional): the lo plane of the split-bf16 policy, what the first plane's rounding left
template <typename T>
__global__ void mcem_pack_kernel(const float* __restrict__ src, int rows, int cols, int ld, T* __restrict__ dst, int nt, int ksteps, T* __restrict__ dst_lo = nullptr) {
    constexpr int E = 16 / (int)sizeof(T), KS = 2 * E;
    const int64_t total = (int64_t)ksteps * nt * 64 * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        const int ln = (int)((i / E) % 64);
        const int tile = (int)((i / (64 * E)) % nt);
        const int ks = (int)(i / ((int64_t)64 * E * nt));
        const int row = 32 * tile + (ln & 31), col = ks * KS + (ln >> 5) * E + e;
        const float v = (row < rows && col < cols) ? src[(int64_t)row * ld + col] : 0.f;
        const T hi = (T)v;
        dst[i] = hi;
        if (dst_lo) dst_lo[i] = (T)(v - (float)hi);
    }
}

__global__ void mcem_bias_kernel(const float* b3, const float* b4, const float* b5, float* dst) {
    for (int i = threadIdx.x; i < 2 * HD + NO; i += blockDim.x)
        dst[i] = i < HD ? b3[i] : (i < 2 * HD ? b4[i - HD] : (i - 2 * HD < XD ? b5[i - 2 * HD] : 0.f));
}

// ---------------------------------------------------------------------------------------------
// M-step (EM.M_step, mcem.py:91-153).  Vx[r] = g Vs[r] + Vb is never materialised.
// Batched form: the frame axis holds U utterances ("segments"), each starting on a multiple of 32 frames
// (seg_start) with seg_count valid frames; W, Wun, norms and cost carry a leading U axis.  Null tables = one
// utterance covering all N frames.
constexpr int KMAX = 16;
constexpr int FT = 1024;          // threads of the frames kernel: 32 frames x 32 bin groups

// W update: one WAVE per (frequency bin f, utterance u), four bins per workgroup:
//   num[k] = sum_n X2 sum_r Vx^-2 H[k,n],  den[k] = sum_n sum_r Vx^-1 H[k,n],  Wun = W sqrt(num / den)   (mcem.py:107-111)
// (a 256-thread workgroup per bin left each thread ~1 frame of work: launch/drain bound, 0.9 TB/s at 76 utterances)
__global__ __launch_bounds__(256) void mstep_w_kernel(const float* __restrict__ X2, const float* __restrict__ Vs, int R, int64_t N, int K,
                                                      const float* __restrict__ W, const float* __restrict__ H, const float* __restrict__ g,
                                                      const float* __restrict__ Vb, float* __restrict__ Wun,
                                                      const int* __restrict__ seg_start, const int* __restrict__ seg_count) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 4 + wave, u = blockIdx.y;
    if (f >= XD) return;
    const int64_t nbeg = seg_start ? seg_start[u] : 0, nend = nbeg + (seg_count ? seg_count[u] : N);
    float num[KMAX], den[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { num[k] = 0.f; den[k] = 0.f; }
    const int64_t FN = (int64_t)XD * N;
    const float* vs_f = Vs + (int64_t)f * N;
    for (int64_t n = nbeg + lane; n < nend; n += 64) {
        const float vb = Vb[(int64_t)f * N + n], gn = g[n], x2 = X2[(int64_t)f * N + n];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 5
        for (int r = 0; r < R; ++r) {
            const float inv = 1.f / fmaf(gn, vs_f[r * FN + n], vb);
            a1 += inv; a2 += inv * inv;
        }
        const float p2 = x2 * a2;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) { const float hk = H[(int64_t)k * N + n]; num[k] = fmaf(p2, hk, num[k]); den[k] = fmaf(a1, hk, den[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            const float a = wave_sum(num[k]), b = wave_sum(den[k]);
            if (lane == k) {
                const int64_t o = ((int64_t)u * XD + f) * K + k;
                Wun[o] = W[o] * sqrtf(a / b);
            }
        }
    }
}

// H update, new Vb, g update, cost: one 1024-thread workgroup per 32 frames (lane & 31 = frame, 32 bin groups).
// Everything after the W update is local to a frame, so the three passes over Vs share one launch.
__global__ __launch_bounds__(FT) void mstep_frames_kernel(const float* __restrict__ X2, const float* __restrict__ Vs, int R, int64_t N, int K,
                                                          const float* __restrict__ Wun_all, float* __restrict__ H, float* __restrict__ g,
                                                          float* __restrict__ Vb, float* __restrict__ norms_out, double* __restrict__ partial,
                                                          const int* __restrict__ seg_start, const int* __restrict__ seg_count,
                                                          const int* __restrict__ tile_seg) {
    extern __shared__ __attribute__((aligned(16))) float lw[];     // Wun [513][K], then wave partials [16][2K][32], then sums [2K][32]
    float* wpart = lw + (XD * K + 3) / 4 * 4;
    float* sums = wpart + 16 * 2 * K * 32;
    __shared__ float nrm[KMAX];
    __shared__ double redc[16];
    const int tid = threadIdx.x, fr = tid & 31, grp = tid >> 5, lane = tid & 63, wave = tid >> 6;
    const int u = tile_seg ? tile_seg[blockIdx.x] : 0;
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    const int64_t nend = seg_start ? (int64_t)seg_start[u] + seg_count[u] : N;
    const bool live = n0 + fr < nend;
    const int64_t n = live ? n0 + fr : nend - 1;
    const int64_t FN = (int64_t)XD * N;
    const float* Wun = Wun_all + (int64_t)u * XD * K;
    for (int i = tid; i < XD * K; i += FT) lw[i] = Wun[i];
    __syncthreads();
    // column norms of the un-normalised W (mcem.py:130): 16 threads per column
    if (tid < 16 * K) {
        const int k = tid >> 4, q = tid & 15;
        float s = 0.f;
        for (int f = q; f < XD; f += 16) s += fabsf(lw[f * K + k]);
        s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
        if (q == 0) { nrm[k] = s; if (n0 == (seg_start ? seg_start[u] : 0)) norms_out[u * KMAX + k] = s; }
    }
    float hk[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) hk[k] = k < K ? H[(int64_t)k * N + n] : 0.f;
    const float gn = g[n];

    // sum over the 32 bin groups of `cnt` per-thread values v[]: wave shuffle (2 groups), LDS (16 waves)
    auto group_sums = [&](auto& v, int cnt) {
        constexpr int M = (int)(sizeof(v) / sizeof(float));
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i < cnt) {
                float a = v[i];
                a += __shfl_xor(a, 32, 64);
                if ((lane >> 5) == 0) wpart[(wave * cnt + i) * 32 + fr] = a;
            }
        }
        __syncthreads();
        if (grp < cnt) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) a += wpart[(w * cnt + grp) * 32 + fr];
            sums[grp * 32 + fr] = a;
        }
        __syncthreads();
    };

    // ---- H update (mcem.py:118-123) with Vb = Wun H ----
    float acc[2 * KMAX];
#pragma unroll
    for (int k = 0; k < 2 * KMAX; ++k) acc[k] = 0.f;
    for (int f = grp; f < XD; f += 32) {
        float vb = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) vb = fmaf(lw[f * K + k], hk[k], vb);
        float a1 = 0.f, a2 = 0.f;
        for (int r = 0; r < R; ++r) {
            const float inv = 1.f / fmaf(gn, Vs[r * FN + (int64_t)f * N + n], vb);
            a1 += inv; a2 += inv * inv;
        }
        const float p2 = X2[(int64_t)f * N + n] * a2;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) { const float w = lw[f * K + k]; acc[2 * k] = fmaf(w, p2, acc[2 * k]); acc[2 * k + 1] = fmaf(w, a1, acc[2 * k + 1]); }
    }
    group_sums(acc, 2 * K);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) if (k < K) hk[k] = hk[k] * sqrtf(sums[(2 * k) * 32 + fr] / sums[(2 * k + 1) * 32 + fr]);
    __syncthreads();

    // ---- new Vb = Wun Hnew (mcem.py:126); g update (mcem.py:137-143) ----
    float gv[2] = {0.f, 0.f};
    for (int f = grp; f < XD; f += 32) {
        float vb = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) vb = fmaf(lw[f * K + k], hk[k], vb);
        if (live) Vb[(int64_t)f * N + n] = vb;
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < R; ++r) {
            const float vs = Vs[r * FN + (int64_t)f * N + n];
            const float inv = 1.f / fmaf(gn, vs, vb);
            s1 = fmaf(vs, inv, s1); s2 = fmaf(vs, inv * inv, s2);
        }
        gv[0] = fmaf(X2[(int64_t)f * N + n], s2, gv[0]); gv[1] += s1;
    }
    group_sums(gv, 2);
    const float gnew = gn * sqrtf(sums[fr] / sums[32 + fr]);

    // ---- cost (mcem.py:69-71) with the updated g; H is stored normalised (mcem.py:134) ----
    double c = 0.0;
    for (int f = grp; f < XD; f += 32) {
        float vb = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) vb = fmaf(lw[f * K + k], hk[k], vb);
        const float x2 = X2[(int64_t)f * N + n];
        float s = 0.f;
        for (int r = 0; r < R; ++r) {
            const float vx = fmaf(gnew, Vs[r * FN + (int64_t)f * N + n], vb);
            s += logf(vx) + x2 / vx;
        }
        c += (double)s;
    }
    if (!live) c = 0.0;
    c = wave_sum(c);
    if (lane == 0) redc[wave] = c;
    if (grp == 0 && live) {
        g[n] = gnew;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) H[(int64_t)k * N + n] = hk[k] * nrm[k];
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += redc[w];
        partial[blockIdx.x] = t;
    }
}

// (the register-resident form of this kernel, mstep_frames_reg_kernel, lives in mcem_mstep.hip: it is compiled without SLP vectorisation)
namespace mstep { int launch_w_reg(const float* X2, const float* Vs, int R, int64_t N, int U, const float* W, const float* H, const float* g, const float* Vb,
                                  float* Wun, const int* seg_start, const int* seg_count, const double* partial_prev, float* cost_prev, hipStream_t s);
                  int launch_frames_reg(const float* X2, const float* Vs, int R, int64_t N, int K, const float* Wun, float* H, float* g, float* Vb,
                                       float* norms, double* partial, const int* seg_start, const int* seg_count, const int* tile_seg, float* Wout, hipStream_t s);
                  int frames_per_workgroup(int64_t N); }

// per utterance: W = Wun / norm (mcem.py:132), cost = mean over (R, F, N_u) of the tile partials
__global__ __launch_bounds__(256) void mstep_finish_kernel(const float* __restrict__ Wun, const float* __restrict__ norms, int K, float* __restrict__ W,
                                                           const double* __restrict__ partial, int R, int64_t N,
                                                           const int* __restrict__ seg_start, const int* __restrict__ seg_count,
                                                           float* __restrict__ cost, int tile_frames) {
    const int u = blockIdx.x;
    const int64_t o = (int64_t)u * XD * K;
    for (int i = threadIdx.x; i < XD * K; i += 256) W[o + i] = Wun[o + i] / norms[u * KMAX + i % K];
    const int64_t nbeg = seg_start ? seg_start[u] : 0, cnt = seg_count ? seg_count[u] : N;
    const int t0 = (int)(nbeg / tile_frames), t1 = (int)((nbeg + cnt + tile_frames - 1) / tile_frames);
    __shared__ double red[4];
    if (cost) mstep_cost_of_partials(partial, t0, t1, (double)R * XD * (double)cnt, cost + u, red);
}

// the cost alone (dvae_mcem_cost_flush: the last iteration of a loop of dvae_mcem_em_iteration_lazy calls)
__global__ __launch_bounds__(256) void mstep_cost_kernel(const double* __restrict__ partial, int R, int64_t N, const int* __restrict__ seg_start,
                                                         const int* __restrict__ seg_count, float* __restrict__ cost, int tile_frames) {
    const int u = blockIdx.x;
    const int64_t nbeg = seg_start ? seg_start[u] : 0, cnt = seg_count ? seg_count[u] : N;
    const int t0 = (int)(nbeg / tile_frames), t1 = (int)((nbeg + cnt + tile_frames - 1) / tile_frames);
    __shared__ double red[4];
    mstep_cost_of_partials(partial, t0, t1, (double)R * XD * (double)cnt, cost + u, red);
}

// Wiener gains (compute_WF, mcem.py:321-327)
__global__ __launch_bounds__(256) void wiener_kernel(const float* __restrict__ Vs, int R, int64_t FN, int64_t N, const float* __restrict__ g,
                                                     const float* __restrict__ Vb, float* __restrict__ WFs, float* __restrict__ WFn) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= FN) return;
    const float gn = g[i % N], vb = Vb[i];
    float s = 0.f, q = 0.f;
    for (int r = 0; r < R; ++r) {
        const float vs = gn * Vs[r * FN + i];
        const float inv = 1.f / (vs + vb);
        s = fmaf(vs, inv, s); q = fmaf(vb, inv, q);
    }
    WFs[i] = s / (float)R;
    WFn[i] = q / (float)R;
}

struct McemLayout { int yp, ld3; int64_t oW3, oW4, oW5, elems, bias_off_bytes, total_bytes; };
static McemLayout mcem_layout(int y_dim, int precision) {
    McemLayout L;
    const int esz = precision == DVAE_PREC_F32 ? 4 : (precision == DVAE_PREC_BF16X3 ? 4 : 2);      // bytes per element over all planes
    L.yp = y_dim == 0 ? 0 : (y_dim + 15) / 16 * 16;
    L.ld3 = ZD + L.yp;
    L.oW3 = 0;
    L.oW4 = L.oW3 + (int64_t)HD * L.ld3;
    L.oW5 = L.oW4 + (int64_t)HD * HD;
    L.elems = L.oW5 + (int64_t)NO * HD;
    L.bias_off_bytes = (L.elems * esz + 255) / 256 * 256;
    L.total_bytes = L.bias_off_bytes + (2 * HD + NO) * (int64_t)sizeof(float) + 256;
    return L;
}

template <typename P, int YP, int RES>
static int launch_mh(const MhArgs& a, hipStream_t s) {
    const size_t lds = MhLds<typename P::T>::bytes(YP, P::NP);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)mcem_mh_kernel<P, YP, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(mcem_mh_kernel, %zu B LDS): %s", lds, hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    hipLaunchKernelGGL((mcem_mh_kernel<P, YP, RES>), dim3(a.ntiles), dim3(256), lds, s, a);
    DVAE_LAUNCH_OK("mcem_mh_kernel");
    return 0;
}

static unsigned long long* g_mcem_dbg = nullptr;      // set by dvae_mcem_debug_stamps

// Z (16, N) <- the last kept sample of every frame's chain, Zs (N, R, 16): EM.run's `self.Z = Z_sampled_t[:, -1, :].T` (mcem.py:234, 300)
__global__ __launch_bounds__(256) void last_sample_kernel(const float* __restrict__ Zs, int R, int64_t N, float* __restrict__ Z) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // i = n * 16 + l: coalesced read of the sample, strided 4-byte writes (64 KB in all)
    if (i >= N * ZD) return;
    const int64_t n = i / ZD;
    const int l = (int)(i - n * ZD);
    Z[(int64_t)l * N + n] = Zs[(n * R + (R - 1)) * ZD + l];
}


static int run_mh(const dvae_mcem_plan_t* plan, const void* wcopy, MhArgs& a, hipStream_t s) {
    a.dbg = g_mcem_dbg;
    const McemLayout L = mcem_layout(plan->y_dim, plan->precision);
    a.ydim = plan->y_dim;
    a.ntiles = (int)((a.N + TB - 1) / TB);
    a.wcopy = wcopy; a.wcopy_bytes = L.bias_off_bytes;
    a.oW3 = L.oW3; a.oW4 = L.oW4; a.oW5 = L.oW5;
    a.bias = (const float*)((const char*)wcopy + L.bias_off_bytes);
    const bool bf = plan->precision == DVAE_PREC_BF16;
    a.wpl = plan->precision == DVAE_PREC_BF16X3 ? (unsigned)(L.elems * 2) : 0u;
    // label rows 0 / 1..16: the weight-stationary chain (mcem_resident.hip); DVAE_MCEM_CHAIN=stream, the 513-row labels and
    // (F, N) matrices of 2 GB and more: the streaming kernels below
    static const bool stream_only = [] { const char* e = getenv("DVAE_MCEM_CHAIN"); return e && !strcmp(e, "stream"); }();
    if (!stream_only && resident_chain_supported(plan->precision, L.yp) && (int64_t)XD * a.N * 4 < ((int64_t)1 << 31) &&
        (int64_t)a.nit * ZD * a.N * 4 < ((int64_t)1 << 31))
        return launch_resident_chain(plan->precision, L.yp, a, s);
    // the streaming kernels do not write the final state themselves
    struct LastSample {
        const MhArgs& a; hipStream_t s;
        int operator()(int rc) const {
            if (rc == 0 && a.Zlast != nullptr && a.nit > 0) {
                hipLaunchKernelGGL(last_sample_kernel, dim3((unsigned)((a.N * ZD + 255) / 256)), dim3(256), 0, s, a.Zs, a.R, a.N, a.Zlast);
                DVAE_LAUNCH_OK("last_sample_kernel");
            }
            return rc;
        }
    } then{a, s};
    if (plan->precision == DVAE_PREC_BF16X3) {
        if (L.yp == 0) return then(launch_mh<PolX3M<0>, 0, 0>(a, s));
        if (L.yp == 16) return then(launch_mh<PolX3M<16>, 16, 0>(a, s));
        if (L.yp == 528) return then(launch_mh<PolX3M<528>, 528, 0>(a, s));
    }
    // fp32: lean streaming variant (168 VGPRs, 3 workgroups per CU co-resident: 3.2 us per tile and chain at >= 768
    // tiles against 4.3 us with resident layers 1-2 at one workgroup per CU; equal at <= 256 tiles).
    // bf16: layers 1-2 resident (1.11 us per tile against 1.18 us).  Measured with tools/exp_mcem_occupancy.py.
    if (L.yp == 0) return then(bf ? launch_mh<PolBF16, 0, 1>(a, s) : launch_mh<PolF32Lean, 0, 0>(a, s));
    if (L.yp == 16) return then(bf ? launch_mh<PolBF16, 16, 1>(a, s) : launch_mh<PolF32Lean, 16, 0>(a, s));
    if (L.yp == 528) return then(bf ? launch_mh<PolBF16, 528, 1>(a, s) : launch_mh<PolF32Lean, 528, 0>(a, s));
    set_error("mcem: y_dim %d not supported (0, 1..16, 513)", plan->y_dim);
    return DVAE_E_BADARG;
}

}  // namespace fused
}  // namespace dvae

using namespace dvae;
using namespace dvae::fused;

extern "C" int dvae_mcem_debug_stamps(void* buf) {
    g_mcem_dbg = (unsigned long long*)buf;
    return 0;
}

extern "C" int dvae_mcem_plan(int y_dim, int precision, dvae_mcem_plan_t* plan) {
    DVAE_CHECK_ARG(plan != nullptr, "mcem_plan: null plan");
    DVAE_CHECK_ARG(y_dim == 0 || (y_dim >= 1 && y_dim <= 16) || y_dim == XD, "mcem_plan: y_dim %d not supported (0, 1..16, 513)", y_dim);
    DVAE_CHECK_ARG(precision == DVAE_PREC_F32 || precision == DVAE_PREC_BF16 || precision == DVAE_PREC_BF16X3, "mcem_plan: bad precision %d", precision);
    const McemLayout L = mcem_layout(y_dim, precision);
    memset(plan, 0, sizeof(*plan));
    plan->y_dim = y_dim; plan->precision = precision; plan->x_dim = XD; plan->z_dim = ZD; plan->h_dim = HD;
    plan->weights_bytes = L.total_bytes;
    return 0;
}

extern "C" int dvae_mcem_pack(const dvae_mcem_plan_t* plan, const float* W3, int ld3, const float* b3, const float* W4, int ld4,
                              const float* b4, const float* W5, int ld5, const float* b5, void* weights, void* stream) {
    DVAE_CHECK_ARG(plan && W3 && b3 && W4 && b4 && W5 && b5 && weights, "mcem_pack: null argument");
    DVAE_CHECK_ARG(ld3 >= ZD + plan->y_dim && ld4 >= HD && ld5 >= HD, "mcem_pack: row strides too small");
    const McemLayout L = mcem_layout(plan->y_dim, plan->precision);
    hipStream_t s = (hipStream_t)stream;
    if (plan->precision == DVAE_PREC_BF16 || plan->precision == DVAE_PREC_BF16X3) {
        __bf16* w = (__bf16*)weights;
        __bf16* const lo = plan->precision == DVAE_PREC_BF16X3 ? w + L.elems : nullptr;       // second plane right behind the first
        hipLaunchKernelGGL(mcem_pack_kernel<__bf16>, dim3(64), dim3(256), 0, s, W3, HD, ZD + plan->y_dim, ld3, w + L.oW3, 4, L.ld3 / 16, lo ? lo + L.oW3 : nullptr);
        hipLaunchKernelGGL(mcem_pack_kernel<__bf16>, dim3(64), dim3(256), 0, s, W4, HD, HD, ld4, w + L.oW4, 4, HD / 16, lo ? lo + L.oW4 : nullptr);
        hipLaunchKernelGGL(mcem_pack_kernel<__bf16>, dim3(128), dim3(256), 0, s, W5, XD, HD, ld5, w + L.oW5, NT_OUT, HD / 16, lo ? lo + L.oW5 : nullptr);
    } else {
        float* w = (float*)weights;
        hipLaunchKernelGGL(mcem_pack_kernel<float>, dim3(64), dim3(256), 0, s, W3, HD, ZD + plan->y_dim, ld3, w + L.oW3, 4, L.ld3 / 8, (float*)nullptr);
        hipLaunchKernelGGL(mcem_pack_kernel<float>, dim3(64), dim3(256), 0, s, W4, HD, HD, ld4, w + L.oW4, 4, HD / 8, (float*)nullptr);
        hipLaunchKernelGGL(mcem_pack_kernel<float>, dim3(128), dim3(256), 0, s, W5, XD, HD, ld5, w + L.oW5, NT_OUT, HD / 8, (float*)nullptr);
    }
    hipLaunchKernelGGL(mcem_bias_kernel, dim3(1), dim3(256), 0, s, b3, b4, b5, (float*)((char*)weights + L.bias_off_bytes));
    DVAE_LAUNCH_OK("mcem_pack");
    return 0;
}

static int mcem_sample_impl(const dvae_mcem_plan_t* plan, const void* weights, const float* Z0, const float* y, const float* g,
                            const float* Vb, const float* X2, const float* noise, const float* logu, int nit, int burnin,
                            float var_rw, int64_t N, float* Zs, float* Vs, float* acc_logratio, unsigned char* accepted, float* Zlast, void* stream) {
    DVAE_CHECK_ARG(plan && weights && Z0 && g && Vb && X2 && noise && logu && Zs, "mcem_sample: null argument");
    DVAE_CHECK_ARG((plan->y_dim == 0) == (y == nullptr), "mcem_sample: y must be given exactly when the plan has y_dim > 0");
    DVAE_CHECK_ARG(N > 0 && nit > 0 && burnin >= 0 && burnin < nit, "mcem_sample: need N > 0 and 0 <= burnin < nit (N=%lld nit=%d burnin=%d)", (long long)N, nit, burnin);
    DVAE_CHECK_ARG(var_rw >= 0.f, "mcem_sample: negative random-walk variance");
    MhArgs a;
    memset(&a, 0, sizeof(a));
    a.Z0 = Z0; a.y = y; a.g = g; a.Vb = Vb; a.X2 = X2; a.noise = noise; a.logu = logu; a.Zs = Zs; a.Vs = Vs; a.accp = acc_logratio; a.accd = accepted;
    a.nit = nit; a.burnin = burnin; a.R = nit - burnin; a.N = N; a.sd = sqrtf(var_rw);
    a.Zlast = Zlast;
    return run_mh(plan, weights, a, (hipStream_t)stream);
}

extern "C" int dvae_mcem_sample(const dvae_mcem_plan_t* plan, const void* weights, const float* Z0, const float* y, const float* g,
                                const float* Vb, const float* X2, const float* noise, const float* logu, int nit, int burnin,
                                float var_rw, int64_t N, float* Zs, float* Vs, float* acc_logratio, unsigned char* accepted, void* stream) {
    return mcem_sample_impl(plan, weights, Z0, y, g, Vb, X2, noise, logu, nit, burnin, var_rw, N, Zs, Vs, acc_logratio, accepted, nullptr, stream);
}

extern "C" int dvae_mcem_decode(const dvae_mcem_plan_t* plan, const void* weights, const float* Zs, const float* y, int R, int64_t N,
                                float* Vs, void* stream) {
    DVAE_CHECK_ARG(plan && weights && Zs && Vs, "mcem_decode: null argument");
    DVAE_CHECK_ARG((plan->y_dim == 0) == (y == nullptr), "mcem_decode: y must be given exactly when the plan has y_dim > 0");
    DVAE_CHECK_ARG(N > 0 && R > 0, "mcem_decode: need N > 0 and R > 0");
    MhArgs a;
    memset(&a, 0, sizeof(a));
    a.y = y; a.Zs = const_cast<float*>(Zs); a.Vs = Vs; a.nit = 0; a.burnin = 0; a.R = R; a.N = N;
    return run_mh(plan, weights, a, (hipStream_t)stream);
}

static size_t mstep_ws_layout(int64_t N, int K, int U, size_t* o_norms, size_t* o_partial) {
    size_t o = ((size_t)U * XD * K * sizeof(float) + 255) / 256 * 256;
    *o_norms = o; o += ((size_t)U * KMAX * sizeof(float) + 255) / 256 * 256;
    *o_partial = o; o += ((size_t)((N + 3) / 4) * sizeof(double) + 255) / 256 * 256;        // one cost partial per workgroup of the frames kernel (4, 8 or 16 frames)
    return o;
}

extern "C" size_t dvae_mcem_m_step_workspace_bytes(int64_t N, int K, int U) {
    size_t a, b;
    return mstep_ws_layout(N, K, U < 1 ? 1 : U, &a, &b);
}

// lazy: W normalised by the frames kernel, no finish launch; the cost stays in the workspace's partial sums until the next lazy call (which
// writes it to cost_prev from its W update) or dvae_mcem_cost_flush
static int m_step_impl(const float* X2, const float* Vs, int R, int64_t N, int K, int U, const int* seg_start,
                       const int* seg_count, const int* tile_seg, float* W, float* H, float* g, float* Vb,
                       float* cost, bool lazy, float* cost_prev, void* workspace, void* stream) {
    DVAE_CHECK_ARG(X2 && Vs && W && H && g && Vb && workspace, "mcem_m_step: null argument");
    DVAE_CHECK_ARG(R > 0 && N > 0 && K > 0 && K <= KMAX && U > 0, "mcem_m_step: need R > 0, N > 0, U > 0, 0 < K <= %d", KMAX);
    DVAE_CHECK_ARG((seg_start != nullptr) == (seg_count != nullptr) && (seg_start != nullptr) == (tile_seg != nullptr),
                   "mcem_m_step: the three segment tables go together");
    DVAE_CHECK_ARG(seg_start != nullptr || U == 1, "mcem_m_step: U > 1 needs the segment tables");
    DVAE_CHECK_ARG(seg_start == nullptr || N % 32 == 0, "mcem_m_step: batched frame axis must be padded to a multiple of 32");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    size_t o_norms, o_partial;
    mstep_ws_layout(N, K, U, &o_norms, &o_partial);
    float* Wun = (float*)ws;
    float* norms = (float*)(ws + o_norms);
    double* partial = (double*)(ws + o_partial);
    const int ntiles = (int)((N + 31) / 32);
    // DVAE_MSTEP=3pass keeps the round-1 kernels (three reads of Vs, one wave-step of loads in flight in the W update); R > 10 samples and
    // ranks other than 10 always take them
    const char* mk = getenv("DVAE_MSTEP");
    const bool reg_form = R <= 10 && K == 10 && (int64_t)R * XD * N * 4 < (int64_t)0x7fffffff && !(mk && strcmp(mk, "3pass") == 0);      // (rank 10: mcem.py / scripts/evaluate_ntcd_M2.py:66)
    if (lazy && !reg_form) { set_error("mcem_em_iteration_lazy: exists for the register-resident M-step only (R <= 10 samples, rank 10): use dvae_mcem_em_iteration"); return DVAE_E_UNSUPPORTED; }
    if (reg_form) {
        const int rcw = mstep::launch_w_reg(X2, Vs, R, N, U, W, H, g, Vb, Wun, seg_start, seg_count, lazy ? partial : nullptr, lazy ? cost_prev : nullptr, s);
        if (rcw) return rcw;
    } else {
        hipLaunchKernelGGL(mstep_w_kernel, dim3((XD + 3) / 4, U), dim3(256), 0, s, X2, Vs, R, N, K, W, H, g, Vb, Wun, seg_start, seg_count);
        DVAE_LAUNCH_OK("mstep_w_kernel");
    }
    if (reg_form) {
        const int rcf = mstep::launch_frames_reg(X2, Vs, R, N, K, Wun, H, g, Vb, norms, partial, seg_start, seg_count, tile_seg, lazy ? W : nullptr, s);
        if (rcf) return rcf;
        if (lazy) return 0;
        hipLaunchKernelGGL(mstep_finish_kernel, dim3(U), dim3(256), 0, s, Wun, norms, K, W, partial, R, N, seg_start, seg_count, cost, mstep::frames_per_workgroup(N));
        DVAE_LAUNCH_OK("mstep_finish_kernel");
        return 0;
    }
    const size_t lds = ((size_t)(XD * K + 3) / 4 * 4 + 16 * 2 * K * 32 + 2 * K * 32) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        const size_t lds_max = ((size_t)(XD * KMAX + 3) / 4 * 4 + 16 * 2 * KMAX * 32 + 2 * KMAX * 32) * sizeof(float);
        hipError_t e = hipFuncSetAttribute((const void*)mstep_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(mstep_frames_kernel, %zu B LDS): %s", lds_max, hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    hipLaunchKernelGGL(mstep_frames_kernel, dim3(ntiles), dim3(FT), lds, s, X2, Vs, R, N, K, Wun, H, g, Vb, norms, partial, seg_start, seg_count, tile_seg);
    DVAE_LAUNCH_OK("mstep_frames_kernel");
    hipLaunchKernelGGL(mstep_finish_kernel, dim3(U), dim3(256), 0, s, Wun, norms, K, W, partial, R, N, seg_start, seg_count, cost, 32);
    DVAE_LAUNCH_OK("mstep_finish_kernel");
    return 0;
}

extern "C" int dvae_mcem_m_step_batch(const float* X2, const float* Vs, int R, int64_t N, int K, int U, const int* seg_start,
                                      const int* seg_count, const int* tile_seg, float* W, float* H, float* g, float* Vb,
                                      float* cost, void* workspace, void* stream) {
    return m_step_impl(X2, Vs, R, N, K, U, seg_start, seg_count, tile_seg, W, H, g, Vb, cost, false, nullptr, workspace, stream);
}

// One EM iteration = the body of EM.run's loop (mcem.py:156-160: E_step, M_step, cost) as ONE host call: the chain launch (with the decoder
// variances of its kept samples and Z <- last kept sample), the M-step's launches.  Nothing here synchronises or allocates; between two calls
// the host only has to point at the next iteration's draws.
extern "C" int dvae_mcem_em_iteration(const dvae_mcem_plan_t* plan, const void* weights, float* Z, const float* y, float* g, float* Vb,
                                      const float* X2, const float* noise, const float* logu, int nit, int burnin, float var_rw, int64_t N,
                                      int K, int U, const int* seg_start, const int* seg_count, const int* tile_seg, float* W, float* H,
                                      float* Zs, float* Vs, float* cost, void* workspace, void* stream) {
    DVAE_CHECK_ARG(Z && Zs && Vs, "mcem_em_iteration: Z, Zs and Vs are required");
    // (Z <- the chain's last kept sample is written by the chain launch itself: Zlast = Z)
    int rc = mcem_sample_impl(plan, weights, Z, y, g, Vb, X2, noise, logu, nit, burnin, var_rw, N, Zs, Vs, nullptr, nullptr, Z, stream);
    if (rc) return rc;
    const int R = nit - burnin;
    return dvae_mcem_m_step_batch(X2, Vs, R, N, K, U, seg_start, seg_count, tile_seg, W, H, g, Vb, cost, workspace, stream);
}

// The same iteration with two launches of the M-step instead of three: W is normalised by the frames kernel (its workgroups share the rows
// out), and the iteration's cost stays in the workspace as the frames kernel's partial sums -- the NEXT lazy call forms it in its W update
// (cost_prev, U floats, or NULL in the first call), dvae_mcem_cost_flush after the last.  Same arithmetic, same bits as
// dvae_mcem_em_iteration; one utterance of 300 frames: 6.5 us of an iteration's 170.  The workspace must not be touched between the calls.
extern "C" int dvae_mcem_em_iteration_lazy(const dvae_mcem_plan_t* plan, const void* weights, float* Z, const float* y, float* g, float* Vb,
                                           const float* X2, const float* noise, const float* logu, int nit, int burnin, float var_rw, int64_t N,
                                           int K, int U, const int* seg_start, const int* seg_count, const int* tile_seg, float* W, float* H,
                                           float* Zs, float* Vs, float* cost_prev, void* workspace, void* stream) {
    DVAE_CHECK_ARG(Z && Zs && Vs, "mcem_em_iteration_lazy: Z, Zs and Vs are required");
    int rc = mcem_sample_impl(plan, weights, Z, y, g, Vb, X2, noise, logu, nit, burnin, var_rw, N, Zs, Vs, nullptr, nullptr, Z, stream);
    if (rc) return rc;
    return m_step_impl(X2, Vs, nit - burnin, N, K, U, seg_start, seg_count, tile_seg, W, H, g, Vb, nullptr, true, cost_prev, workspace, stream);
}

extern "C" int dvae_mcem_cost_flush(int R, int64_t N, int K, int U, const int* seg_start, const int* seg_count, float* cost, void* workspace, void* stream) {
    DVAE_CHECK_ARG(cost && workspace && R > 0 && N > 0 && U > 0 && K > 0 && K <= KMAX, "mcem_cost_flush: bad argument");
    DVAE_CHECK_ARG((seg_start != nullptr) == (seg_count != nullptr) && (seg_start != nullptr || U == 1), "mcem_cost_flush: U > 1 needs the segment tables");
    size_t o_norms, o_partial;
    mstep_ws_layout(N, K, U, &o_norms, &o_partial);
    hipLaunchKernelGGL(mstep_cost_kernel, dim3(U), dim3(256), 0, (hipStream_t)stream, (const double*)((char*)workspace + o_partial), R, N, seg_start, seg_count, cost,
                       mstep::frames_per_workgroup(N));
    DVAE_LAUNCH_OK("mstep_cost_kernel");
    return 0;
}

extern "C" int dvae_mcem_m_step(const float* X2, const float* Vs, int R, int64_t N, int K, float* W, float* H, float* g, float* Vb,
                                float* cost, void* workspace, void* stream) {
    return dvae_mcem_m_step_batch(X2, Vs, R, N, K, 1, nullptr, nullptr, nullptr, W, H, g, Vb, cost, workspace, stream);
}

extern "C" int dvae_mcem_wiener(const float* Vs, int R, int64_t N, const float* g, const float* Vb, float* WFs, float* WFn, void* stream) {
    DVAE_CHECK_ARG(Vs && g && Vb && WFs && WFn, "mcem_wiener: null argument");
    DVAE_CHECK_ARG(R > 0 && N > 0, "mcem_wiener: need R > 0 and N > 0");
    const int64_t FN = (int64_t)XD * N;
    hipLaunchKernelGGL(wiener_kernel, dim3((unsigned)((FN + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Vs, R, FN, N, g, Vb, WFs, WFn);
    DVAE_LAUNCH_OK("wiener_kernel");
    return 0;
}
