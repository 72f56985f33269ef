// MFMA tile machinery shared by the fused train step (train_fused.hip) and the MCEM kernels (mcem.hip):
// operand policies (exact fp32 / bf16), fragment-major weight fetch through buffer loads, the
// register-ring GEMM block, C-tile epilogue helpers.
#pragma once
#include "common.hpp"

namespace dvae {
namespace fused {

constexpr int XD = 513, HD = 128, ZD = 16;
constexpr int XP = 528;   // 513 input features padded to a multiple of 16
constexpr int NO = 544;   // 513 output features padded to 17 row tiles of 32
constexpr int TB = 32;    // frames per tile
constexpr int NT_OUT = 17;
// Weight-copy layout.  false: row-major [out][in_padded] (each wave load touches 32 rows x 32 B).
// true: fragment-major [k-step][row tile][lane][E] (one contiguous 1 KB per wave load), fetched with buffer
// loads (see WRef below; with flat 64-bit addresses this layout spilled address pairs and was 3-4x slower).
// The stash (written and read once, by different kernels) is fragment-major as well.
constexpr bool WFRAG = true;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct PolF32 {
    typedef float T;
    typedef f32x4 Frag;
    typedef f32x4 Pack4;
    static constexpr int E = 4;        // elements per 16-byte fragment
    static constexpr int NP = 1;       // operand planes (2 = bf16 hi + lo, see PolX3)
    static constexpr int BDMAX = 4;    // LDS activation fragments read ahead of their MFMA (k-steps)
    static constexpr int KSTEP = 8;    // reduction depth per fragment pair
    static constexpr int PD = 6;       // weight fragments in flight per wave (k-steps ahead of the MFMAs)
    static constexpr int PRE = 2;      // of those, requested before the previous layer's epilogue
    static constexpr int PRE128 = 2;   // the same for the 128-deep layers
    static constexpr int PREBIG = 2;   // long GEMMs whose prefetch site has registers to spare
    static constexpr int WRING = 4;    // wgrad: k-steps of operand fragments in flight per wave
    static constexpr bool EARLY_Y = false;
    static constexpr bool XFULL = false;   // fp32 x tile does not fit LDS next to fp32 activations: streamed in 128-column slices
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ T cvt(float v) { return v; }
    static __device__ __forceinline__ Frag ones() { return Frag{1.f, 1.f, 1.f, 1.f}; }
    static __device__ __forceinline__ float tanh_(float v) { return tanhf(v); }
    static __device__ __forceinline__ float exp_(float v) { return expf(v); }
    static __device__ __forceinline__ float log_(float v) { return logf(v); }
    static __device__ __forceinline__ float div_(float a, float b) { return a / b; }
};

struct PolBF16 {
    typedef __bf16 T;
    typedef bf16x8 Frag;
    typedef bf16x4 Pack4;
    static constexpr int E = 8;
    static constexpr int NP = 1;
    static constexpr int BDMAX = 4;
    static constexpr int KSTEP = 16;
    static constexpr int PD = 16;
    static constexpr int PRE = 6;
#ifndef DVAE_PREBIG_BF16
#define DVAE_PREBIG_BF16 12
#endif
    static constexpr int PREBIG = DVAE_PREBIG_BF16;
    static constexpr int PRE128 = 8;   // 128-deep layers: the whole weight tile of a wave (8 fragments) is requested ahead; with 6, the
                                       // last two arrive one L2 round trip (~0.5 us) after the GEMM starts, in every one of ~12 such phases
#ifndef DVAE_WRING_BF16
#define DVAE_WRING_BF16 8
#endif
    static constexpr int WRING = DVAE_WRING_BF16;
    static constexpr bool EARLY_Y = true;   // request the y tile before the x GEMM (68 VGPRs held across it)
    static constexpr bool XFULL = true;    // whole fp32 x tile stays in LDS for the loss epilogue
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ T cvt(float v) { return (__bf16)v; }
    static __device__ __forceinline__ Frag ones() {
        const __bf16 o = (__bf16)1.0f;
        return Frag{o, o, o, o, o, o, o, o};
    }
    // throughput mode: hardware exp2/log2/rcp based transcendentals
    // raw v_exp_f32 / v_log_f32 (base 2): arguments here are never denormal (x + eps >= 1e-8), so the
    // denormal-scaling sequence __expf / __logf wrap around them (~10 VALU each) is dropped; at one wave
    // per SIMD every VALU instruction costs 4 issue cycles and the loss epilogue is VALU-bound
    static __device__ __forceinline__ float exp_(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }
    static __device__ __forceinline__ float log_(float v) { return __builtin_amdgcn_logf(v) * 0.693147180559945309f; }
    static __device__ __forceinline__ float tanh_(float v) {
        const float e = __builtin_amdgcn_exp2f(v * 2.88539008177792681f);
        return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    }
    static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
};

// Split-bf16 ("bf16x3"): every MFMA operand is a PAIR of bf16 planes, v = hi + lo with hi = bf16(v), lo = bf16(v - hi)
// (16 mantissa bits), and a product takes three MFMAs: hi*hi + lo*hi + hi*lo (lo*lo is below fp32 resolution).
// tools/exp_precision.py: with ANY operand role left at one bf16 the weight gradients are off by 2e-3..4e-2 of their
// maximum at 8192 frames; with every operand split they hold 1e-4 (losses 1e-7) -- the parity-grade throughput mode,
// at 3/16 of the fp32-MFMA cost.  Planes sit at fixed offsets: LDS activations `Pl<P>::lds` elements apart, weight
// copies and stash tiles a run-time plane stride apart; the layouts inside a plane are exactly PolBF16's.
struct PolX3 : PolBF16 {
    static constexpr int NP = 2;
    static constexpr int PD = 8;       // k-steps of weight fragments in flight (x 2 planes = the registers of PolBF16's 16)
    static constexpr int PRE = 3;
    static constexpr int PRE128 = 4;
    static constexpr int PREBIG = 6;
#ifndef DVAE_WRING_X3
#define DVAE_WRING_X3 4
#endif
    static constexpr int WRING = DVAE_WRING_X3;
    static constexpr bool EARLY_Y = false;
    static constexpr bool XFULL = false;   // two operand planes fill the LDS: the fp32 x tile streams in 128-column slices
};

// Ring depths for the 8-wave rows kernel (train_rows2.hip): two waves per SIMD leave 256 registers per wave
#ifndef R2_PD_X3
#define R2_PD_X3 4
#endif
#ifndef R2_PRE_X3
#define R2_PRE_X3 2
#endif
#ifndef R2_PD_BF
#define R2_PD_BF 10
#endif
#ifndef R2_PRE_BF
#define R2_PRE_BF 4
#endif
#ifndef R2_BD_X3
#define R2_BD_X3 2
#endif
#ifndef R2_DBIG_X3
#define R2_DBIG_X3 8
#endif
#ifndef R2_PBIG_X3
#define R2_PBIG_X3 2
#endif
#ifndef R2_D128_X3
#define R2_D128_X3 8
#endif
#ifndef R2_P128_X3
#define R2_P128_X3 2
#endif
#ifndef R2_DBIG_BF
#define R2_DBIG_BF 16
#endif
#ifndef R2_PBIG_BF
#define R2_PBIG_BF 4
#endif
#ifndef R2_PDO_X3
#define R2_PDO_X3 6
#endif
// Split-FP16 operands for the ONE GEMM whose 16-bit planes limit the policy (round 4): the x block of encoder layer 1 (and of the M2_info
// classifier's layer 1).  tools/r04/sim_l1x.py: with every operand split into bf16 planes the worst gradient element is 7.5e-5 ... 3.3e-4 of
// its tensor's maximum -- all of it from this GEMM, whose pre-activations sum heavy-tailed power spectra (x up to 1e4) so that
// 2^-17 of the LARGEST term moves a unit on the knee of tanh; with this GEMM exact the whole step holds 1.4e-5.  fp16 planes carry
// 11 + 11 mantissa bits at the MFMA rate of bf16 (v_mfma_f32_32x32x16_f16) and v_mfma keeps fp16 subnormals (tools/r04/mfma_f16_denorm.hip,
// measured), so a FIXED power-of-two scale serves: x * 2^-3 (finite up to 5.2e5 = twice the largest power a peak-normalised 1024-point
// Hann frame can hold; an element below 2^-21 keeps an absolute error of 2^-22, i.e. 3e-8 |w| in a pre-activation of order one), the
// weights * 2^6 (finite up to 1023; lo planes normal from |w| = 2e-3), products hi*hi + lo*hi + hi*lo as before, the accumulator * 2^-3.
// An input beyond the range turns the step's loss into NaN (inf - inf in the lo plane), never into a silently wrong number.
struct X16 {
    static constexpr float XS = 0.125f, WS = 64.f, ACC = 0.125f, XINV = 8.f;
    template <typename T> static __device__ __forceinline__ T hi(float v) { const _Float16 h = (_Float16)v; return __builtin_bit_cast(T, h); }
    template <typename T> static __device__ __forceinline__ float val(T b) { return (float)__builtin_bit_cast(_Float16, b); }
};

struct PolX3v2 : PolX3 {
    static constexpr int PDO = R2_PDO_X3;     // ring depth of the train-step instantiation (loss epilogue on the helper waves)
    static constexpr int PD = R2_PD_X3, PRE = R2_PRE_X3, PRE128 = R2_P128_X3, PREBIG = R2_PBIG_X3;   // PD / PRE: the output-layer loop (the register peak)
    static constexpr int DBIG = R2_DBIG_X3, D128 = R2_D128_X3;   // ring depths of the long GEMMs / the 128-deep layers
    static constexpr int BDMAX = R2_BD_X3;
#ifndef R2_XF16
#define R2_XF16 1
#endif
    static constexpr bool XF16 = R2_XF16 != 0;     // the x block of layer 1 in split fp16 (see struct X16)
};
struct PolBF16v2 : PolBF16 {
    static constexpr bool XF16 = false;
    static constexpr int PDO = R2_PD_BF;
    static constexpr int PD = R2_PD_BF, PRE = R2_PRE_BF, PRE128 = 8, PREBIG = R2_PBIG_BF;
    static constexpr int DBIG = R2_DBIG_BF, D128 = 8;
    static constexpr bool XFULL = false, EARLY_Y = false;
};

// LDS row strides (elements): an odd number of 16-byte slots per row
template <typename T> struct Ld {
    static constexpr int per16 = 16 / (int)sizeof(T);
    static constexpr int u = NO + per16;          // [frame][544 features]   (x / y / da)
    static constexpr int hh = HD + per16;         // [frame][128]
    static constexpr int z = 32 + per16;          // [frame][32]              (z | pad, dmu | dlv)
    static constexpr int xt = 129;                // fp32 [frame][128] slice of x for the loss epilogue
    static constexpr int nbias = 4 * HD + 32 + NO + HD;   // b1 b2 [bmu|blv] b3 b4 b5(padded), last row of W5: fp32 copies for the epilogues
    static constexpr int ninfo = 6 * HD + 8;         // M2_info: bc1 bc2 wc3 ba1 ba2 wa3, bc3, ba3
    static constexpr int xf_floats = (TB * XD + 63) / 64 * 64;
    static constexpr int xt_floats = (TB * xt + 63) / 64 * 64;
    static constexpr int act_elems = TB * (u + 2 * hh + z);      // one plane of U | Ha | Hb | Zb
    static constexpr size_t float_bytes(bool xfull) { return (size_t)((xfull ? xf_floats : xt_floats) + nbias + ninfo) * sizeof(float) + 64; }
};
// LDS plane stride (elements) between the hi and lo images of the activation buffers; dynamic LDS bytes of the rows kernel
template <typename P> struct Pl {
    static constexpr int lds = P::NP == 2 ? Ld<typename P::T>::act_elems : 0;
    static constexpr size_t bytes = (size_t)Ld<typename P::T>::act_elems * P::NP * sizeof(typename P::T) + Ld<typename P::T>::float_bytes(P::XFULL);
};

__device__ __forceinline__ int feat_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// First weight fragments of a GEMM, requested ahead of time (weights never depend on data, so the
// next layer's first fragments are in flight across the current layer's epilogue and barrier).
template <typename P, int NSTEPS, int PREN = P::PRE> struct WPre {
    static constexpr int N = NSTEPS < PREN ? NSTEPS : PREN;
    typename P::Frag a[N > 0 ? N : 1][P::NP];
};

// Weight fragments are fetched with buffer loads: one resource descriptor for the whole weight-copy
// buffer, the per-lane byte offset in ONE VGPR and the (matrix, row tile, k-step) offset in an SGPR.
// With plain 64-bit global addresses hipcc materialises one VGPR address pair per in-flight fragment
// whenever the k-step stride exceeds the 4 KB immediate range, spills them, and reloads each from
// scratch behind an s_waitcnt vmcnt(0) in front of every weight load (measured: 600 cycles per k-step).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WRef { int voff; unsigned soff; unsigned pl; };      // per-lane byte offset (VGPR), wave-uniform byte offset (SGPR), byte stride between operand planes (SGPR)

template <typename P>
__device__ __forceinline__ typename P::Frag wload(__amdgpu_buffer_rsrc_t rs, WRef r, unsigned step_bytes) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, r.voff, r.soff + step_bytes, 0);
    return __builtin_bit_cast(typename P::Frag, v);
}
// all planes of one weight fragment
template <typename P>
__device__ __forceinline__ void wloadp(typename P::Frag (&a)[P::NP], __amdgpu_buffer_rsrc_t rs, WRef r, unsigned step_bytes) {
    a[0] = wload<P>(rs, r, step_bytes);
    if constexpr (P::NP == 2) a[1] = wload<P>(rs, r, step_bytes + r.pl);
}
// all planes of one LDS activation fragment
template <typename P>
__device__ __forceinline__ void bloadp(typename P::Frag (&b)[P::NP], const typename P::T* p) {
    b[0] = *reinterpret_cast<const typename P::Frag*>(p);
    if constexpr (P::NP == 2) b[1] = *reinterpret_cast<const typename P::Frag*>(p + Pl<P>::lds);
}
// acc += a * b over all plane pairs that matter: hi*hi, lo*hi, hi*lo
// `blo` (wave-uniform): the B operand has a non-zero lo plane (false for label tiles that one bf16 plane holds exactly)
template <typename P>
__device__ __forceinline__ void mmap(f32x16& acc, const typename P::Frag (&a)[P::NP], const typename P::Frag (&b)[P::NP], bool blo = true) {
    if constexpr (P::NP == 2) { P::mma(acc, a[1], b[0]); if (blo) P::mma(acc, a[0], b[1]); }
    P::mma(acc, a[0], b[0]);
}
// the same on operand planes that hold fp16 bit patterns (struct X16)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename P>
__device__ __forceinline__ void mmap_f16(f32x16& acc, const typename P::Frag (&a)[P::NP], const typename P::Frag (&b)[P::NP]) {
    static_assert(P::NP == 2 && sizeof(typename P::T) == 2, "split fp16: two 16-bit planes");
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), acc, 0, 0, 0);
}

template <typename P, int NSTEPS, int PREN>
__device__ __forceinline__ void wprefetch(WPre<P, NSTEPS, PREN>& w, __amdgpu_buffer_rsrc_t rs, WRef wr, unsigned WSTR) {
#pragma unroll
    for (int i = 0; i < WPre<P, NSTEPS, PREN>::N; ++i) wloadp<P>(w.a[i], rs, wr, i * WSTR);
    // hipcc otherwise sinks these loads down to their first use (after the epilogue and barrier)
    __builtin_amdgcn_sched_barrier(0);
}

// acc += W[rows of this lane][k-block] * act^T : NSTEPS fragment pairs.  Weights stream straight from
// L2 into a ring of D = min(PD, NSTEPS) fragment registers: the slot an MFMA has just consumed is
// re-requested D steps ahead, so D loads per wave stay in flight (an L2 round trip under load is
// ~1000 cycles, an MFMA step 32).  Activations come from LDS.  Only the outer loop is rolled.
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `after_fill` runs once the ring is requested and before the first MFMA: the place for global STORES
// (the previous layer's stash tile).  vmcnt retires loads and stores in issue order, so a store issued
// right before a load that the next MFMA needs exposes a full write-acknowledge round trip; issued
// here, the store acks overlap the D k-steps the ring already covers.
// DMAX: weight-ring depth at this call site (k-steps in flight); call sites with few live registers ask for more
template <typename P, int NSTEPS, typename Hook = NoHook, int PREN = P::PRE, int DMAX = P::PD>
__device__ __forceinline__ void gemm_block(f32x16& acc, const WPre<P, NSTEPS, PREN>& w, __amdgpu_buffer_rsrc_t rs, WRef wr,
                                           const typename P::T* brow, unsigned WSTR, Hook after_fill = Hook(), bool blo = true) {
    typedef typename P::Frag Frag;
    constexpr int STR = 2 * P::E;        // LDS activations: k-step = 2E consecutive features of a frame row
    constexpr int D = NSTEPS < DMAX ? NSTEPS : DMAX;
    constexpr int NIT = D > 0 ? NSTEPS / D : 0, REM = D > 0 ? NSTEPS % D : 0;
    // activation fragments are read from LDS BD k-steps ahead; BD divides D so ring slots are static across laps
    constexpr int BDM = P::BDMAX;
    constexpr int BD = (D % 4 == 0 && BDM >= 4) ? 4 : ((D % 3 == 0 && BDM >= 3) ? 3 : ((D % 2 == 0 && BDM >= 2) ? 2 : 1));
    static_assert(NIT <= 1 || D % BD == 0, "B ring must divide the weight ring");
    Frag a[D > 0 ? D : 1][P::NP];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        if (i < WPre<P, NSTEPS, PREN>::N) {
#pragma unroll
            for (int q = 0; q < P::NP; ++q) a[i][q] = w.a[i][q];
        } else wloadp<P>(a[i], rs, wr, i * WSTR);
    }
    // Order pins: without them hipcc moves every weight load down to just above the MFMA that
    // consumes it (one exposed L2 round trip per k-step, measured 150 ns/step instead of ~30).
    __builtin_amdgcn_sched_barrier(0);
    after_fill();
    __builtin_amdgcn_sched_barrier(0);
    // B ring: with the order pinned, an LDS read issued right before its MFMA exposes the full LDS
    // latency every k-step (measured ~250 cycles/step on an idle chip); keep BD reads in flight instead.
    Frag bq[BD][P::NP];
#pragma unroll
    for (int i = 0; i < BD; ++i)
        if (i < NSTEPS) bloadp<P>(bq[i], brow + i * STR);
    if (NIT > 1) {
#pragma unroll 1
        for (int c = 0; c < NIT - 1; ++c) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                mmap<P>(acc, a[i], bq[i % BD], blo);
                // D is a multiple of BD whenever NIT > 1, so slot i % BD is static across laps
                bloadp<P>(bq[i % BD], brow + (c * D + i + BD) * STR);
                wloadp<P>(a[i], rs, wr, ((c + 1) * D + i) * WSTR);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (NIT > 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            constexpr int base = (NIT - 1) * D;
            mmap<P>(acc, a[i], bq[(base + i) % BD], blo);
            if (base + i + BD < NSTEPS) bloadp<P>(bq[(base + i) % BD], brow + (base + i + BD) * STR);
            if (i < REM) wloadp<P>(a[i], rs, wr, (NIT * D + i) * WSTR);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
        constexpr int base = NIT * D;
        mmap<P>(acc, a[i], bq[(base + i) % BD], blo);
        if (base + i + BD < NSTEPS) bloadp<P>(bq[(base + i) % BD], brow + (base + i + BD) * STR);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc += W * act^T with the weight fragments already in registers (kernels that reuse one small weight set many
// times, e.g. the Metropolis-Hastings chain): only the LDS activation fragments are fetched, 4 k-steps ahead.
template <typename P, int NSTEPS>
__device__ __forceinline__ void gemm_resident(f32x16& acc, const typename P::Frag (&w)[NSTEPS], const typename P::T* brow) {
    typedef typename P::Frag Frag;
    constexpr int STR = 2 * P::E;
    constexpr int BD = NSTEPS < 4 ? NSTEPS : 4;
    Frag bq[BD];
#pragma unroll
    for (int i = 0; i < BD; ++i) bq[i] = *reinterpret_cast<const Frag*>(brow + i * STR);
#pragma unroll
    for (int i = 0; i < NSTEPS; ++i) {
        P::mma(acc, w[i], bq[i % BD]);
        if (i + BD < NSTEPS) bq[i % BD] = *reinterpret_cast<const Frag*>(brow + (i + BD) * STR);
    }
}

// the same with every operand plane of the policy (split bf16: resident hi / lo fragments, three MFMAs per k-step)
template <typename P, int NSTEPS>
__device__ __forceinline__ void gemm_resident_p(f32x16& acc, const typename P::Frag (&w)[NSTEPS][P::NP], const typename P::T* brow) {
    typedef typename P::Frag Frag;
    constexpr int STR = 2 * P::E;
    constexpr int BD = NSTEPS < 4 ? NSTEPS : 4;
    Frag bq[BD][P::NP];
#pragma unroll
    for (int i = 0; i < BD; ++i) bloadp<P>(bq[i], brow + i * STR);
#pragma unroll
    for (int i = 0; i < NSTEPS; ++i) {
        mmap<P>(acc, w[i], bq[i % BD]);
        if (i + BD < NSTEPS) bloadp<P>(bq[i % BD], brow + (i + BD) * STR);
    }
}

template <typename P> __device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
}

// 16 bias values of this lane's C-tile rows (features fbase + 8g + 4h + 0..3) from the LDS bias table
__device__ __forceinline__ void bias16(const float* bl, int fbase, int h, float (&b)[16]) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bl + fbase + 8 * gq + 4 * h);
        b[4 * gq] = v[0]; b[4 * gq + 1] = v[1]; b[4 * gq + 2] = v[2]; b[4 * gq + 3] = v[3];
    }
}

// Stash layout ("fragment-major"): a matrix of R features x Bp frames is stored as
//   [feature tile (32)][k-step (KSTEP frames)][lane = h*32 + feature%32][E frames]
// i.e. exactly the order in which one wgrad wave-instruction consumes it: every operand load of
// the wgrad kernel and every store here is one contiguous 1 KB block.  A 32-feature tile block
// starts at element 32 * tile * Bp, as in a plain [feature][Bp] matrix.
//
// put_tile: values v[r] of a 32-feature x 32-frame C tile (feature = fbase + feat_of(r,h), frame = l31)
// go to LDS [frame][feature] (the next layer's B operand); the same wave then reads its own 32
// columns back transposed (E consecutive frames of one feature = one fragment) for the stash.
template <typename P>
__device__ __forceinline__ void put_lds(const float (&v)[16], typename P::T* lds, int ldl, int fbase, int l31, int h) {
    typedef typename P::Pack4 Pack4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        Pack4 p;
        p[0] = P::cvt(v[4 * gq]); p[1] = P::cvt(v[4 * gq + 1]); p[2] = P::cvt(v[4 * gq + 2]); p[3] = P::cvt(v[4 * gq + 3]);
        *reinterpret_cast<Pack4*>(lds + l31 * ldl + fbase + 8 * gq + 4 * h) = p;
        if constexpr (P::NP == 2) {          // lo plane: what the hi plane rounded away
            Pack4 q;
            q[0] = P::cvt(v[4 * gq] - (float)p[0]); q[1] = P::cvt(v[4 * gq + 1] - (float)p[1]);
            q[2] = P::cvt(v[4 * gq + 2] - (float)p[2]); q[3] = P::cvt(v[4 * gq + 3] - (float)p[3]);
            *reinterpret_cast<Pack4*>(lds + Pl<P>::lds + l31 * ldl + fbase + 8 * gq + 4 * h) = q;
        }
    }
}

// inverse of put_lds: the 16 C-tile values of this lane back from the operand plane(s) (hi + lo for PolX3)
template <typename P>
__device__ __forceinline__ void get_lds(float (&v)[16], const typename P::T* lds, int ldl, int fbase, int l31, int h) {
    typedef typename P::Pack4 Pack4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const Pack4 p = *reinterpret_cast<const Pack4*>(lds + l31 * ldl + fbase + 8 * gq + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * gq + j] = (float)p[j];
        if constexpr (P::NP == 2) {
            const Pack4 q = *reinterpret_cast<const Pack4*>(lds + Pl<P>::lds + l31 * ldl + fbase + 8 * gq + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * gq + j] += (float)q[j];
        }
    }
}

}  // namespace fused
}  // namespace dvae
