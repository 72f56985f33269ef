// The weight-stationary Metropolis-Hastings chain (mcem_resident.hip) on SIXTEEN-frame tiles, for chains too short to fill the chip with
// 32-frame tiles: ONE utterance of 300 frames -- what scripts/evaluate_ntcd_M2.py runs per process (packages/models/mcem.py:207-290) --
// is ten 32-frame tiles on 256 CUs.  A chain step is bound by its epilogue (three transcendentals per bin and frame: exp of the
// pre-activation, log and reciprocal of the mixture variance, 64 of a step's elements per lane), not by the matrix pipe, so half the
// frames per workgroup is most of the way to half the time per step, on twice the CUs (round 5: 5.5 -> 3.6 us per step at 300 frames).
//
// Same decomposition as the 32-frame kernel -- one 256-thread workgroup per tile, one wave per SIMD, the whole decoder in the lane's 512
// registers for the launch, the tile's X2 / Vb on chip for the chain, four workgroup barriers per step -- on v_mfma_f32_16x16x32_bf16
// (split bf16 = three MFMAs per product, and plain bf16) or v_mfma_f32_16x16x4_f32 (exact fp32 products: MFMA-bound, half the frames = half
// the matrix work per workgroup):
//   * the lane is (frame f = lane & 15, quarter q = lane >> 4): a 16 x 16 C tile holds rows 4 q .. 4 q + 3 of frame f in the lane's four
//     accumulator registers; an operand fragment is the 16-byte chunk (row or frame = lane & 15, chunk q) of a k-step of four chunks (32
//     deep in bf16; 16 deep in fp32, where MFMA j of the k-step takes element j of the chunk: lane quarter q supplies k = 4 q + j);
//   * the weight copies are the 32-row fragment-major ones of the 32-frame kernel ([k-step of 16][32-row tile][lane][8]): both shapes are made
//     of the same 16-byte chunks (row, k octet), so a lane gathers its chunk of a (16-row tile, k-step of 32) fragment by its own offset
//     -- once per launch;
//   * output layer: wave w owns bins 128 w .. 128 w + 127 as eight 16-row tiles (64 fragments with both planes = 256 registers, 240 of
//     them accumulation registers named by the MFMA statements themselves, as in the 32-frame kernel: see its hazard table); bin 512 is
//     the same fp32 dot product finished by wave 3; layers 1 and 2: wave w owns features 32 w .. 32 w + 31 as two row tiles;
//   * the likelihood terms of output tile t run between the MFMAs of tile t + 1, one bin per k-step.
// The arithmetic per element is that of the 32-frame kernel (hardware exp2 / log2 / rcp, per-tile float sums, double across tiles); the
// partition of a frame's 513 terms over lanes differs, so the two kernels agree to rounding, not bit for bit.  Which one runs depends on
// the frame count only (launch_resident_chain, mcem_resident.hip): a given call is deterministic.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "fused_tiles.hpp"
#include "mcem_types.hpp"
#include "../../include/dvae_mcem.h"

namespace dvae {
namespace fused {

struct PolX3S : PolX3 {};
struct PolB1S : PolBF16 {};
struct PolF32S : PolF32 {            // exact fp32 products (v_mfma_f32_16x16x4_f32, four per 16-deep k-step); epilogue on the hardware exp2 / log2 / rcp
                                     // units, as the 32-frame fp32 chain (mcem_resident.hip: PolF32C)
    static __device__ __forceinline__ float exp_(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }
    static __device__ __forceinline__ float log_(float v) { return __builtin_amdgcn_logf(v) * 0.693147180559945309f; }
    static __device__ __forceinline__ float tanh_(float v) {
        const float e = __builtin_amdgcn_exp2f(v * 2.88539008177792681f);
        return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    }
    static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
};
struct PolX3SY : PolX3S {};          // the label image of a tile (513 label rows), its own plane stride
struct PolB1SY : PolB1S {};
struct PolF32SY : PolF32S {};
constexpr int T16 = 16;                                                  // frames per tile
// LDS rows ([frame][feature]): one 16-byte chunk of padding; a k-step is four chunks (32 deep in bf16, 16 deep in fp32)
template <typename T> struct SL {
    static constexpr int E = 16 / (int)sizeof(T), KW = 4 * E;
    static constexpr int ldh = HD + E, ldz = (ZD > KW ? ZD : KW) + E, ldy = NO + E;
    static constexpr int plane = T16 * (2 * ldh + ldz);                  // elements of one operand plane: h1, h2, latents
};
constexpr int S_LDC = HD + 4;
template <> struct Pl<PolX3S> { static constexpr int lds = SL<__bf16>::plane; };
template <> struct Pl<PolB1S> { static constexpr int lds = 0; };
template <> struct Pl<PolF32S> { static constexpr int lds = 0; };
template <> struct Pl<PolX3SY> { static constexpr int lds = T16 * SL<__bf16>::ldy; };
template <> struct Pl<PolB1SY> { static constexpr int lds = 0; };
template <> struct Pl<PolF32SY> { static constexpr int lds = 0; };
template <typename P> struct YPolS;
template <> struct YPolS<PolX3S> { typedef PolX3SY type; };
template <> struct YPolS<PolB1S> { typedef PolB1SY type; };
template <> struct YPolS<PolF32S> { typedef PolF32SY type; };

constexpr size_t S_O_X2 = (size_t)SL<__bf16>::plane * 2 * sizeof(__bf16);      // (one fp32 plane is exactly as large: 16 x (2 x 132 + 20) x 4 bytes)
static_assert((size_t)SL<float>::plane * sizeof(float) <= S_O_X2, "the fp32 activation plane fits the two bf16 planes");
constexpr size_t S_O_VB = S_O_X2 + (size_t)32 * 64 * 4 * sizeof(float);             // X2: [row tile 0..31][lane][4]
constexpr size_t S_O_C1 = S_O_VB + (size_t)32 * 64 * 4 * sizeof(float);             // Vb likewise
constexpr size_t S_O_BIAS = S_O_C1 + (size_t)T16 * S_LDC * sizeof(float);
constexpr size_t S_O_W512 = S_O_BIAS + (size_t)(2 * HD + NO) * sizeof(float);
constexpr size_t S_O_P512 = S_O_W512 + (size_t)HD * sizeof(float);
constexpr size_t S_O_RED = S_O_P512 + (size_t)4 * T16 * sizeof(float);
constexpr size_t S_O_ACC = S_O_RED + (size_t)4 * T16 * sizeof(double);           // [frame]: the last step's accept decision, for the other waves
constexpr size_t S_O_VC = (S_O_ACC + (size_t)T16 * sizeof(int) + 15) / 16 * 16;     // decoder variances of the chain's current state: [row tile 0..31][lane][4]
constexpr size_t S_LDS = S_O_VC + (size_t)32 * 64 * 4 * sizeof(float);
static_assert(S_O_X2 % 16 == 0 && S_O_C1 % 16 == 0 && S_O_BIAS % 16 == 0 && S_O_RED % 8 == 0 && S_LDS <= 160 * 1024, "resident chain (16 frames): LDS layout");
static_assert((size_t)T16 * SL<__bf16>::ldy * 2 * sizeof(__bf16) <= S_O_C1 - S_O_X2 && (size_t)T16 * SL<float>::ldy * sizeof(float) <= S_O_C1 - S_O_X2,
              "the label image fits the X2 / Vb area");

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void sfor16(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor16<I + 1, N>(f); }
}

// acc += a * b over the plane pairs that matter (hi*hi, lo*hi, hi*lo), compiler-scheduled (layers 1 and 2, the label GEMM)
template <typename P>
__device__ __forceinline__ void mm16(f32x4_t& acc, const typename P::Frag (&a)[P::NP], const typename P::Frag (&b)[P::NP]) {
    if constexpr (sizeof(typename P::T) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][j], b[0][j], acc, 0, 0, 0);
        return;
    } else
    if constexpr (P::NP == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}

// One 16-row output tile: acc (holding the bias, read from LDS into the accumulator registers) += W(resident fragments) * h2^T over the four
// k-steps of 32, written out like the chain of the 32-frame kernel (its hazard table applies with the 4-pass figures: the wait states
// behind the last MFMA, 12, are the 8-pass requirement; a leading s_nop 1 covers a VALU-written SrcC should hipcc ever move the bias through
// a VALU copy; dependent MFMAs accumulate back to back).  between(k-step i, slot j): work placed behind the j-th MFMA of k-step i.
template <typename P, int NAG, typename Between>
__device__ __forceinline__ void gemm16_agpr(f32x4_t& acc, const typename P::Frag (&w)[HD / (4 * P::E)][P::NP], const typename P::T* brow, Between&& between) {
    typedef typename P::Frag Frag;
    constexpr int KW = 4 * P::E, NKS = HD / KW;                            // bf16: 4 k-steps of 32; fp32: 8 k-steps of 16
    constexpr bool F32 = sizeof(typename P::T) == 4;
    Frag bq[2][P::NP];
    bloadp<P>(bq[0], brow);
    bloadp<P>(bq[1], brow + KW);
    auto mm = [&](auto first, auto last, auto ag, const Frag& a, const Frag& b) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value, LAST = decltype(last)::value, AG = decltype(ag)::value;
        if constexpr (FIRST) {
            if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        } else if constexpr (LAST) {
            if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
        } else {
            if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
    };
    // fp32: one v_mfma_f32_16x16x4_f32 (8 passes) per fragment element: lane quarter q supplies k = 4 q + j of the k-step to MFMA j
    auto mf = [&](auto first, auto last, auto ag, float a, float b) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value, LAST = decltype(last)::value, AG = decltype(ag)::value;
        if constexpr (FIRST) {
            if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        } else if constexpr (LAST) {
            if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
        } else {
            if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
    };
    typedef std::integral_constant<bool, false> F;
    typedef std::integral_constant<bool, true> Tr;
    sfor16<0, NKS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        typedef std::integral_constant<bool, (i < NAG)> Ag;
        typedef std::integral_constant<bool, i == 0> Fi;
        typedef std::integral_constant<bool, i == NKS - 1> L;
        if constexpr (F32) {
            mf(Fi{}, F{}, Ag{}, w[i][0][0], bq[i & 1][0][0]);
            mf(F{}, F{}, Ag{}, w[i][0][1], bq[i & 1][0][1]);
            between(ic, std::integral_constant<int, 0>{});
            mf(F{}, F{}, Ag{}, w[i][0][2], bq[i & 1][0][2]);
            between(ic, std::integral_constant<int, 1>{});
            mf(F{}, L{}, Ag{}, w[i][0][3], bq[i & 1][0][3]);
        } else if constexpr (P::NP == 2) {
            // (hipcc moves the VALU work freely across the statements; pinning one part of a bin between two MFMAs with scheduling fences
            // measured slower, 4000 against 3600 clocks per step: tools/r05/mfma16_chain_bench.hip -- a dependent chain of these MFMAs issues
            // back to back at 16 clocks each, and any VALU work between two of them costs its own time plus 8 clocks, chained or not)
            mm(Fi{}, F{}, Ag{}, w[i][1], bq[i & 1][0]);
            between(ic, std::integral_constant<int, 0>{});
            mm(F{}, F{}, Ag{}, w[i][0], bq[i & 1][1]);
            between(ic, std::integral_constant<int, 1>{});
            mm(F{}, L{}, Ag{}, w[i][0], bq[i & 1][0]);
        } else {
            if constexpr (i == 0) {
                mm(Tr{}, F{}, Ag{}, w[i][0], bq[i & 1][0]);
            } else {
                mm(F{}, L{}, Ag{}, w[i][0], bq[i & 1][0]);
            }
            between(ic, std::integral_constant<int, 0>{});
            between(ic, std::integral_constant<int, 1>{});
        }
        if constexpr (i + 2 < NKS) bloadp<P>(bq[i & 1], brow + (i + 2) * KW);
        between(ic, std::integral_constant<int, 2>{});
    });
}

template <typename P, int YP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mcem_resident16_kernel(const MhArgs g) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    typedef typename P::Pack4 Pack4;
    constexpr int NP = P::NP, E = P::E;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int KW = 4 * E;                                             // depth of a k-step: four 16-byte chunks (bf16: 32, fp32: 16)
    constexpr int NK = HD / KW;                                           // k-steps of the 128-deep layers (4 / 8)
    constexpr int NTW = 8;                                                // output tiles (16 bins) per wave
    constexpr int NAGL = F32 ? 4 : (NP == 2 ? 2 : NK);                    // k-steps of the last tile whose fragments live in AGPRs (<= 240 in all)
    constexpr int S_LDH = SL<T>::ldh, S_LDZ = SL<T>::ldz, S_LDY = SL<T>::ldy;
    constexpr int OB4 = HD, OB5 = 2 * HD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const Ha = reinterpret_cast<T*>(smem);
    T* const Hb = Ha + T16 * S_LDH;
    T* const Zb = Hb + T16 * S_LDH;
    f32x4_t* const X2s = reinterpret_cast<f32x4_t*>(smem + S_O_X2);      // [row tile 0..31][lane]: bins 16 t + 4 q + 0..3 of frame f
    f32x4_t* const Vbs = reinterpret_cast<f32x4_t*>(smem + S_O_VB);
    float* const c1s = reinterpret_cast<float*>(smem + S_O_C1);           // [frame][S_LDC]: b3 + W3[:, 16:] y
    float* const Bias = reinterpret_cast<float*>(smem + S_O_BIAS);
    float* const w512s = reinterpret_cast<float*>(smem + S_O_W512);
    float* const p512 = reinterpret_cast<float*>(smem + S_O_P512);        // [wave][frame]
    double* const red = reinterpret_cast<double*>(smem + S_O_RED);        // [wave][frame]
    int* const accf = reinterpret_cast<int*>(smem + S_O_ACC);
    f32x4_t* const Vcs = reinterpret_cast<f32x4_t*>(smem + S_O_VC);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int f16 = lane & 15, q = lane >> 4;
    const int fb = 32 * wave_u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
    auto ld16 = [&](unsigned byteoff) __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)byteoff, 0, 0);
        return __builtin_bit_cast(Frag, v);
    };
    // the lane's chunk of fragment (16-row tile rt16, k octet o) of a copy with nt 32-row tiles per k-step of 16
    auto chunk_off = [&](int64_t base_elems, int nt, int rt16, int o) __attribute__((always_inline)) {
        const int row = 16 * rt16 + f16;
        return (unsigned)(base_elems * (int64_t)sizeof(T)) + (unsigned)((((o >> 1) * nt + (row >> 5)) * 64 + (o & 1) * 32 + (row & 31)) * 16);   // chunk o: E elements
    };
    const Frag zfrag = __builtin_bit_cast(Frag, u32x4{0u, 0u, 0u, 0u});

    // ---- resident weight fragments (once per launch) ----
    Frag w3zR[2][NP], w4R[2][NK][NP], w5R[NTW][NK][NP];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {                                       // the 16 latent columns of decoder layer 1: bf16: k octets 0, 1 (octets 2, 3 zero); fp32: quads 0 .. 3
        const unsigned o3 = chunk_off(g.oW3, 4, 2 * wave_u + rt, F32 ? q : (q & 1));
        w3zR[rt][0] = ld16(o3);
        if constexpr (NP == 2) w3zR[rt][1] = ld16(o3 + g.wpl);
        if constexpr (!F32) { if (q >= 2) { w3zR[rt][0] = zfrag; if constexpr (NP == 2) w3zR[rt][1] = zfrag; } }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const unsigned o4 = chunk_off(g.oW4, 4, 2 * wave_u + rt, 4 * ks + q);
            w4R[rt][ks][0] = ld16(o4);
            if constexpr (NP == 2) w4R[rt][ks][1] = ld16(o4 + g.wpl);
        }
    }
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const unsigned o5 = chunk_off(g.oW5, NT_OUT, NTW * wave_u + tt, 4 * ks + q);
            w5R[tt][ks][0] = ld16(o5);
            if constexpr (NP == 2) w5R[tt][ks][1] = ld16(o5 + g.wpl);
        }
    }
    for (int i = tid; i < 2 * HD + NO; i += 256) Bias[i] = g.bias[i];
    // element (row, column k) of a copy: k-step k / (2 E), lane' = (k % (2 E)) / E * 32 + row % 32, element k % E
    const T* const wc = reinterpret_cast<const T*>(g.wcopy);
    auto welem = [&](int64_t base, int nt, int row, int k) __attribute__((always_inline)) {
        const int64_t e = base + ((int64_t)((k / (2 * E)) * nt + (row >> 5)) * 64 + ((k % (2 * E)) / E) * 32 + (row & 31)) * E + (k % E);
        float v = (float)wc[e];
        if constexpr (NP == 2) v += (float)*reinterpret_cast<const T*>(reinterpret_cast<const char*>(wc + e) + g.wpl);
        return v;
    };
    if (tid < HD) w512s[tid] = welem(g.oW5, NT_OUT, 512, tid);            // row 512 of the output layer
    __syncthreads();
    const float b512 = Bias[OB5 + 512];
    const T* const Zbr = Zb + f16 * S_LDZ + q * E;
    const T* const Har = Ha + f16 * S_LDH + q * E;
    const T* const Hbr = Hb + f16 * S_LDH + q * E;

    // (one tile per workgroup -- the launcher's grid is the tile count: nothing of the chain has to stay live for a next tile)
    if (const int tile = blockIdx.x; tile < g.ntiles) {
        const int64_t n0 = (int64_t)tile * T16;
        const bool live = n0 + f16 < g.N;
        const int64_t nf = live ? n0 + f16 : g.N - 1;                      // clamped frame index of this lane
        const float g_n = g.g ? g.g[nf] : 1.f;

        // ---- per tile: label part of decoder layer 1 (fp32, constant along the chain), X2 / Vb -> LDS ----
        if constexpr (YP == NO) {
            // 513 label rows: one GEMM per tile on the label block of W3 (17 k-steps of 32 / 34 of 16, gathered from the copy), the tile's labels
            // as a [frame][544] operand image in the area X2 / Vb take afterwards
            typedef typename YPolS<P>::type PY;
            T* const Yb = reinterpret_cast<T*>(smem + S_O_X2);
            for (int idx = tid; idx < T16 * NO; idx += 256) {
                const int f = idx >> 4, fr = idx & 15;                     // consecutive threads: consecutive frames of one label row
                float yv = 0.f;
                if (f < g.ydim && n0 + fr < g.N) yv = g.y[(int64_t)f * g.N + n0 + fr];
                const T hi = P::cvt(yv);
                Yb[fr * S_LDY + f] = hi;
                if constexpr (NP == 2) Yb[Pl<PY>::lds + fr * S_LDY + f] = P::cvt(yv - (float)hi);
            }
            __syncthreads();
            f32x4_t cacc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
            const T* const Ybr = Yb + f16 * S_LDY + q * E;
#pragma unroll 1
            for (int ks = 0; ks < NO / KW; ++ks) {
                const int o = 4 * ks + q;                                  // 16-byte chunk within the label block; the block starts behind the 16 latent columns
                const bool in = o < XP / E;                                // the copy holds 528 label columns
                Frag a[2][NP], b[NP];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const unsigned oy = chunk_off(g.oW3, 4, 2 * wave_u + rt, in ? ZD / E + o : 0);
                    a[rt][0] = ld16(oy);
                    if constexpr (NP == 2) a[rt][1] = ld16(oy + g.wpl);
                    if (!in) { a[rt][0] = zfrag; if constexpr (NP == 2) a[rt][1] = zfrag; }
                }
                bloadp<PY>(b, Ybr + ks * KW);
                mm16<P>(cacc[0], a[0], b);
                mm16<P>(cacc[1], a[1], b);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4_t b3v = *reinterpret_cast<const f32x4_t*>(Bias + fb + 16 * rt + 4 * q);
                *reinterpret_cast<f32x4_t*>(c1s + f16 * S_LDC + fb + 16 * rt + 4 * q) = cacc[rt] + b3v;
            }
            __syncthreads();                                               // the label image is consumed: its area is X2 / Vb from here on
        } else {
            const int f = tid & (HD - 1), fg = tid >> 7;                   // feature, group of 8 frames
            float wy[16];
            if constexpr (YP > 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) wy[j] = welem(g.oW3, 4, f, ZD + j);          // W3[f][16 + j]
            }
            const float b3 = Bias[f];
            for (int fr = 8 * fg; fr < 8 * fg + 8; ++fr) {
                float c = b3;
                if constexpr (YP > 0) {
                    const bool in = n0 + fr < g.N;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float yv = (j < g.ydim && in) ? g.y[(int64_t)j * g.N + n0 + fr] : 0.f;
                        c = fmaf(wy[j], yv, c);
                    }
                }
                c1s[fr * S_LDC + f] = c;
            }
        }
        // (F, N) matrices through buffer descriptors: bin 16 t + 4 q + j of frame nf = ONE per-lane byte offset (bin 4 q) + a wave-uniform offset
        const int fn_bytes = (int)((int64_t)XD * g.N * 4);                 // < 2^31: checked by the launcher
        const int voff = (int)(((int64_t)(4 * q) * g.N + nf) * 4);
        const unsigned rowb = (unsigned)g.N * 4u;
        auto soff = [&](int t, int j) __attribute__((always_inline)) { return (int)((unsigned)(16 * t + j) * rowb); };
        float x2_512 = 0.f, vb_512 = 0.f;
        if (g.X2) {
            const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X2), 0, fn_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_vb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Vb), 0, fn_bytes, 0x00020000);
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) {
                const int t = NTW * wave_u + tt;
                f32x4_t xv, vv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x2, voff, soff(t, j), 0));
                    vv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_vb, voff, soff(t, j), 0));
                }
                X2s[t * 64 + lane] = xv;
                Vbs[t * 64 + lane] = vv;
            }
            if (wave_u == 3) { x2_512 = g.X2[(int64_t)512 * g.N + nf]; vb_512 = g.Vb[(int64_t)512 * g.N + nf]; }
        }

        float z[4], zp[4];
        float prior_cur = 0.f;
        double ll_cur = 0.0;
        const int zoff = (int)(((int64_t)(4 * q) * g.N + nf) * 4);         // latent 4 q + j of frame nf in a (16, N) matrix: + j rows
        if (wave_u == 0 && g.nit > 0) {
            const __amdgpu_buffer_rsrc_t rs_z0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Z0), 0, (int)((int64_t)ZD * g.N * 4), 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z0, zoff, (int)((unsigned)j * rowb), 0));
        }
        if (!F32 && wave_u == 0) {                                           // bf16: the zero half of the latent operand rows (k 16 .. 31), once per tile
            Pack4 zz;
            zz[0] = P::cvt(0.f); zz[1] = zz[0]; zz[2] = zz[0]; zz[3] = zz[0];
            *reinterpret_cast<Pack4*>(Zb + f16 * S_LDZ + 16 + 4 * q) = zz;
            if constexpr (NP == 2) *reinterpret_cast<Pack4*>(Zb + Pl<P>::lds + f16 * S_LDZ + 16 + 4 * q) = zz;
        }

        unsigned long long tlast = 0ull, tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto stamp = [&](int k) __attribute__((always_inline)) {
            if (g.dbg) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tsum[k] += t - tlast; tlast = t; }
        };
        // the lane's four values of one 16-feature row tile -> the next layer's operand rows ([frame][feature], hi and lo planes)
        auto put4 = [&](const f32x4_t& v, T* dst) __attribute__((always_inline)) {
            Pack4 p;
            p[0] = P::cvt(v[0]); p[1] = P::cvt(v[1]); p[2] = P::cvt(v[2]); p[3] = P::cvt(v[3]);
            *reinterpret_cast<Pack4*>(dst) = p;
            if constexpr (NP == 2) {
                Pack4 r;
                r[0] = P::cvt(v[0] - (float)p[0]); r[1] = P::cvt(v[1] - (float)p[1]); r[2] = P::cvt(v[2] - (float)p[2]); r[3] = P::cvt(v[3] - (float)p[3]);
                *reinterpret_cast<Pack4*>(dst + Pl<P>::lds) = r;
            }
        };
        // one decoder pass over the latents in Zb: EPI_BEGIN(tt) / EPI(tt, tile t, bin r, part, pre-activation incl. bias, x2, vb) / EPI_END(tt) for this
        // wave's output tiles -- a bin's work comes in three parts, one behind each MFMA of the k-step it shares -- EPI512(pre-activation) on
        // wave 3; ends BEHIND the output layer (no trailing barrier)
        // (PRE runs behind the requests of the latents and in front of the first MFMA: work that needs no MFMA result fills the LDS round trip)
        auto pass = [&](auto&& pre, auto&& epi_begin, auto&& epi, auto&& epi_end, auto&& epi512) __attribute__((always_inline)) {
            f32x4_t a2[2], v2[2];
            // layer 1: [z | 0] -> h1
            {
                Frag b[NP];
                bloadp<P>(b, Zbr);
                pre();
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    a2[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    mm16<P>(a2[rt], w3zR[rt], b);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4_t c = *reinterpret_cast<const f32x4_t*>(c1s + f16 * S_LDC + fb + 16 * rt + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) v2[rt][j] = P::tanh_(a2[rt][j] + c[j]);
                put4(v2[rt], Ha + f16 * S_LDH + fb + 16 * rt + 4 * q);
            }
            stamp(2);
            __syncthreads();                                               // B1
            stamp(3);
            // layer 2: h1 -> h2, and this wave's 32 terms of bin 512's pre-activation
            {
                Frag b[2][NP];
                bloadp<P>(b[0], Har);
                bloadp<P>(b[1], Har + KW);
                a2[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                a2[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    mm16<P>(a2[0], w4R[0][ks], b[ks & 1]);
                    mm16<P>(a2[1], w4R[1][ks], b[ks & 1]);
                    if (ks + 2 < NK) bloadp<P>(b[ks & 1], Har + (ks + 2) * KW);
                }
            }
            {
                float p = 0.f;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const f32x4_t c = *reinterpret_cast<const f32x4_t*>(Bias + OB4 + fb + 16 * rt + 4 * q);
                    const f32x4_t w = *reinterpret_cast<const f32x4_t*>(w512s + fb + 16 * rt + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v2[rt][j] = P::tanh_(a2[rt][j] + c[j]); p = fmaf(v2[rt][j], w[j], p); }
                    put4(v2[rt], Hb + f16 * S_LDH + fb + 16 * rt + 4 * q);
                }
                p = xsum16(p);
                p = xsum32(p);
                if (q == 0) p512[wave_u * T16 + f16] = p;
            }
            stamp(4);
            __syncthreads();                                               // B2
            stamp(5);
            // output layer: eight resident 16-row tiles per wave; the likelihood terms of tile tt (one bin per k-step) run between the MFMAs of
            // tile tt + 1; X2 / Vb of a tile are read from LDS one tile ahead
            f32x4_t acc, accn, xq[2], vq[2];
            auto bias_into = [&](f32x4_t& dst, int t) __attribute__((always_inline)) {
                dst = *reinterpret_cast<const f32x4_t*>(Bias + OB5 + 16 * t + 4 * q);
            };
            bias_into(acc, NTW * wave_u);
            xq[0] = X2s[(NTW * wave_u) * 64 + lane];
            vq[0] = Vbs[(NTW * wave_u) * 64 + lane];
            gemm16_agpr<P, NK>(acc, w5R[0], Hbr, [](auto, auto) {});
            sfor16<0, NTW>([&](auto tc) {
                constexpr int tt = decltype(tc)::value;
                const int t = NTW * wave_u + tt;
                f32x4_t& cur = (tt & 1) ? accn : acc;
                f32x4_t& nxt = (tt & 1) ? acc : accn;
                epi_begin(tc);
                if constexpr (tt + 1 < NTW) {
                    bias_into(nxt, t + 1);
                    gemm16_agpr<P, (tt + 1 < NTW - 1 ? NK : NAGL)>(nxt, w5R[tt + 1], Hbr, [&](auto ic, auto jc) {
                        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
                        // bf16: bin i of the tile behind k-step i, its three parts in the three slots; fp32 (eight k-steps): bin i / 2, parts
                        // 0 and 1 behind the even k-step, part 2 behind the odd one
                        if constexpr (!F32) epi(tc, t, ic, jc, cur[i], xq[tt & 1][i], vq[tt & 1][i]);
                        else if constexpr ((i & 1) == 0 && j < 2) epi(tc, t, std::integral_constant<int, i / 2>{}, jc, cur[i / 2], xq[tt & 1][i / 2], vq[tt & 1][i / 2]);
                        else if constexpr ((i & 1) == 1 && j == 0) epi(tc, t, std::integral_constant<int, i / 2>{}, std::integral_constant<int, 2>{}, cur[i / 2], xq[tt & 1][i / 2], vq[tt & 1][i / 2]);
                        if constexpr (j == 2 && i == 0) {
                            xq[(tt + 1) & 1] = X2s[(t + 1) * 64 + lane];
                            vq[(tt + 1) & 1] = Vbs[(t + 1) * 64 + lane];
                        }
                    });
                } else {
                    sfor16<0, 4>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        sfor16<0, 3>([&](auto jc) { epi(tc, t, rc, jc, cur[r], xq[tt & 1][r], vq[tt & 1][r]); });
                    });
                }
                epi_end(tc);
            });
            if (wave_u == 3) {
                const float a = b512 + ((p512[f16] + p512[T16 + f16]) + (p512[2 * T16 + f16] + p512[3 * T16 + f16]));
                epi512(a);
            }
            stamp(6);
        };

        // The decoder variances of the kept samples (compute_Vs, mcem.py:280-290) are those of the chain's STATE at the kept steps, and a state's
        // variances were computed by the pass that proposed it: a lane keeps exp(pre-activation) of its 32 bins of the last proposal in
        // registers, moves them to the state's copy in LDS when wave 0 has accepted the proposal (flag through LDS, read behind the next
        // barrier) and writes that copy out at the kept steps -- the bits a decoder pass over the stored sample returns (tested), without
        // the ten extra passes per chain that were a fifth of its time.
        const bool want_vs = g.Vs != nullptr && g.nit > 0;
        float vprop[NTW][4];
        float vprop512 = 0.f, vcur512 = 0.f;
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) vprop[tt][j] = 0.f;
        auto settle = [&](int md) __attribute__((always_inline)) {         // md: the step whose decision `accf` holds (-1: the evaluation of the initial state)
            if (accf[f16] != 0) {
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) Vcs[(NTW * wave_u + tt) * 64 + lane] = f32x4_t{vprop[tt][0], vprop[tt][1], vprop[tt][2], vprop[tt][3]};
                vcur512 = vprop512;
            }
            if (md >= g.burnin && live) {
                float* const vs_r = g.Vs + (int64_t)(md - g.burnin) * XD * g.N;
                const __amdgpu_buffer_rsrc_t rs_vs = __builtin_amdgcn_make_buffer_rsrc(vs_r, 0, fn_bytes, 0x00020000);
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) {
                    const f32x4_t v = Vcs[(NTW * wave_u + tt) * 64 + lane];
                    const float vj[4] = {v[0], v[1], v[2], v[3]};          // (indexing the vector with the loop variable made hipcc store element 0 four times)
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vj[j]), rs_vs, voff, soff(NTW * wave_u + tt, j), 0);
                }
                if (wave_u == 3 && q == 0) vs_r[(int64_t)512 * g.N + nf] = vcur512;
            }
        };
        const int mstart = g.nit > 0 ? -1 : 0;
        const int mend = g.nit > 0 ? g.nit : 0;
        // the draws of chain step m + 1 are requested while step m runs (wave 0)
        float nzv[4], lu = 0.f;
        const __amdgpu_buffer_rsrc_t rs_nz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.noise), 0, (int)((int64_t)g.nit * ZD * g.N * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.logu), 0, (int)((int64_t)g.nit * g.N * 4), 0x00020000);
        auto load_draws = [&](int m) __attribute__((always_inline)) {
            if (m < g.nit) {
                // (wave-uniform offsets pinned to scalar registers: hipcc kept m * rowb as a VECTOR induction variable, read it back lane by lane
                // in a waterfall loop around the logu load, and waited for that load on the spot -- a memory round trip, 1 100 of wave 0's
                // 1 700 serial clocks per chain step)
                const unsigned mb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)m * (unsigned)ZD * rowb));
                const int mlu = __builtin_amdgcn_readfirstlane((int)((unsigned)m * rowb));
#pragma unroll
                for (int j = 0; j < 4; ++j) nzv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_nz, zoff, (int)(mb + (unsigned)j * rowb), 0));
                lu = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_lu, (int)(nf * 4), mlu, 0));
            }
        };
        if (wave_u == 0 && g.nit > 0) load_draws(0);
        // the kept sample and the trace of step m leave the CU behind step m + 1's proposal (see the 32-frame kernel)
        int pend_m = -1; float pend_prob = 0.f; bool pend_acc = false;
        auto flush_step = [&]() __attribute__((always_inline)) {
            if (pend_m >= 0) {
                if (live && q == 0) {
                    if (g.accp) g.accp[(int64_t)pend_m * g.N + nf] = pend_prob;
                    if (g.accd) g.accd[(int64_t)pend_m * g.N + nf] = pend_acc ? 1 : 0;
                }
                if (pend_m >= g.burnin && live) {                                                    // mcem.py:271-273
                    float* dst = g.Zs + ((int64_t)nf * g.R + (pend_m - g.burnin)) * ZD;
                    *reinterpret_cast<f32x4_t*>(dst + 4 * q) = f32x4_t{z[0], z[1], z[2], z[3]};
                }
                pend_m = -1;
            }
        };
        if (g.dbg) tlast = __builtin_amdgcn_s_memtime();
        for (int m = mstart; m < mend; ++m) {
            float prior_p = 0.f, lu_cur = 0.f;
            if (wave_u == 0) {
                if (m >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) zp[j] = z[j] + g.sd * nzv[j];                     // mcem.py:244
                    // a copy hipcc cannot sink below the request of the next step's draws: with a plain assignment the old value stayed live
                    // across that load, the new one landed in a temporary, and the loop-carried copy waited for it on the spot
                    asm volatile("v_mov_b32 %0, %1" : "=v"(lu_cur) : "v"(lu));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) zp[j] = z[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) prior_p += zp[j] * zp[j];
                put4(f32x4_t{zp[0], zp[1], zp[2], zp[3]}, Zb + f16 * S_LDZ + 4 * q);
                prior_p = xsum16(prior_p);
                prior_p = xsum32(prior_p);
                __builtin_amdgcn_sched_barrier(0);
                if (m >= 0) load_draws(m + 1);
            }
            stamp(0);
            __syncthreads();                                               // B0
            stamp(1);
            double ll = 0.0;
            float slog = 0.f, sdiv = 0.f, vx = 0.f;                        // sums of log2(vx) and x2 / vx over one tile
            pass(
                [&]() { if (want_vs && m > mstart) settle(m - 1); },      // (its flag travels beside the latents)
                [&](auto) { slog = 0.f; sdiv = 0.f; },
                [&](auto tc, int, auto rc, auto part, float a, float x2, float vb) {
                    constexpr int pt = decltype(part)::value;
                    if constexpr (pt == 0) {
                        const float ea = P::exp_(a);
                        vprop[decltype(tc)::value][decltype(rc)::value] = ea;
                        vx = fmaf(g_n, ea, vb);                                                    // mcem.py:248-249
                    } else if constexpr (pt == 1) slog += __builtin_amdgcn_logf(vx);              // mcem.py:252-253: log(vx) + x2 / vx
                    else sdiv = fmaf(x2, __builtin_amdgcn_rcpf(vx), sdiv);
                },
                [&](auto) { ll += (double)fmaf(slog, 0.693147180559945309f, sdiv); },
                [&](float a) {
                    const float ea = P::exp_(a);
                    vprop512 = ea;
                    const float vx = fmaf(g_n, ea, vb_512);
                    const float term = P::log_(vx) + P::div_(x2_512, vx);
                    if (q == 0) ll += (double)term;
                });
            ll = xsum16(ll);
            ll = xsum32(ll);
            if (q == 0) red[wave_u * T16 + f16] = ll;
            if (wave_u == 0) flush_step();                                 // (the last step's kept sample and trace: in front of B3, where wave 0 waits for wave 3's bin 512 -- not in the serial section)
            stamp(7);
            __syncthreads();                                               // B3
            stamp(8);
            if (wave_u == 0) {
                const double ll_p = red[f16] + red[T16 + f16] + red[2 * T16 + f16] + red[3 * T16 + f16];
                if (m < 0) {
                    ll_cur = ll_p; prior_cur = prior_p;
                    if (q == 0) accf[f16] = 1;
                } else {
                    const float acc_prob = (float)(ll_cur - ll_p) + 0.5f * (prior_cur - prior_p);   // mcem.py:252-254
                    const bool is_acc = lu_cur < acc_prob;                                           // mcem.py:257
                    if (is_acc) {
                        ll_cur = ll_p; prior_cur = prior_p;
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[j] = zp[j];
                    }
                    if (q == 0) accf[f16] = is_acc ? 1 : 0;
                    pend_m = m; pend_prob = acc_prob; pend_acc = is_acc;                             // stored behind the next proposal (flush_step)
                }
            }
            // red / p512 / Zb are next written behind the barriers of the following pass
        }
        if (wave_u == 0) flush_step();
        if (wave_u == 0 && g.nit > 0 && g.Zlast != nullptr && live) {          // the chain's final state (a frame's Z0 is read by these lanes only: Zlast may be Z0)
#pragma unroll
            for (int j = 0; j < 4; ++j) g.Zlast[(int64_t)(4 * q + j) * g.N + nf] = z[j];
        }
        if (g.dbg && lane == 0 && g.nit > 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + k] = tsum[k];
            g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + 9] = (unsigned long long)(mend - mstart);
        }

        if (want_vs) {                                                     // the last step's decision
            __syncthreads();
            settle(mend - 1);
        }
        // ---- decode mode (dvae_mcem_decode): Vs[r] = decoder([Zs[:, r, :] | y])  (mcem.py:280-290) ----
        if (g.Vs != nullptr && g.nit == 0) {
            for (int r_s = 0; r_s < g.R; ++r_s) {
                __syncthreads();
                if (wave_u == 0) {
                    const float* src = g.Zs + ((int64_t)nf * g.R + r_s) * ZD;
                    const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(src + 4 * q);
                    put4(s0, Zb + f16 * S_LDZ + 4 * q);
                }
                __syncthreads();
                float* const vs_r = g.Vs + (int64_t)r_s * XD * g.N;
                const __amdgpu_buffer_rsrc_t rs_vs = __builtin_amdgcn_make_buffer_rsrc(vs_r, 0, fn_bytes, 0x00020000);
                pass(
                    []() {},
                    [](auto) {},
                    [&](auto, int t, auto jc, auto part, float a, float, float) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (decltype(part)::value == 0) {
                            if (live) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, P::exp_(a)), rs_vs, voff, soff(t, j), 0);
                        }
                    },
                    [](auto) {},
                    [&](float a) { if (live && q == 0) vs_r[(int64_t)512 * g.N + nf] = P::exp_(a); });
            }
        }
        __syncthreads();
    }
}

template <typename P, int YP>
static int launch_resident16_t(const MhArgs& a, hipStream_t s) {
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)mcem_resident16_kernel<P, YP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S_LDS);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(mcem_resident16_kernel, %zu B LDS): %s", S_LDS, hipGetErrorString(e)); return (int)e; }
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((mcem_resident16_kernel<P, YP>), dim3(a.ntiles), dim3(256), S_LDS, s, a);
    DVAE_LAUNCH_OK("mcem_resident16_kernel");
    return 0;
}

bool resident16_chain_supported(int precision, int yp) {
    return (precision == DVAE_PREC_BF16X3 || precision == DVAE_PREC_BF16 || precision == DVAE_PREC_F32) && (yp == 0 || yp == 16 || yp == XP);
}

// a.ntiles: 16-frame tiles
int launch_resident16_chain(int precision, int yp, const MhArgs& a, hipStream_t s) {
    const bool x3 = precision == DVAE_PREC_BF16X3, f32 = precision == DVAE_PREC_F32;
    if (yp == 0) return x3 ? launch_resident16_t<PolX3S, 0>(a, s) : f32 ? launch_resident16_t<PolF32S, 0>(a, s) : launch_resident16_t<PolB1S, 0>(a, s);
    if (yp == 16) return x3 ? launch_resident16_t<PolX3S, 16>(a, s) : f32 ? launch_resident16_t<PolF32S, 16>(a, s) : launch_resident16_t<PolB1S, 16>(a, s);
    if (yp == XP) return x3 ? launch_resident16_t<PolX3S, NO>(a, s) : f32 ? launch_resident16_t<PolF32S, NO>(a, s) : launch_resident16_t<PolB1S, NO>(a, s);
    set_error("mcem resident chain (16 frames): label rows 0, 1..16 or 513 only");
    return DVAE_E_UNSUPPORTED;
}

}  // namespace fused
}  // namespace dvae
