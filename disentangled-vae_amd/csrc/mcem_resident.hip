// Metropolis-Hastings chain of the MCEM E-step, WEIGHT-STATIONARY form (round 4): sample_posterior + compute_Vs of
// packages/models/mcem.py:207-290 for the split-bf16 policy, label rows 0 / 1..16.
//
// Why: the chain applies ONE small decoder ([16+y]-128-128-513) to the same 32 frames 41 times per E-step.  The streaming kernel
// (mcem.hip) pulls the decoder's fragment copies (336 KB with both bf16 planes) plus the tile's X2 / Vb rows (131 KB) from L2 in every
// chain step: 9.3 us per step at one workgroup per CU, bound by that stream, not by the 8 MMAC of MFMA work (1.7 us at peak).  A CU's
// register file holds 512 KB.  Here ONE 256-thread workgroup per CU -- one wave per SIMD, all 512 registers of the lane (256 VGPRs + 256
// AGPRs: MFMA reads its A operand from either) -- keeps the whole decoder in registers for the launch and the tile's X2 / Vb on chip for
// the chain:
//   * output layer (513 x 128, both planes, 262 KB): wave w owns bins 128 w .. 128 w + 127 as four 32-row tiles: 256 registers of
//     resident fragments per lane;
//   * bin 512 (the 17th row tile would hold this one row): its pre-activation is a 128-term dot product with h2; every wave adds the
//     terms of the h2 features it has just produced (fp32, in the layer-2 epilogue), one LDS exchange, wave 3 finishes it;
//   * layers 1 and 2: wave w owns features 32 w .. 32 w + 31 (64 + 8 registers of resident fragments);
//   * the label part of layer 1 (constant along the chain) is a per-tile fp32 table in LDS; the tile's X2 sits in LDS in the order the
//     epilogue reads it (wave-private, conflict-free), its Vb in 64 registers per lane.
// Nothing but the draws (noise, logu) and the kept samples crosses the CU boundary inside the chain.  Per chain step: 4 workgroup barriers
// (latents -> h1 -> h2 -> per-frame likelihood), exactly the phases of the streaming kernel and the same arithmetic per element (hardware
// exp2 / log2 / rcp, likelihood sums in double); bin 512 is summed in fp32 instead of on split planes.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "fused_tiles.hpp"
#include "mcem_types.hpp"
#include "../../include/dvae_mcem.h"

namespace dvae {
namespace fused {

struct PolX3C : PolX3 {};            // split bf16: two operand planes, three MFMAs per product
struct PolB1C : PolBF16 {};          // one bf16 per operand (opt-in fast policy)
struct PolF32C : PolF32 {            // exact fp32 products (v_mfma_f32_32x32x2_f32); epilogue on the hardware exp2 / log2 / rcp units as the
                                     // streaming fp32 chain (mcem.hip: PolF32Deep)
    static __device__ __forceinline__ float exp_(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }
    static __device__ __forceinline__ float log_(float v) { return __builtin_amdgcn_logf(v) * 0.693147180559945309f; }
    static __device__ __forceinline__ float tanh_(float v) {
        const float e = __builtin_amdgcn_exp2f(v * 2.88539008177792681f);
        return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    }
    static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
};
constexpr int C8_LDC = HD + 4;
constexpr int C8_PLANE = TB * (2 * (HD + 8) + 32 + 8);                    // bf16 elements of one operand plane: h1, h2, latents
template <> struct Pl<PolX3C> { static constexpr int lds = C8_PLANE; };
template <> struct Pl<PolB1C> { static constexpr int lds = 0; };
template <> struct Pl<PolF32C> { static constexpr int lds = 0; };
// the 513-row label image of a tile ([frame][528 + pad], once per tile, aliased over the X2 / Vb area): same policies, its own plane stride
struct PolX3Y : PolX3C {};
struct PolB1Y : PolB1C {};
struct PolF32Y : PolF32C {};
constexpr int C8_YP = 528;
template <> struct Pl<PolX3Y> { static constexpr int lds = TB * (C8_YP + 8); };
template <> struct Pl<PolB1Y> { static constexpr int lds = 0; };
template <> struct Pl<PolF32Y> { static constexpr int lds = 0; };
template <typename P> struct YPol;
template <> struct YPol<PolX3C> { typedef PolX3Y type; };
template <> struct YPol<PolB1C> { typedef PolB1Y type; };
template <> struct YPol<PolF32C> { typedef PolF32Y type; };
static_assert((size_t)TB * (C8_YP + 8) * 2 * sizeof(__bf16) <= (size_t)(16 + 8) * 16 * 64 * sizeof(float) && (size_t)TB * (C8_YP + 4) * sizeof(float) <= (size_t)(16 + 8) * 16 * 64 * sizeof(float), "the label image fits the X2 / Vb area");
static_assert((size_t)TB * (2 * (HD + 4) + 32 + 4) * sizeof(float) <= (size_t)C8_PLANE * 2 * sizeof(__bf16), "the fp32 activation plane fits the two bf16 planes");

constexpr size_t C8_O_X2 = (size_t)C8_PLANE * 2 * sizeof(__bf16);
constexpr size_t C8_O_VB = C8_O_X2 + (size_t)16 * 16 * 64 * sizeof(float);        // Vb of each wave's output tiles 2 and 3 (tiles 0, 1: registers)
constexpr size_t C8_O_C1 = C8_O_VB + (size_t)8 * 16 * 64 * sizeof(float);
constexpr size_t C8_O_BIAS = C8_O_C1 + (size_t)TB * C8_LDC * sizeof(float);
constexpr size_t C8_O_W512 = C8_O_BIAS + (size_t)(2 * HD + NO) * sizeof(float);
constexpr size_t C8_O_P512 = C8_O_W512 + (size_t)HD * sizeof(float);
constexpr size_t C8_O_RED = C8_O_P512 + (size_t)4 * TB * sizeof(float);
constexpr size_t C8_O_REJ = C8_O_RED + (size_t)4 * TB * sizeof(double);       // [frame]: bit r = kept step r rejected its proposal
constexpr size_t C8_O_ZSV = C8_O_REJ + (size_t)TB * sizeof(unsigned long long);  // [lane of wave 0][8]: the state at the end of the burn-in
constexpr size_t C8_LDS = C8_O_ZSV + (size_t)64 * 8 * sizeof(float);
static_assert(C8_O_RED % 8 == 0 && C8_O_ZSV % 16 == 0 && C8_LDS <= 160 * 1024, "resident chain: LDS layout");

// acc = W(resident fragments, hi / lo planes) * act^T over NSTEPS k-steps of 16, the weight operand read straight from ACCUMULATION registers.
// hipcc never assigns an AGPR to an MFMA source by itself (it parks the fragments there and copies them to VGPRs in front of every MFMA: 340
// v_accvgpr_read per chain step), so the MFMAs of the output layer are written out: "a" pins a fragment to the AGPR file for its whole life
// (the buffer load that defines it writes the AGPR directly).  The compiler does not see into the statement: the chain starts from the
// inline constant 0 (no VALU-written SrcC), every MFMA accumulates into the one before it (same opcode, back to back), and the s_nops behind
// the last one cover the longest XDL-write -> VALU-read distance (19 wait states) before anything outside may touch `acc`.
//
// Hazard table of the hand-written chain (nothing inside an asm statement is padded by hipcc; outside it adds ONE state after ;;#ASMEND
// before a VALU that touches the outputs).  Producer -> consumer, the wait states the ISA asks for, and what provides them here:
//
//   | producer                                   | consumer                                        | required            | provided by                                   |
//   |--------------------------------------------|-------------------------------------------------|---------------------|-----------------------------------------------|
//   | v_mfma D (XDL write of acc)                | the NEXT v_mfma of the chain taking acc whole   | 0 (same opcode,     | back-to-back statements, "+v"(acc): the       |
//   |                                            | as SrcC (accumulate chain)                      | same size SrcC = D) | hardware interlocks the dependent issue       |
//   | last v_mfma of a tile: 32x32x16 bf16 /     | any VALU / store / LDS write reading acc, any   | 8-pass: 12 states   | s_nop 7 + s_nop 7 + s_nop 3 = 20 states       |
//   | f16 (8 passes), 32x32x2 f32 (16 passes)    | non-MFMA writer of acc (compiler code after the | 16-pass: 18 states  | INSIDE the last statement (LAST = true) + the |
//   |                                            | statement)                                      |                     | compiler's one state after ;;#ASMEND          |
//   | buffer_load defining an "a" (AGPR) fragment| v_mfma reading it as SrcA                       | s_waitcnt vmcnt     | the loads are compiler-visible builtins: hipcc|
//   |                                            |                                                 | (compiler-counted)  | waits before the first asm statement reading  |
//   | VALU / ds_read writing a "v" B operand (bq)| v_mfma reading it as SrcB                       | VALU write: 2 states| bq comes from ds_read (lgkmcnt, compiler-     |
//   |                                            |                                                 | (Table 38); ds_read:| counted) one k-step ahead; never VALU-written |
//   |                                            |                                                 | lgkmcnt             | between the read and the MFMA                 |
//   | v_accvgpr_write (hipcc parks spilled VGPRs | v_mfma reading an AGPR as SrcA                  | 2 states, and never | hipcc pads its OWN v_accvgpr_write -> MFMA    |
//   | in the 16 AGPRs the fragments leave free:  |                                                 | into a register an  | pairs only for MFMAs it emitted; the asm MFMAs|
//   | 80 writes / 656 reads over the 9 kernels,  |                                                 | in-flight MFMA reads| name their AGPR through "a": the allocator    |
//   | ROCm 7.2)                                  |                                                 |                     | keeps spill slots and fragments disjoint as   |
//   |                                            |                                                 |                     | long as 16 AGPRs stay unassigned (a full file |
//   |                                            |                                                 |                     | made it rotate fragments through temporaries: |
//   |                                            |                                                 |                     | the round-4 hazard, see the kernel's header)  |
//   | VALU write of acc (bias preload)           | first v_mfma reading acc as SrcC                | 2 states            | acc is preloaded by ds_read (no VALU write);  |
//   |                                            |                                                 |                     | the fp32 form starts from the inline 0        |
//
// The figures for rows 2 and 4 are the ISA's (cdna4 ISA, "MFMA dependent instruction" and "VALU write -> MFMA read" tables); 20 states cover
// both MFMA sizes used, so the bf16 forms carry 8 states of slack.  What a compiler or flag change can break without a build error: (a) the
// register allocator moving a fragment (a v_accvgpr_write into a register an MFMA of the chain still reads); (b) a VALU instruction that
// writes acc scheduled between the bias ds_read and the first MFMA.  Neither shows in the build log; both show as launch-to-launch or
// policy-to-oracle differences: tests/test_gpu_mcem.py::test_chain_launches_are_bit_identical (every policy and label variant, repeated
// launches bit for bit) and the oracle / reference-golden tests are the guards, and tools/r05/audit_resident.sh prints the per-kernel
// register, scratch and v_accvgpr counts of these kernels and of the 16-frame ones to compare after a toolchain change (ROCm 7.2, end of round 5:
// profiles/r05_mcem_resident_register_audit.txt -- 0 - 140 bytes of scratch per lane here, all of it in the per-tile prologue and epilogue, none
// between the barriers of a chain step; none at all in the 16-frame kernels).
template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// between(k-step i, j): independent work placed in program order behind the j-th MFMA of the k-step -- a wave issues in order and every MFMA of
// the chain waits for the one before it (32 clocks), so whatever should overlap with the matrix pipe has to sit BETWEEN the MFMAs (here: the
// likelihood terms of the previous tile, one bin behind each of the first two MFMAs, the next bins' LDS reads behind the third).
// `acc` comes in holding the bias (read from LDS straight into the accumulator registers): no bias add per bin, no VALU-written SrcC.
template <typename P, int NSTEPS, int NAG, typename Between>
__device__ __forceinline__ void gemm_resident_agpr(f32x16& acc, const typename P::Frag (&w)[NSTEPS][P::NP], const typename P::T* brow, Between&& between) {
    typedef typename P::Frag Frag;
    constexpr int STR = 2 * P::E;
    constexpr int BD = NSTEPS < 2 ? NSTEPS : 2;
    Frag bq[BD][P::NP];
#pragma unroll
    for (int i = 0; i < BD; ++i) bloadp<P>(bq[i], brow + i * STR);
    // one MFMA of the chain; LAST carries the s_nops; AG: the weight fragment lives in an AGPR
    auto mm = [&](auto last, auto ag, const Frag& a, const Frag& b) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last)::value, AG = decltype(ag)::value;
        if constexpr (LAST) {
            if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
        } else {
            if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
    };
    // fp32: one v_mfma_f32_32x32x2_f32 per fragment element (k = j and 4 + j of the k-step)
    auto mf = [&](auto last, auto ag, float a, float b) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last)::value, AG = decltype(ag)::value;
        if constexpr (LAST) {
            if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
        } else {
            if constexpr (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
    };
    sfor<0, NSTEPS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        typedef std::integral_constant<bool, (i < NAG)> Ag;
        typedef std::integral_constant<bool, false> F;
        typedef std::integral_constant<bool, i + 1 == NSTEPS> L;
        if constexpr (sizeof(typename P::T) == 4) {
            mf(F{}, Ag{}, w[i][0][0], bq[i % BD][0][0]);
            mf(F{}, Ag{}, w[i][0][1], bq[i % BD][0][1]);
            between(ic, std::integral_constant<int, 0>{});
            mf(F{}, Ag{}, w[i][0][2], bq[i % BD][0][2]);
            between(ic, std::integral_constant<int, 1>{});
            mf(L{}, Ag{}, w[i][0][3], bq[i % BD][0][3]);
        } else if constexpr (P::NP == 2) {
            mm(F{}, Ag{}, w[i][1], bq[i % BD][0]);
            between(ic, std::integral_constant<int, 0>{});
            mm(F{}, Ag{}, w[i][0], bq[i % BD][1]);
            between(ic, std::integral_constant<int, 1>{});
            mm(L{}, Ag{}, w[i][0], bq[i % BD][0]);
        } else {
            mm(L{}, Ag{}, w[i][0], bq[i % BD][0]);
            between(ic, std::integral_constant<int, 0>{});
            between(ic, std::integral_constant<int, 1>{});
        }
        if constexpr (i + BD < NSTEPS) bloadp<P>(bq[i % BD], brow + (i + BD) * STR);
        between(ic, std::integral_constant<int, 2>{});
    });
}

template <typename P, int YP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mcem_resident_kernel(const MhArgs g) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    constexpr int NP = P::NP, E = P::E, KS = P::KSTEP, NK = HD / KS;     // bf16: 8 k-steps of 16; fp32: 16 k-steps of 8
    constexpr int BPK = 16 / NK;                                          // bins of the previous tile handled per k-step: 2 / 1
    constexpr int NAG3 = sizeof(T) == 4 ? 12 : (NP == 2 ? 6 : NK);        // k-steps of tile 3 whose fragments live in AGPRs (<= 240 AGPRs in all)
    constexpr int LDH = HD + 16 / (int)sizeof(T), LDZ = 32 + 16 / (int)sizeof(T), LDC = C8_LDC;
    constexpr int OB4 = HD, OB5 = 2 * HD;
    constexpr int NTW = 4;                                                 // output tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const Ha = reinterpret_cast<T*>(smem);
    T* const Hb = Ha + TB * LDH;
    T* const Zb = Hb + TB * LDH;
    float* const X2s = reinterpret_cast<float*>(smem + C8_O_X2);          // [out tile 0..15][r 0..15][lane]: this lane's bins of its wave's tiles
    float* const Vbs = reinterpret_cast<float*>(smem + C8_O_VB);          // [wave][tile 2..3][r][lane]
    float* const c1s = reinterpret_cast<float*>(smem + C8_O_C1);          // [frame][LDC]: b3 + W3[:, 16:] y
    float* const Bias = reinterpret_cast<float*>(smem + C8_O_BIAS);
    float* const w512s = reinterpret_cast<float*>(smem + C8_O_W512);      // row 512 of the output layer, fp32 (hi + lo)
    float* const p512 = reinterpret_cast<float*>(smem + C8_O_P512);       // [wave][frame]: partial pre-activation of bin 512
    double* const red = reinterpret_cast<double*>(smem + C8_O_RED);       // [wave][frame]: partial likelihood sums
    unsigned long long* const rejs = reinterpret_cast<unsigned long long*>(smem + C8_O_REJ);
    float* const zsv = reinterpret_cast<float*>(smem + C8_O_ZSV);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int l31 = lane & 31, h = lane >> 5;                             // the lane is the frame, h the k half / feature half
    const int fb = 32 * wave_u;
    const unsigned long long t_entry = g.dbg ? __builtin_amdgcn_s_memtime() : 0ull;   // (diagnostic stamps: slots 10-12 = prologue, chain, tail in shader clocks)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
    auto ld16 = [&](unsigned byteoff) __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)byteoff, 0, 0);
        return __builtin_bit_cast(Frag, v);
    };

    // X2 of the wave's four output tiles and Vb of the last two go HBM -> LDS by LDS-direct loads (`buffer_load ... lds`: wave-uniform
    // destination + lane * 4 = the [tile][r][lane] layout), requested before anything else: through registers -- all 512 of a lane hold
    // resident fragments under bf16x3 / fp32 -- a few of the 96 loads were in flight at a time, 35 of the 58 us a launch spends outside
    // its passes (25 utterances; profiles/r05_mcem_tile32_prologue_ab.txt).  (Label images of 513 rows use the area first: requested behind them.)
    auto request_x2vb = [&](int voff_, unsigned rowb_) __attribute__((always_inline)) {
        const int fnb = (int)((int64_t)XD * g.N * 4);
        const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X2), 0, fnb, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_vb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Vb), 0, fnb, 0x00020000);
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
            const int t = NTW * wave_u + tt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = (int)((unsigned)(32 * t + (r & 3) + 8 * (r >> 2)) * rowb_);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (__attribute__((address_space(3))) void*)(X2s + (t * 16 + r) * 64), 4, voff_, so, 0, 0);
                if (tt >= 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_vb, (__attribute__((address_space(3))) void*)(Vbs + ((wave_u * 2 + tt - 2) * 16 + r) * 64), 4, voff_, so, 0, 0);
            }
        }
    };
    if constexpr (YP != C8_YP) {
        if ((int)blockIdx.x < g.ntiles && g.X2) {
            const int64_t n0_ = (int64_t)blockIdx.x * TB;
            const int64_t nf_ = n0_ + l31 < g.N ? n0_ + l31 : g.N - 1;
            request_x2vb((int)(((int64_t)(4 * h) * g.N + nf_) * 4), (unsigned)g.N * 4u);
        }
    }

    // ---- resident weight fragments (once per launch); copies: [k-step of 16][32-row tile][lane][8] ----
    Frag w3zR[ZD / KS][NP], w4R[NK][NP], w5R[NTW][NK][NP];
#pragma unroll
    for (int ks = 0; ks < ZD / KS; ++ks) {                                 // the 16 latent columns of decoder layer 1
        const unsigned o3 = (unsigned)(g.oW3 * sizeof(T)) + (unsigned)(((ks * 4 + wave_u) * 64 + lane) * 16);
        w3zR[ks][0] = ld16(o3);
        if constexpr (NP == 2) w3zR[ks][1] = ld16(o3 + g.wpl);
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const unsigned o4 = (unsigned)(g.oW4 * sizeof(T)) + (unsigned)(((ks * 4 + wave_u) * 64 + lane) * 16);
        w4R[ks][0] = ld16(o4);
        if constexpr (NP == 2) w4R[ks][1] = ld16(o4 + g.wpl);
    }
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const unsigned o5 = (unsigned)(g.oW5 * sizeof(T)) + (unsigned)(((ks * NT_OUT + NTW * wave_u + tt) * 64 + lane) * 16);
            w5R[tt][ks][0] = ld16(o5);
            if constexpr (NP == 2) w5R[tt][ks][1] = ld16(o5 + g.wpl);
        }
    }
    for (int i = tid; i < 2 * HD + NO; i += 256) Bias[i] = g.bias[i];
    // element (row, column k) of a copy: k-step k / KS, lane' = (k % KS) / E * 32 + row % 32, element k % E
    const T* const wc = reinterpret_cast<const T*>(g.wcopy);
    auto welem = [&](int64_t base, int nt, int row, int k) __attribute__((always_inline)) {
        const int64_t e = base + ((int64_t)((k / KS) * nt + (row >> 5)) * 64 + ((k % KS) / E) * 32 + (row & 31)) * E + (k % E);
        float v = (float)wc[e];
        if constexpr (NP == 2) v += (float)*reinterpret_cast<const T*>(reinterpret_cast<const char*>(wc + e) + g.wpl);
        return v;
    };
    if (tid < HD) w512s[tid] = welem(g.oW5, NT_OUT, 512, tid);           // row 512 of the output layer
    __syncthreads();
    const unsigned long long t_w = g.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    const float b512 = Bias[OB5 + 512];
    const T* const Zbr = Zb + l31 * LDZ + h * E;
    const T* const Har = Ha + l31 * LDH + h * E;
    const T* const Hbr = Hb + l31 * LDH + h * E;

    // (one tile per workgroup -- the launcher's grid is the tile count: the resident fragments are dead behind the chain, their registers
    // hold the rows the fix-up below moves)
    if (const int tile = blockIdx.x; tile < g.ntiles) {
        const int64_t n0 = (int64_t)tile * TB;
        const bool live = n0 + l31 < g.N;
        const int64_t nf = live ? n0 + l31 : g.N - 1;                      // clamped frame index of this lane
        const float g_n = g.g ? g.g[nf] : 1.f;

        // ---- per tile: label part of decoder layer 1 (fp32, constant along the chain), X2 -> LDS, Vb -> registers ----
        if constexpr (YP == C8_YP) {
            // 513 label rows: one GEMM per tile on the streamed label block of W3 (33 / 66 k-steps: mcem.hip does the same), the tile's labels
            // as a [frame][528] operand image in the area X2 / Vb take afterwards
            typedef typename YPol<P>::type PY;
            constexpr int LDY = C8_YP + 16 / (int)sizeof(T);
            T* const Yb = reinterpret_cast<T*>(smem + C8_O_X2);
            for (int idx = tid; idx < TB * C8_YP; idx += 256) {
                const int f = idx >> 5, fr = idx & 31;              // consecutive threads: consecutive frames of one label row
                float yv = 0.f;
                if (f < g.ydim && n0 + fr < g.N) yv = g.y[(int64_t)f * g.N + n0 + fr];
                const T hi = P::cvt(yv);
                Yb[fr * LDY + f] = hi;
                if constexpr (NP == 2) Yb[Pl<PY>::lds + fr * LDY + f] = P::cvt(yv - (float)hi);
            }
            __syncthreads();
            constexpr unsigned FBB = 64 * 16;                       // bytes of one fragment block (64 lanes x 16 B)
            const WRef w3y{lane * 16, (unsigned)(g.oW3 * sizeof(T)) + (unsigned)((ZD / KS) * 4 + wave_u) * FBB, g.wpl};
            f32x16 cacc;
            zero_acc<PY>(cacc);
            WPre<PY, C8_YP / KS, 2> wp;                             // (a shallow ring: this runs once per tile beside 400 resident registers)
            wprefetch<PY, C8_YP / KS, 2>(wp, wrs, w3y, 4 * FBB);
            gemm_block<PY, C8_YP / KS, NoHook, 2, 4>(cacc, wp, wrs, w3y, Yb + l31 * LDY + h * E, 4 * FBB);
            float b3v[16];
            bias16(Bias, fb, h, b3v);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<f32x4*>(c1s + l31 * LDC + fb + 8 * gq + 4 * h) =
                    f32x4{cacc[4 * gq] + b3v[4 * gq], cacc[4 * gq + 1] + b3v[4 * gq + 1], cacc[4 * gq + 2] + b3v[4 * gq + 2], cacc[4 * gq + 3] + b3v[4 * gq + 3]};
            __syncthreads();                                        // the label image is consumed: its area is X2 / Vb from here on
        } else {
            const int f = tid & (HD - 1), fg = tid >> 7;                   // feature, group of 16 frames
            float wy[16];
            if constexpr (YP > 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) wy[j] = welem(g.oW3, 4, f, ZD + j);          // W3[f][16 + j]
            }
            const float b3 = Bias[f];
            if constexpr (YP > 0) {
                // the tile's labels through LDS ([frame][16], in the area the burn-in state takes later): read one by one through wave-uniform
                // addresses they were sixteen scalar loads in a row, each waited for -- most of a launch's 18 us before its first pass
                for (int idx = tid; idx < 16 * TB; idx += 256) {
                    const int j = idx >> 5, fr = idx & 31;
                    zsv[fr * 16 + j] = (j < g.ydim && n0 + fr < g.N) ? g.y[(int64_t)j * g.N + n0 + fr] : 0.f;
                }
                __syncthreads();
            }
#pragma unroll 4
            for (int fr = 16 * fg; fr < 16 * fg + 16; ++fr) {
                float c = b3;
                if constexpr (YP > 0) {
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        const f32x4 yv = *reinterpret_cast<const f32x4*>(zsv + fr * 16 + 4 * jq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) c = fmaf(wy[4 * jq + e], yv[e], c);
                    }
                }
                c1s[fr * LDC + f] = c;
            }
        }
        // (F, N) matrices are addressed through buffer descriptors: bin 32 t + feat_of(r, h) of frame nf = ONE per-lane byte offset (bin 4 h)
        // plus a wave-uniform offset -- with 64-bit pointers hipcc keeps the 64 addresses of a lane live across the chain (128 registers)
        const int fn_bytes = (int)((int64_t)XD * g.N * 4);                 // < 2^31: checked by the launcher
        const int voff = (int)(((int64_t)(4 * h) * g.N + nf) * 4);
        const unsigned rowb = (unsigned)g.N * 4u;
        auto soff = [&](int t, int r) __attribute__((always_inline)) { return (int)((unsigned)(32 * t + (r & 3) + 8 * (r >> 2)) * rowb); };
        float vbR[2][16];
        float x2_512 = 0.f, vb_512 = 0.f;
        if (g.X2) {
            if constexpr (YP == C8_YP) request_x2vb(voff, rowb);            // (the other label forms: requested at the top of the kernel)
            const __amdgpu_buffer_rsrc_t rs_vb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Vb), 0, fn_bytes, 0x00020000);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = NTW * wave_u + tt;
#pragma unroll
                for (int r = 0; r < 16; ++r) vbR[tt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_vb, voff, soff(t, r), 0));
            }
            if (wave_u == 3) { x2_512 = g.X2[(int64_t)512 * g.N + nf]; vb_512 = g.Vb[(int64_t)512 * g.N + nf]; }
            // the LDS-direct loads have landed before the first pass reads X2s / Vbs (the wave that requested a slot is the one that reads it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) vbR[tt][r] = 0.f;
            }
        }

        float z[8], zp[8];
        float prior_cur = 0.f;
        double ll_cur = 0.0;
        if (wave_u == 0 && g.nit > 0) {
            const __amdgpu_buffer_rsrc_t rs_z0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Z0), 0, (int)((int64_t)ZD * g.N * 4), 0x00020000);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                z[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z0, voff, (int)((unsigned)((r & 3) + 8 * (r >> 2)) * rowb), 0));
        }

        // one decoder pass over the latents in Zb: EPI_BEGIN(tt) / EPI(tt, tile t, r, pre-activation incl. bias) / EPI_END(tt) for this wave's output
        // tiles (tt, r: compile-time constants), EPI512(pre-activation) on wave 3; ends BEHIND the output layer (no trailing barrier)
        unsigned long long tlast = 0ull, tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto stamp = [&](int k) __attribute__((always_inline)) {
            if (g.dbg) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tsum[k] += t - tlast; tlast = t; }
        };
        auto pass = [&](auto&& epi_begin, auto&& epi, auto&& epi_end, auto&& epi512) __attribute__((always_inline)) {
            f32x16 acc;
            float v[16], bv[16];
            // layer 1: [z | y] -> h1
            zero_acc<P>(acc);
            gemm_resident_p<P, ZD / KS>(acc, w3zR, Zbr);
            bias16(c1s + l31 * LDC, fb, h, bv);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = P::tanh_(acc[r] + bv[r]);
            put_lds<P>(v, Ha, LDH, fb, l31, h);
            stamp(2);
            __syncthreads();                                               // B1
            stamp(3);
            // layer 2: h1 -> h2, and this wave's 32 terms of bin 512's pre-activation
            zero_acc<P>(acc);
            gemm_resident_p<P, NK>(acc, w4R, Har);
            bias16(Bias + OB4, fb, h, bv);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = P::tanh_(acc[r] + bv[r]);
            put_lds<P>(v, Hb, LDH, fb, l31, h);
            {
                bias16(w512s, fb, h, bv);
                float p = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) p = fmaf(v[r], bv[r], p);
                p = xsum32(p);
                if (h == 0) p512[wave_u * TB + l31] = p;
            }
            stamp(4);
            __syncthreads();                                               // B2
            stamp(5);
            // output layer: four resident 32-row tiles per wave; the epilogue of tile tt (two bins per k-step) runs between the MFMAs of tile tt + 1.
            // 240 of the 256 AGPRs hold fragments (tiles 0-2 whole, 6 of tile 3's 8 k-steps); the rest of tile 3 sits in VGPRs
            f32x16 accn;
            auto bias_into = [&](f32x16& dst, int t) __attribute__((always_inline)) {
                bias16(Bias + OB5, 32 * t, h, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r] = bv[r];
            };
            bias_into(acc, NTW * wave_u);
            gemm_resident_agpr<P, NK, NK>(acc, w5R[0], Hbr, [](auto, auto) {});
            sfor<0, NTW>([&](auto tc) {
                constexpr int tt = decltype(tc)::value;
                const int t = NTW * wave_u + tt;
                f32x16& cur = (tt & 1) ? accn : acc;
                f32x16& nxt = (tt & 1) ? acc : accn;
                float xq[16], vq[16];                                      // X2 / Vb of the tile's bins, read one k-step ahead of their use
                auto loadxv = [&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = decltype(rc)::value;
                    xq[r] = X2s[(t * 16 + r) * 64 + lane];
                    if constexpr (tt < 2) vq[r] = vbR[tt][r]; else vq[r] = Vbs[((wave_u * 2 + tt - 2) * 16 + r) * 64 + lane];
                };
                sfor<0, BPK>([&](auto bc) { loadxv(bc); });
                epi_begin(tc);
                if constexpr (tt + 1 < NTW) {
                    bias_into(nxt, t + 1);
                    gemm_resident_agpr<P, NK, (tt + 1 < NTW - 1 ? NK : NAG3)>(nxt, w5R[tt + 1], Hbr, [&](auto ic, auto jc) {
                        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
                        if constexpr (j < BPK) epi(tc, t, std::integral_constant<int, BPK * i + j>{}, cur[BPK * i + j], xq[BPK * i + j], vq[BPK * i + j]);
                        else if constexpr (j == 2 && i + 1 < NK) sfor<0, BPK>([&](auto bc) { loadxv(std::integral_constant<int, BPK * (i + 1) + decltype(bc)::value>{}); });
                    });
                } else {
                    sfor<BPK, 16>([&](auto rc) { loadxv(rc); });
                    sfor<0, 16>([&](auto rc) { constexpr int r = decltype(rc)::value; epi(tc, t, rc, cur[r], xq[r], vq[r]); });
                }
                epi_end(tc);
            });
            if (wave_u == 3) {
                const float a = b512 + ((p512[l31] + p512[TB + l31]) + (p512[2 * TB + l31] + p512[3 * TB + l31]));
                epi512(a);
            }
            stamp(6);
        };

        const int mstart = g.nit > 0 ? -1 : 0;
        const int mend = g.nit > 0 ? g.nit : 0;
        // the draws of chain step m + 1 are requested while step m runs (wave 0): a load waited for on the spot costs 1-2 us per step
        float nzv[8], lu = 0.f;
        // (one per-lane offset + wave-uniform offsets again: eight 64-bit row addresses would be spilled and reloaded one by one, each behind
        // its own s_waitcnt vmcnt(0) -- measured 4 us per chain step)
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
                for (int r = 0; r < 8; ++r)
                    nzv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_nz, voff, (int)(mb + (unsigned)((r & 3) + 8 * (r >> 2)) * rowb), 0));
                lu = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_lu, (int)(nf * 4), mlu, 0));
            }
        };
        if (wave_u == 0 && g.nit > 0) load_draws(0);
        // The kept sample and the trace of step m leave the CU behind step m + 1's proposal: vmcnt retires in issue order, so stores issued
        // in front of the proposal made the wait for the prefetched draws a wait for the stores' acknowledgement too (the serial section of
        // wave 0 -- everyone else waits at B0 -- was 2400 of a step's 13 200 clocks, tools/stamp_mcem.py)
        int pend_m = -1; float pend_prob = 0.f; bool pend_acc = false;
        auto flush_step = [&]() __attribute__((always_inline)) {
            if (pend_m >= 0) {
                if (live && h == 0) {
                    if (g.accp) g.accp[(int64_t)pend_m * g.N + nf] = pend_prob;
                    if (g.accd) g.accd[(int64_t)pend_m * g.N + nf] = pend_acc ? 1 : 0;
                }
                if (pend_m >= g.burnin && live) {                                                    // mcem.py:271-273
                    float* dst = g.Zs + ((int64_t)nf * g.R + (pend_m - g.burnin)) * ZD;
                    *reinterpret_cast<f32x4*>(dst + 4 * h) = f32x4{z[0], z[1], z[2], z[3]};
                    *reinterpret_cast<f32x4*>(dst + 8 + 4 * h) = f32x4{z[4], z[5], z[6], z[7]};
                }
                pend_m = -1;
            }
        };
        // The decoder variances of the kept samples (compute_Vs, mcem.py:280-290) are those of the chain's STATE at the kept steps, and a
        // state's variances were computed by the pass that proposed it.  Neither registers nor LDS have room for a second set of 64 values
        // per lane here, so a kept step r stores exp(pre-activation) of its PROPOSAL to Vs[r] as it computes it; where the proposal was
        // rejected the state is that of step r - 1, and the lane copies its own values Vs[r - 1] -> Vs[r] behind the chain (runs of
        // rejections: one load, several stores); the state at the end of the burn-in takes ONE decoder pass.  42 passes per chain of
        // 30 + 10 steps instead of 51, the same bits (tested against the decode mode).
        // Exact fp32 only: a copied row costs its HBM bytes (4.5 - 6.4 us per kept sample at 25 utterances, measured), a pass 12.6 us under
        // fp32 but 5.2 / 4 us under the bf16 policies -- those keep the decoder passes (chain launch, 25 utterances, 30 + 10 steps:
        // fp32 748 -> 680 us; bf16x3 370 -> 361, with 75 + 25 steps 771 -> 795: profiles/r05_mcem_tile32_no_decode_ab.txt)
        const bool want_vs = sizeof(T) == 4 && g.Vs != nullptr && g.nit > 0 && g.R >= 1 && g.R <= 64;
        constexpr int OOR = 0x7fffffff;                                    // a per-lane offset behind every buffer: the access is dropped
        if (wave_u == 0 && h == 0) rejs[l31] = 0ull;                       // (touched by these lanes only until the chain has ended)
        if (g.dbg) tlast = __builtin_amdgcn_s_memtime();
        const unsigned long long t_loop = tlast;
        for (int m = mstart; m < mend; ++m) {
            const bool keepst = want_vs && m >= g.burnin;
            float prior_p = 0.f, lu_cur = 0.f;
            if (wave_u == 0) {
                if (m >= 0) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) zp[r] = z[r] + g.sd * nzv[r];                     // mcem.py:244
                    // a copy hipcc cannot sink below the request of the next step's draws: with a plain assignment the old value stayed live
                    // across that load, the new one landed in a temporary, and the loop-carried copy waited for it on the spot
                    asm volatile("v_mov_b32 %0, %1" : "=v"(lu_cur) : "v"(lu));
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) zp[r] = z[r];
                }
                float zv[16];
#pragma unroll
                for (int r = 0; r < 8; ++r) { zv[r] = zp[r]; zv[r + 8] = 0.f; prior_p += zp[r] * zp[r]; }
                put_lds<P>(zv, Zb, LDZ, 0, l31, h);
                prior_p = xsum32(prior_p);
                __builtin_amdgcn_sched_barrier(0);
                if (m >= 0) load_draws(m + 1);
            }
            stamp(0);
            __syncthreads();                                               // B0
            stamp(1);
            double ll = 0.0;
            float slog = 0.f, sdiv = 0.f;                                  // sums of log2(vx) and x2 / vx over one tile
            auto epi_begin = [&](auto) { slog = 0.f; sdiv = 0.f; };
            auto epi_end = [&](auto) { ll += (double)fmaf(slog, 0.693147180559945309f, sdiv); };
            // (two instances of the pass: the stores of the kept steps cost registers the other steps do not have to pay for)
            if (keepst) {
                const int voff_st = live ? voff : OOR;
                float* const vs_m = g.Vs + (int64_t)(m - g.burnin) * XD * g.N;
                const __amdgpu_buffer_rsrc_t rs_vsm = __builtin_amdgcn_make_buffer_rsrc(vs_m, 0, fn_bytes, 0x00020000);
                pass(
                    epi_begin,
                    [&](auto, int t, auto rc, float a, float x2, float vb) {
                        const float ea = P::exp_(a);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ea), rs_vsm, voff_st, soff(t, decltype(rc)::value), 0);
                        const float vx = fmaf(g_n, ea, vb);                                       // mcem.py:248-249
                        slog += __builtin_amdgcn_logf(vx);                                        // mcem.py:252-253: log(vx) + x2 / vx
                        sdiv = fmaf(x2, __builtin_amdgcn_rcpf(vx), sdiv);
                    },
                    epi_end,
                    [&](float a) {
                        const float ea = P::exp_(a);
                        if (live && h == 0) vs_m[(int64_t)512 * g.N + nf] = ea;
                        const float vx = fmaf(g_n, ea, vb_512);
                        const float term = P::log_(vx) + P::div_(x2_512, vx);
                        if (h == 0) ll += (double)term;
                    });
            } else {
                pass(
                    epi_begin,
                    [&](auto, int, auto, float a, float x2, float vb) {
                        const float vx = fmaf(g_n, P::exp_(a), vb);                               // mcem.py:248-249
                        slog += __builtin_amdgcn_logf(vx);                                        // mcem.py:252-253: log(vx) + x2 / vx
                        sdiv = fmaf(x2, __builtin_amdgcn_rcpf(vx), sdiv);
                    },
                    epi_end,
                    [&](float a) {
                        const float vx = fmaf(g_n, P::exp_(a), vb_512);
                        const float term = P::log_(vx) + P::div_(x2_512, vx);
                        if (h == 0) ll += (double)term;
                    });
            }
            ll = xsum32(ll);
            if (h == 0) red[wave_u * TB + l31] = ll;
            if (wave_u == 0) flush_step();                                 // (the last step's kept sample and trace: in front of B3, where wave 0 waits for wave 3's bin 512 -- not in the serial section)
            stamp(7);
            __syncthreads();                                               // B3
            stamp(8);
            if (wave_u == 0) {
                const double ll_p = red[l31] + red[TB + l31] + red[2 * TB + l31] + red[3 * TB + l31];
                if (m < 0) {
                    ll_cur = ll_p; prior_cur = prior_p;
                } else {
                    const float acc_prob = (float)(ll_cur - ll_p) + 0.5f * (prior_cur - prior_p);   // mcem.py:252-254
                    const bool is_acc = lu_cur < acc_prob;                                           // mcem.py:257
                    if (is_acc) {
                        ll_cur = ll_p; prior_cur = prior_p;
#pragma unroll
                        for (int r = 0; r < 8; ++r) z[r] = zp[r];
                    }
                    if (!is_acc && m >= g.burnin && h == 0) rejs[l31] |= 1ull << ((m - g.burnin) & 63);
                    pend_m = m; pend_prob = acc_prob; pend_acc = is_acc;                             // stored behind the next proposal (flush_step)
                }
                if (m == g.burnin - 1) {                                                             // the state the kept steps start from
                    *reinterpret_cast<f32x4*>(zsv + lane * 8) = f32x4{z[0], z[1], z[2], z[3]};
                    *reinterpret_cast<f32x4*>(zsv + lane * 8 + 4) = f32x4{z[4], z[5], z[6], z[7]};
                }
            }
            // red / p512 / Zb are next written behind the barriers of the following pass
        }
        if (wave_u == 0) flush_step();
        if (wave_u == 0 && g.nit > 0 && g.Zlast != nullptr && live) {          // the chain's final state (a frame's Z0 is read by these lanes only: Zlast may be Z0)
#pragma unroll
            for (int r = 0; r < 8; ++r) g.Zlast[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * h) * g.N + nf] = z[r];
        }
        const unsigned long long t_chain_end = g.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
        if (g.dbg && lane == 0 && g.nit > 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + k] = tsum[k];
            g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + 9] = (unsigned long long)(mend - mstart);
            g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + 10] = t_loop - t_entry;
            g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + 13] = t_w - t_entry;
            g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + 11] = t_chain_end - t_loop;
        }

        if (want_vs) {
            if (wave_u == 0) {
                float zv[16];
#pragma unroll
                for (int r = 0; r < 8; ++r) { zv[r] = zsv[lane * 8 + r]; zv[r + 8] = 0.f; }
                put_lds<P>(zv, Zb, LDZ, 0, l31, h);
            }
            __syncthreads();
            const unsigned long long myrej = rejs[l31];
            // Vs[0] of the frames whose first kept step was rejected: the state at the end of the burn-in
            if (__builtin_amdgcn_ballot_w64((myrej & 1ull) != 0ull) != 0ull) {
                const bool st0 = live && (myrej & 1ull) != 0ull;
                const int voff0 = st0 ? voff : OOR;
                const __amdgpu_buffer_rsrc_t rs_v0 = __builtin_amdgcn_make_buffer_rsrc(g.Vs, 0, fn_bytes, 0x00020000);
                pass(
                    [](auto) {},
                    [&](auto, int t, auto rc, float a, float, float) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, P::exp_(a)), rs_v0, voff0, soff(t, decltype(rc)::value), 0);
                    },
                    [](auto) {},
                    [&](float a) { if (st0 && h == 0) g.Vs[(int64_t)512 * g.N + nf] = P::exp_(a); });
            }
            // rejected kept steps r >= 1: Vs[r] <- Vs[r - 1], lane by lane (a lane reads what it stored itself)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float buf[NTW][16], buf512 = 0.f;
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) buf[tt][r] = 0.f;
            bool have = false;
            const bool own512 = wave_u == 3 && h == 0;
            for (int rk = 1; rk < g.R; ++rk) {
                const bool rj = ((myrej >> rk) & 1ull) != 0ull;
                if (__builtin_amdgcn_ballot_w64(rj) == 0ull) { have = false; continue; }
                const bool need = rj && !have && live;
                float* const src = g.Vs + (int64_t)(rk - 1) * XD * g.N;
                float* const dst = g.Vs + (int64_t)rk * XD * g.N;
                if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(src, 0, fn_bytes, 0x00020000);
                    const int vo = need ? voff : OOR;
#pragma unroll
                    for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_src, vo, soff(NTW * wave_u + tt, r), 16));
                            buf[tt][r] = need ? x : buf[tt][r];
                        }
                    if (own512 && need) buf512 = __builtin_nontemporal_load(src + (int64_t)512 * g.N + nf);
                }
                have = rj;
                const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst, 0, fn_bytes, 0x00020000);
                const int vs = (rj && live) ? voff : OOR;
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, buf[tt][r]), rs_dst, vs, soff(NTW * wave_u + tt, r), 0);
                if (own512 && rj && live) dst[(int64_t)512 * g.N + nf] = buf512;
                // (no wait for these stores: a lane that stored slot rk holds its values -- it never reads them back)
            }
        }
        // ---- decode mode (dvae_mcem_decode), and chains keeping more than 64 samples: Vs[r] = decoder([Zs[:, r, :] | y])  (mcem.py:280-290) ----
        if (g.Vs != nullptr && !want_vs) {
            for (int r_s = 0; r_s < g.R; ++r_s) {
                __syncthreads();
                if (wave_u == 0) {
                    const float* src = g.Zs + ((int64_t)nf * g.R + r_s) * ZD;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(src + 4 * h);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(src + 8 + 4 * h);
                    float zv[16];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { zv[r] = s0[r]; zv[4 + r] = s1[r]; zv[8 + r] = 0.f; zv[12 + r] = 0.f; }
                    put_lds<P>(zv, Zb, LDZ, 0, l31, h);
                }
                __syncthreads();
                float* const vs_r = g.Vs + (int64_t)r_s * XD * g.N;
                const __amdgpu_buffer_rsrc_t rs_vs = __builtin_amdgcn_make_buffer_rsrc(vs_r, 0, fn_bytes, 0x00020000);
                pass(
                    [](auto) {},
                    [&](auto, int t, auto rc, float a, float, float) {
                        constexpr int r = decltype(rc)::value;
                        if (live) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, P::exp_(a)), rs_vs, voff, soff(t, r), 0);
                    },
                    [](auto) {},
                    [&](float a) { if (live && h == 0) vs_r[(int64_t)512 * g.N + nf] = P::exp_(a); });
            }
        }
        __syncthreads();
        if (g.dbg && lane == 0 && g.nit > 0) g.dbg[((size_t)blockIdx.x * 4 + wave_u) * 16 + 12] = __builtin_amdgcn_s_memtime() - t_chain_end;
    }
}

template <typename P, int YP>
static int launch_resident_t(const MhArgs& a, hipStream_t s) {
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)mcem_resident_kernel<P, YP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C8_LDS);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(mcem_resident_kernel, %zu B LDS): %s", C8_LDS, hipGetErrorString(e)); return (int)e; }
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((mcem_resident_kernel<P, YP>), dim3(a.ntiles), dim3(256), C8_LDS, s, a);
    DVAE_LAUNCH_OK("mcem_resident_kernel");
    return 0;
}

bool resident_chain_supported(int precision, int yp) { return (precision == DVAE_PREC_BF16X3 || precision == DVAE_PREC_BF16 || precision == DVAE_PREC_F32) && (yp == 0 || yp == 16 || yp == C8_YP); }

int launch_resident_chain(int precision, int yp, const MhArgs& a, hipStream_t s) {
    if ((int64_t)XD * a.N * 4 >= ((int64_t)1 << 31) || (int64_t)a.nit * ZD * a.N * 4 >= ((int64_t)1 << 31)) {
        set_error("mcem resident chain: (F, N) matrices of 2 GB and more are not addressed");
        return DVAE_E_UNSUPPORTED;
    }
    // chains too short to fill the chip with 32-frame tiles (one utterance of 300 frames: ten tiles) run on 16-frame tiles: a step is bound
    // by its epilogue, so half the frames per workgroup take about half the time (mcem_resident16.hip)
    const char* const tile_s = getenv("DVAE_MCEM_TILE");                   // A/B and test switch, read per call
    const int tile_env = tile_s ? atoi(tile_s) : 0;
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    // exact fp32: every MFMA shape has the same rate and nothing overlaps the matrix work, so the narrowest tile wins while its workgroups
    // fit the chip in two rounds (mcem_resident4.hip: 2.3 us per step against 7.5 on 16 frames)
    const int64_t tiles4 = (a.N + 3) / 4;
    if ((tile_env == 0 || tile_env == 4) && resident4_chain_supported(precision, yp) && (tile_env == 4 || tiles4 <= 2 * n_cu)) {
        MhArgs b = a;
        b.ntiles = (int)tiles4;
        return launch_resident4_chain(yp, b, s);
    }
    const int64_t tiles16 = (a.N + 15) / 16;
    if (tile_env != 32 && resident16_chain_supported(precision, yp) && (tile_env == 16 || tiles16 <= n_cu)) {
        MhArgs b = a;
        b.ntiles = (int)tiles16;
        return launch_resident16_chain(precision, yp, b, s);
    }
    if (yp == 0) return precision == DVAE_PREC_BF16X3 ? launch_resident_t<PolX3C, 0>(a, s) : precision == DVAE_PREC_BF16 ? launch_resident_t<PolB1C, 0>(a, s) : launch_resident_t<PolF32C, 0>(a, s);
    if (yp == 16) return precision == DVAE_PREC_BF16X3 ? launch_resident_t<PolX3C, 16>(a, s) : precision == DVAE_PREC_BF16 ? launch_resident_t<PolB1C, 16>(a, s) : launch_resident_t<PolF32C, 16>(a, s);
    if (yp == C8_YP) return precision == DVAE_PREC_BF16X3 ? launch_resident_t<PolX3C, C8_YP>(a, s) : precision == DVAE_PREC_BF16 ? launch_resident_t<PolB1C, C8_YP>(a, s) : launch_resident_t<PolF32C, C8_YP>(a, s);
    set_error("mcem resident chain: label rows 0, 1..16 or 513 only");
    return DVAE_E_UNSUPPORTED;
}

}  // namespace fused
}  // namespace dvae
