// Weight stream of a chain wave of the 8-wave rows kernel (train_rows2.hip).
//
// The weight fragments a chain wave consumes over one tile -- every layer, forward and backward -- form ONE static sequence
// of k-steps ("positions"), described at compile time as a list of segments (matrix, number of k-steps, byte stride between
// k-steps).  A ring of D fragment registers walks that sequence: the slot a k-step has consumed is re-requested for position
// + D at once, whatever segment (layer, tile of the output layer, next tile of the persistent loop) that position belongs to.
// So D fragments per wave are in flight ALL the time -- across epilogues, barriers and layer boundaries -- instead of a ring
// that drains at the end of every GEMM and refills (one exposed fetch latency per GEMM start, ~16 per tile) at the next.
// Ring slots are compile-time indices (position % D); the sequence length is padded to a multiple of D so the slot phase is
// the same for every tile of the persistent loop.
#pragma once
#include <type_traits>
#include "fused_tiles.hpp"

#ifndef R2_WAUX
#define R2_WAUX 0
#endif
// Work the helper waves take off the chain (both waves of a SIMD then keep a weight ring in flight: the k-step of a GEMM phase is
// latency-bound by loads in flight per SIMD, t = t0 + L / D):
//   R2_NX1   : k-steps of the encoder's x block (33) the CHAIN computes; the partner helper wave computes the rest of the same row tile
//              during the L1 x phase and hands its partial 32 x 32 tile over through LDS (the fp32 h1 slot, free until the epilogue).
//              Default 33 = off: on a workgroup's FIRST tile the helper's label-tile requests are in flight in that phase and vmcnt
//              retires in order, so a helper wave cannot wait for a weight fragment without waiting for the whole label tile.
//   R2_HELPY : 513-label models: the label block of decoder layer 1 (33 k-steps, independent of z) runs on the helper waves during
//              the chain's L1 y GEMM, and the helpers finish decoder layer 1 themselves (the z block is one k-step): no hand-over.
//              Default 0 = off, measured (round 3, same box, alternating): the heads phase loses the 33 k-steps (4.3 -> 1.4 us) and the
//              L1 y phase gains 2.5 us -- two GEMM streams on one SIMD finish 66 k-steps in 5.9 us, 89 ns per k-step against 100 ns
//              for one stream: the k-step is not bound by loads in flight per SIMD but by what the CU's vector-memory path delivers
//              beside the MFMAs (8 KB per k-step and CU) -- step 75.4 us with, 74.8 us without.
#ifndef R2_NX1
#define R2_NX1 33
#endif
#ifndef R2_HELPY
#define R2_HELPY 0
#endif

namespace dvae {
namespace fused {

// segments in consumption order (G_W5A..D: the four output-layer tiles of a wave; G_PAD: dummy positions)
// G_A1 .. G_A1T: M2_info's auxiliary classifier on z (16-128-128-1), forward and backward, between the heads and decoder layer 1
enum { G_W1X, G_W1Y, G_W2, G_WMV, G_W3Y, G_A1, G_A2, G_A2T, G_A1T, G_W3Z, G_W4, G_W5A, G_W5B, G_W5C, G_W5D, G_W5T, G_W4T, G_W3ZT, G_WMVT, G_W2T, G_PAD, G_N };

template <typename P, int YP, bool YENC, int D, bool INFO = false> struct Sched {
    static constexpr int KS = P::KSTEP;
    static constexpr int NSEG = G_N;
    static constexpr unsigned FBB = 64u * P::E * sizeof(typename P::T);          // bytes of one (row tile, k-step) fragment block
    static constexpr int NX1 = (R2_NX1 > 0 && R2_NX1 < XP / KS) ? R2_NX1 : XP / KS;   // the chain's share of the x block of layer 1
    static constexpr bool HELPY = R2_HELPY && YP == XP;                            // decoder layer 1 lives on the helper waves
    static constexpr int raw(int s) {
        return s == G_W1X ? NX1 : s == G_W1Y ? (YENC ? YP / KS : 0) : s == G_W3Y ? (HELPY ? 0 : YP / KS) : s == G_W3Z ? (HELPY ? 0 : ZD / KS)
             : s == G_A1 ? (INFO ? ZD / KS : 0) : (s == G_A2 || s == G_A2T || s == G_A1T) ? (INFO ? HD / KS : 0)
             : s == G_W5T ? NO / KS : s == G_WMVT ? 32 / KS : s == G_PAD ? 0 : HD / KS;
    }
    static constexpr int sum_raw() { int t = 0; for (int s = 0; s < G_PAD; ++s) t += raw(s); return t; }
    static constexpr int pad = (D - sum_raw() % D) % D;
    static constexpr int total = sum_raw() + pad;
    static constexpr bool WRAP = true;
    static constexpr int n(int s) { return s == G_PAD ? pad : raw(s); }
    static constexpr int start(int s) { int t = 0; for (int i = 0; i < s; ++i) t += n(i); return t; }
    static constexpr int seg_of(int q) { int s = 0; while (q >= start(s) + n(s)) ++s; return s; }
    // k-step stride: 4-tile matrices are [k-step][4 tiles], the single-tile heads [k-step][1], the output layer [k-step][17]
    static constexpr unsigned stride(int s) { return (s == G_WMV || s == G_W3ZT || s == G_A1T) ? FBB : (s >= G_W5A && s <= G_W5D) ? NT_OUT * FBB : 4u * FBB; }
};

// the helper waves' stream: the rest of the x block of layer 1, then (513-label models) decoder layer 1; M2_info: the classifier on x
// (513-128-128-1: layer 1, layer 2, backward through layer 2), which runs beside the chain's encoder
enum { H_W1X, H_W3Y, H_W3Z, H_C1, H_C2, H_C2T, H_PAD, H_N };
template <typename P, int YP, bool YENC, int D, bool INFO = false> struct HSched {
    typedef Sched<P, YP, YENC, D, INFO> C;
    static constexpr int KS = P::KSTEP;
    static constexpr int NSEG = H_N;
    static constexpr unsigned FBB = C::FBB;
    static constexpr bool HELPY = C::HELPY;
    static constexpr int raw(int s) {
        return s == H_W1X ? XP / KS - C::NX1 : s == H_W3Y ? (C::HELPY ? YP / KS : 0) : s == H_W3Z ? (C::HELPY ? ZD / KS : 0)
             : s == H_C1 ? (INFO ? XP / KS : 0) : (s == H_C2 || s == H_C2T) ? (INFO ? HD / KS : 0) : 0;
    }
    static constexpr int sum_raw() { int t = 0; for (int s = 0; s < H_PAD; ++s) t += raw(s); return t; }
    // the helpers' stream does not wrap: it is started (fill) when the tile's label image is about to be committed and requests nothing
    // past its last position, so no fragment registers stay live through the rest of the tile
    static constexpr int pad = 0;
    static constexpr int total = sum_raw();
    static constexpr bool any = sum_raw() > 0;
    static constexpr bool WRAP = false;
    static constexpr int n(int s) { return s == H_PAD ? pad : raw(s); }
    static constexpr int start(int s) { int t = 0; for (int i = 0; i < s; ++i) t += n(i); return t; }
    static constexpr int seg_of(int q) { int s = 0; while (q >= start(s) + n(s)) ++s; return s; }
    static constexpr unsigned stride(int s) { return 4u * FBB; }
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <typename P, typename SC, int D, int AUX = R2_WAUX> struct WStream {
    typename P::Frag r[D][P::NP];
    __amdgpu_buffer_rsrc_t rs;
    int voff;                  // lane * 16
    unsigned pl;               // bytes between the hi and lo planes
    unsigned sb[SC::NSEG];     // byte offset of k-step 0 of every segment for THIS wave
    unsigned cur;              // byte offset of the position requested last: positions are requested strictly in sequence
    // The soffset of every request is ONE running SGPR: + stride inside a segment, = sb[segment] at its first k-step, opaque to the
    // optimiser after each update.  (Written as sb[s] + constant, hipcc computes all ~430 offsets of the sequence once per kernel --
    // they are invariant over the tile loop -- keeps them in VGPR lanes (529 "SGPR spills") and pays a v_readlane + s_nop 4 in front of
    // EVERY weight load, on the chain's critical path.)
    template <int Q> __device__ __forceinline__ void req() {
        constexpr int s = SC::seg_of(Q);
        if constexpr (Q == SC::start(s)) cur = sb[s];
        else cur += SC::stride(s);
        asm volatile("" : "+s"(cur));
        // R2_WAUX: cache-policy bits of the weight stream's loads (gfx950 buffer aux: 1 = sc0, 2 = nt, 16 = sc1)
        const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, cur, AUX);
        r[Q % D][0] = __builtin_bit_cast(typename P::Frag, v0);
        if constexpr (P::NP == 2) {
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, cur + pl, AUX);
            r[Q % D][1] = __builtin_bit_cast(typename P::Frag, v1);
        }
    }
    // the next position of segment S (not its first) into ring slot SLOT: the request of a rolled lap of gemm_seg, where the position
    // is a run-time quantity but the segment, the stride and the slot are not
    template <int S, int SLOT> __device__ __forceinline__ void req_next() {
        cur += SC::stride(S);
        asm volatile("" : "+s"(cur));
        const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, cur, AUX);
        r[SLOT][0] = __builtin_bit_cast(typename P::Frag, v0);
        if constexpr (P::NP == 2) {
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, cur + pl, AUX);
            r[SLOT][1] = __builtin_bit_cast(typename P::Frag, v1);
        }
    }
    __device__ __forceinline__ void fill() {
        static_for<0, (D < SC::total ? D : SC::total)>([&](auto ic) { this->template req<decltype(ic)::value>(); });
        __builtin_amdgcn_sched_barrier(0);
    }
};

// acc += W(segment S) * act^T.  `active` (wave-uniform): false for the waves that do not own this segment (the single-tile heads
// belong to wave 0): they skip the arithmetic but keep requesting, so every wave's ring stays in phase with the schedule.
// F16: both operands of this segment hold split-fp16 planes (fused_tiles.hpp: struct X16)
template <typename P, typename SC, int D, int S, typename WS, bool F16 = false>
__device__ __forceinline__ void gemm_seg(f32x16& acc, WS& w, const typename P::T* brow, bool blo = true, bool active = true) {
    typedef typename P::Frag Frag;
    constexpr int N = SC::n(S), Q0 = SC::start(S), STR = 2 * P::E;
    if constexpr (N > 0) {
        constexpr int BD = N < P::BDMAX ? N : P::BDMAX;
        Frag bq[BD][P::NP];
        if (active) {
#pragma unroll
            for (int i = 0; i < BD; ++i) bloadp<P>(bq[i], brow + i * STR);
        }
        // R2_ROLL=1 (experiment, off): long segments run the k-steps whose request (position + D) stays inside the segment as a ROLLED
        // loop of ring laps -- the same D k-steps of code executed NL times.  Motive: the kernel's straight-line code (~110 KB) does not fit
        // the 64 KB instruction cache two CUs share and a tile executes every k-step's code once (tools/r03/kstep_bench.hip: 145 -> 168
        // clocks per k-step with the instruction cache invalidated every 18 k-steps).  Same operations in the same order (the fused tests
        // pass bit for bit).  Measured, same box, alternating: 75.9 / 76.6 / 76.9 against 78.0 / 76.9 / 77.6 us per step, per-workgroup
        // median 39.3 against 39.0 us, L1 x GEMM 3.12 against 3.32 us, L1 y phase 5.44 against 5.24: no gain beyond the noise, with
        // 258 instead of 96 SGPR spills and 2 VGPR spills -- instruction fetch is not what separates the kernel's 100 ns k-step from
        // the micro-benchmark's 72.
#ifndef R2_ROLL
#define R2_ROLL 0
#endif
        constexpr int NL = (R2_ROLL && D % BD == 0 && N >= 3 * D) ? (N - D) / D : 0;     // laps (at least two)
        constexpr int NR = NL * D;                                                  // k-steps of the rolled part
        if constexpr (NL > 0) {
            const typename P::T* bl = brow;
#pragma unroll 1
            for (int lap = 0; lap < NL; ++lap) {
                static_for<0, D>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if (active) {
#ifndef R2_NOMFMA
                        if constexpr (F16) mmap_f16<P>(acc, w.r[(Q0 + i) % D], bq[i % BD]);
                        else mmap<P>(acc, w.r[(Q0 + i) % D], bq[i % BD], blo);
#else
                        acc[0] += (float)w.r[(Q0 + i) % D][0][0] + (float)bq[i % BD][0][0];
#endif
#ifndef R2_NOBLOAD
                        bloadp<P>(bq[i % BD], bl + (i + BD) * STR);                 // (lap * D + i + BD) < N: NR + BD - 1 <= N - 1
#endif
                    }
#ifndef R2_NOWLOAD
                    w.template req_next<S, (Q0 + i) % D>();                        // position Q0 + lap * D + i + D: inside segment S
#endif
                    __builtin_amdgcn_sched_barrier(0);
                });
                bl += D * STR;
            }
        }
        static_for<NR, N>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if (active) {
#ifndef R2_NOMFMA
                if constexpr (F16) mmap_f16<P>(acc, w.r[(Q0 + I) % D], bq[I % BD]);
                else mmap<P>(acc, w.r[(Q0 + I) % D], bq[I % BD], blo);
#else
                acc[0] += (float)w.r[(Q0 + I) % D][0][0] + (float)bq[I % BD][0][0];      // experiment: operands consumed, no MFMA
#endif
#ifndef R2_NOBLOAD
                if constexpr (I + BD < N) bloadp<P>(bq[I % BD], brow + (I + BD) * STR);
#endif
            }
#ifndef R2_NOWLOAD
            if constexpr (SC::WRAP) w.template req<(Q0 + I + D) % SC::total>();
            else if constexpr (Q0 + I + D < SC::total) w.template req<Q0 + I + D>();
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

}  // namespace fused
}  // namespace dvae
