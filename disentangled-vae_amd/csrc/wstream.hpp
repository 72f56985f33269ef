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

namespace dvae {
namespace fused {

// segments in consumption order (G_W5A..D: the four output-layer tiles of a wave; G_PAD: dummy positions)
enum { G_W1X, G_W1Y, G_W2, G_WMV, G_W3Y, G_W3Z, G_W4, G_W5A, G_W5B, G_W5C, G_W5D, G_W5T, G_W4T, G_W3ZT, G_WMVT, G_W2T, G_PAD, G_N };

template <typename P, int YP, bool YENC, int D> struct Sched {
    static constexpr int KS = P::KSTEP;
    static constexpr unsigned FBB = 64u * P::E * sizeof(typename P::T);          // bytes of one (row tile, k-step) fragment block
    static constexpr int raw(int s) {
        return s == G_W1X ? XP / KS : s == G_W1Y ? (YENC ? YP / KS : 0) : s == G_W3Y ? YP / KS : s == G_W3Z ? ZD / KS
             : s == G_W5T ? NO / KS : s == G_WMVT ? 32 / KS : s == G_PAD ? 0 : HD / KS;
    }
    static constexpr int sum_raw() { int t = 0; for (int s = 0; s < G_PAD; ++s) t += raw(s); return t; }
    static constexpr int pad = (D - sum_raw() % D) % D;
    static constexpr int total = sum_raw() + pad;
    static constexpr int n(int s) { return s == G_PAD ? pad : raw(s); }
    static constexpr int start(int s) { int t = 0; for (int i = 0; i < s; ++i) t += n(i); return t; }
    static constexpr int seg_of(int q) { int s = 0; while (q >= start(s) + n(s)) ++s; return s; }
    // k-step stride: 4-tile matrices are [k-step][4 tiles], the single-tile heads [k-step][1], the output layer [k-step][17]
    static constexpr unsigned stride(int s) { return (s == G_WMV || s == G_W3ZT) ? FBB : (s >= G_W5A && s <= G_W5D) ? NT_OUT * FBB : 4u * FBB; }
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <typename P, typename SC, int D> struct WStream {
    typename P::Frag r[D][P::NP];
    __amdgpu_buffer_rsrc_t rs;
    int voff;                  // lane * 16
    unsigned pl;               // bytes between the hi and lo planes
    unsigned sb[G_N];          // byte offset of k-step 0 of every segment for THIS wave
    template <int Q> __device__ __forceinline__ void req() {
        constexpr int s = SC::seg_of(Q);
        constexpr unsigned off = (unsigned)(Q - SC::start(s)) * SC::stride(s);
        // R2_WAUX: cache-policy bits of the weight stream's loads (gfx950 buffer aux: 1 = sc0, 2 = nt, 16 = sc1)
        const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sb[s] + off, R2_WAUX);
        r[Q % D][0] = __builtin_bit_cast(typename P::Frag, v0);
        if constexpr (P::NP == 2) {
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sb[s] + off + pl, R2_WAUX);
            r[Q % D][1] = __builtin_bit_cast(typename P::Frag, v1);
        }
    }
    __device__ __forceinline__ void fill() {
        static_for<0, D>([&](auto ic) { this->template req<decltype(ic)::value>(); });
        __builtin_amdgcn_sched_barrier(0);
    }
};

// acc += W(segment S) * act^T.  `active` (wave-uniform): false for the waves that do not own this segment (the single-tile heads
// belong to wave 0): they skip the arithmetic but keep requesting, so every wave's ring stays in phase with the schedule.
template <typename P, typename SC, int D, int S, typename WS>
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
        static_for<0, N>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if (active) {
#ifndef R2_NOMFMA
                mmap<P>(acc, w.r[(Q0 + I) % D], bq[I % BD], blo);
#else
                acc[0] += (float)w.r[(Q0 + I) % D][0][0] + (float)bq[I % BD][0][0];      // experiment: operands consumed, no MFMA
#endif
#ifndef R2_NOBLOAD
                if constexpr (I + BD < N) bloadp<P>(bq[I % BD], brow + (I + BD) * STR);
#endif
            }
#ifndef R2_NOWLOAD
            w.template req<(Q0 + I + D) % SC::total>();
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

}  // namespace fused
}  // namespace dvae
