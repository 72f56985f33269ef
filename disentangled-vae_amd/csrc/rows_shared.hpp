// Shared by the rows kernels train_rows2.hip (4 chain + 4 helper waves) and train_rows3.hip (8 GEMM + 4 helper waves): the LDS carve, the
// LDS-only workgroup barrier, the raw fp32 hand-over of the output layer, the diagnostic stamps.
#pragma once
#include "fused_tiles.hpp"
#include "rows_common.hpp"

namespace dvae {
namespace fused {

template <typename P, bool INFO = false> struct Lds2 {
    typedef typename P::T T;
    static constexpr int nbias = Ld<T>::nbias;
    static constexpr size_t o_bias = (size_t)Ld<T>::act_elems * P::NP * sizeof(T);
    static constexpr size_t o_red = o_bias + (size_t)nbias * sizeof(float);
    static constexpr size_t o_flags = o_red + 16 * sizeof(float);
    static constexpr size_t o_rows = o_flags + 16 * sizeof(int);
    static constexpr size_t o_keep = o_rows + 2 * TB * sizeof(int64_t);                  // fp32 h1 | h2 of the chain waves: [2][16][256]
    static constexpr size_t o_keepz = o_keep + 2 * 16 * 256 * sizeof(float);             // fp32 mu | log_var of wave 0: [16][64]
    // M2_info: the classifier / auxiliary-net tables (bc1 bc2 wc3 ba1 ba2 wa3, then bc3, ba3), the partial output dot products of the
    // four waves of a side net ([4][32]), and d BCE_aux / d z of wave 0's latent tile ([8][64], kept until the backward z phase)
    static constexpr size_t o_info = o_keepz + 16 * 64 * sizeof(float);
    static constexpr size_t o_red2 = o_info + (INFO ? (size_t)Ld<T>::ninfo * sizeof(float) : 0);
    static constexpr size_t o_dzu = o_red2 + (INFO ? 144 * sizeof(float) : 0);      // [128], [129]: the tile's BCE sums (classifier, auxiliary)
    static constexpr size_t bytes = o_dzu + (INFO ? 8 * 64 * sizeof(float) : 0);
    static_assert(o_bias % 16 == 0 && o_rows % 8 == 0 && o_info % 16 == 0, "LDS carve alignment");
    static_assert(bytes <= 160 * 1024, "LDS budget");
};

// Workgroup barrier that orders LDS only: waits for this wave's LDS operations, not for its global stores (the helpers'
// stash stores stay in flight across phases; __syncthreads() carries a fence that drains vmcnt at every barrier, which made
// the chain wait for the write acknowledgements of every stash tile: measured 60 -> 3x us per tile).  Global data handed
// between the roles does not exist: the stash is consumed by the NEXT kernel, inputs are read-only.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Output-layer hand-over between the roles (train step, two operand planes): a chain wave leaves the fp32 pre-activations of its
// 32 x 32 tile in the tile's OWN columns of U -- upper 16 bits of every value in plane 0, lower 16 bits in plane 1 -- and the partner
// helper wave turns them in place into the (hi, lo) planes of da.  Same lane, same elements on both sides: no extra LDS.
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
template <typename P>
__device__ __forceinline__ void put_raw4(const float (&v)[4], typename P::T* lds, int ldl, int col, int l31) {
    u16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const unsigned b = __float_as_uint(v[j]); hi[j] = (unsigned short)(b >> 16); lo[j] = (unsigned short)(b & 0xffffu); }
    *reinterpret_cast<u16x4*>(lds + l31 * ldl + col) = hi;
    *reinterpret_cast<u16x4*>(lds + Pl<P>::lds + l31 * ldl + col) = lo;
}
template <typename P>
__device__ __forceinline__ void get_raw4(float (&v)[4], const typename P::T* lds, int ldl, int col, int l31) {
    const u16x4 hi = *reinterpret_cast<const u16x4*>(lds + l31 * ldl + col);
    const u16x4 lo = *reinterpret_cast<const u16x4*>(lds + Pl<P>::lds + l31 * ldl + col);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(((unsigned)hi[j] << 16) | (unsigned)lo[j]);
}

#define R2_STAMP(i) do { if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#ifdef R2_FINE      // diagnostic build: the helper's stamp slots 16 .. 29 carry chain-side sub-phase stamps of the output layer instead
#define R2_HSTAMP(i) do { } while (0)
#define R2_FSTAMP(i) do { if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define R2_HSTAMP(i) do { if (g.dbg && ht == 0) g.dbg[(size_t)blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#define R2_FSTAMP(i) do { } while (0)
#endif


}  // namespace fused
}  // namespace dvae
