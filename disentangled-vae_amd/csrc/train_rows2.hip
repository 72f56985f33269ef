// rows kernel, second generation: one 512-thread workgroup (8 waves, two per SIMD) per 32-frame tile.
//
//   chain waves 0-3  : the serial layer chain of the train step -- per layer one MFMA GEMM on 32x32 tiles
//                      (out^T[features x frames] = W * in^T, weights streamed from L2 into a register ring, activations
//                      from LDS), the bias / tanh / loss epilogue in registers, the next layer's operand into LDS.
//                      Nothing else: a chain wave never touches the inputs' HBM stream or the stash.
//   helper waves 4-7 : everything beside the chain, on the SAME SIMDs (VALU / LDS / memory instructions issue beside
//                      the partner wave's MFMAs): the x / y tiles HBM -> registers -> (hi, lo) bf16 planes in LDS, the
//                      LDS -> stash transposition (ds_read_b64_tr_b16) and every stash store, the bias table, the
//                      gather table, and -- in the persistent tile loop -- the NEXT tile's x load, issued during the
//                      backward phases and committed once the output-gradient tile has been consumed.
// Both roles run the same sequence of workgroup barriers (the phase list below); a helper always transposes the
// buffer the chain is reading in that phase (both only read it), while the chain's epilogue fills the other one.
// Compared with the 4-wave kernel of train_fused.hip (one wave per SIMD, 512 registers, 136 of them staging
// registers, stash stores and tile loads on the chain's critical path): <= 256 registers per wave, a second
// instruction stream per SIMD, the loss epilogue reads x straight from global memory into registers (no LDS slices,
// no barriers inside the output layer), and the label part of decoder layer 1 is computed while the label tile is
// in LDS for the encoder (it does not depend on z).
//
// phase list (barrier after each; [C] chain, [H] helpers).  U = [frame][544] tile buffer, Ha/Hb = [frame][128], Zb = [frame][32]
//   (first tile only)  [H] gather table            | BROW (only with a gather table)
//   (first tile only)  [H] x -> U, bias table      | BX
//   [C] L1 x GEMM (U)              [H] y loads in flight, stash x          | BL1X      (y: only models with labels)
//   [C] weight prefetch            [H] y -> U                              | BY
//   [C] L1 y GEMM (U), h1 -> Ha    [H] stash y                             | BH1
//   [C] L2 (Ha), h2 -> Hb          [H] stash h1 (Ha)                       | BH2
//   [C] heads (Hb; wave 0), z -> Zb; dec-L1 label GEMM (U; all waves)   [H] stash h2 (Hb)   | BZ
//   [C] dec L1 z (Zb), d1 -> Ha    [H] stash z (Zb)                        | BD1
//   [C] dec L2 (Ha), d2 -> Hb      [H] stash d1 (Ha)                       | BD2
//   [C] output layer (Hb), loss, da -> U          [H] stash d2 (Hb)        | BDA
//       (train step under bf16x3: four rounds [C] GEMM of a tile, a -> U raw | RBi | [H] loss terms of the round's tiles, da -> U)
//   [C] bwd out (U), dpre_d2 -> Hb (over d2)      [H] stash da (U), next gather table     | BDD2
//   [C] bwd d2 (Hb), dpre_d1 -> Ha (over d1)      [H] stash dpre_d2 (Hb), next x loads issued | BDD1
//   [C] bwd z (Ha), dmu|dlv -> Zb  [H] stash dpre_d1 (Ha)                  | BDML
//   [C] bwd heads (Zb), dpre_h2 -> Hb  [H] stash dmu|dlv (Zb)              | BDH2
//   [C] bwd h2 (Hb), dpre_h1 -> Ha [H] stash dpre_h2 (Hb)                  | BDH1
//   [C] loss sums -> red           [H] stash dpre_h1 (Ha), next x -> U     | BRED   (= BX of the next tile)
#include <math.h>
#include <stdlib.h>
#include "fused_tiles.hpp"
#include "rows_common.hpp"
#include "wstream.hpp"
#include "apply_common.hpp"
#include "../../include/dvae_train.h"

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

// MODE (RowsArgs::mode, compile time so the train-step instantiation carries none of the other modes' code or registers):
// 0 fused train step, 1 forward outputs only, 2 backward from upstream gradients
// DEFER: the deferred optimizer step (apply_common.hpp) -- the previous step's Adam update on the chain waves at the top of the launch,
// an arrival barrier in front of the first weight / bias load (which are sc1 loads then), the loss scalars by the last workgroup at the end
template <typename P, int YP, bool YENC, int MODE, bool INFO = false, bool DEFER = false>
__global__ __launch_bounds__(512) void vae_rows2_kernel(const RowsArgs g) {
    static_assert(!INFO || (MODE == 0 && YP == 16 && !YENC), "M2_info: train step, 1-dim label, encoder on x only");
    static_assert(!DEFER || (MODE == 0 && !INFO && sizeof(typename P::T) == 2), "deferred optimizer step: M1 / M2 train step, bf16 copies");
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    constexpr int E = P::E, KS = P::KSTEP, NP = P::NP;
    constexpr int LDU = Ld<T>::u, LDH = Ld<T>::hh, LDZ = Ld<T>::z;
    constexpr bool Y513 = (YP == XP);
    // train step with two operand planes: the loss epilogue of the output layer runs on the helper waves (see put_raw4), which takes
    // 64 registers (x prefetch, terms) off the chain waves' peak and pays for a deeper weight ring (PDO): the GEMM phases are
    // latency-bound by that depth (k-step time = t0 + L / D: 207 / 132 / 100 ns at D = 2 / 4 / 6)
#ifndef R2_OFFL
#define R2_OFFL 1
#endif
#ifndef R2_LATE_Y
#define R2_LATE_Y 1
#endif
#ifndef R2_EARLY_Y
#define R2_EARLY_Y 1
#endif
#ifndef R2_DH
#define R2_DH 6
#endif
#ifndef R2_HPRIO
#define R2_HPRIO 0
#endif
#ifndef R2_YSPREAD
#define R2_YSPREAD 8
#endif
#ifndef R2_YLOSEG
#define R2_YLOSEG 0
#endif
#ifndef R2_STAGGER
#define R2_STAGGER 0
#endif
    constexpr bool OFFL = R2_OFFL && MODE == 0 && NP == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const U = reinterpret_cast<T*>(smem);
    T* const Ha = U + TB * LDU;
    T* const Hb = Ha + TB * LDH;
    T* const Zb = Hb + TB * LDH;
    typedef Lds2<P, INFO> LD2;
    float* const Bias = reinterpret_cast<float*>(smem + LD2::o_bias);
    float* const red = reinterpret_cast<float*>(smem + LD2::o_red);
    int* const flags = reinterpret_cast<int*>(smem + LD2::o_flags);          // [0]: the label tile has a non-zero lo plane
    int64_t* const rowsrc = reinterpret_cast<int64_t*>(smem + LD2::o_rows);  // [2][TB] gather table, double buffered
    // M2_info (DeepGenerativeModel_v5, models.py:390-444; loop body scripts/training_M2_info_vad.py:159-198): the classifier on x runs on
    // the HELPER waves beside the chain's encoder (its layer images live in columns 32 .. 415 of U, which are free between the x image
    // and the output layer when the label is one column wide); the auxiliary net on z is four extra chain phases after the heads.
    float* const Binfo = reinterpret_cast<float*>(smem + LD2::o_info);
    float* const red2 = reinterpret_cast<float*>(smem + LD2::o_red2);
    float* const dzs = reinterpret_cast<float*>(smem + LD2::o_dzu);
    constexpr int OBC1 = 0, OBC2 = HD, OWC3 = 2 * HD, OBA1 = 3 * HD, OBA2 = 4 * HD, OWA3 = 5 * HD, OS3 = 6 * HD;
    constexpr int IMA = 32, IMB = 160, IMC = 288;                            // classifier layer images: first column in U
    // Two waves per SIMD leave 256 registers per wave: the tanh outputs the backward pass needs again do not stay in
    // registers.  h1 / h2 (needed ten phases later) wait in private fp32 LDS slots; d1 / d2 are re-read from their own
    // operand planes (hi + lo), which the backward tiles then overwrite in place (same lane, same elements).
    float* const keep = reinterpret_cast<float*>(smem + LD2::o_keep);
    float* const keepz = reinterpret_cast<float*>(smem + LD2::o_keepz);
    constexpr int OB1 = 0, OB2 = HD, OBMV = 2 * HD, OB3 = 2 * HD + 32, OB4 = 3 * HD + 32, OB5 = 4 * HD + 32;

    constexpr int DH = R2_DH;                                                    // ring depth of the helper waves' own weight stream
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const bool gather = g.rows != nullptr;
#ifndef R2_XCDMAP
#define R2_XCDMAP 0
#endif
    // first tile of this workgroup.  R2_XCDMAP (diagnostic): workgroup b runs on XCD b % 8; with the map, XCD x takes the CONSECUTIVE
    // tiles [x * grid / 8, (x + 1) * grid / 8) of a round, i.e. the frames of one slice of the weight-gradient kernel (which keeps slice
    // s on XCD s % 8), so that a stash line is written and read through the same L2
    int tile0 = (int)blockIdx.x;
#if R2_XCDMAP
    if ((gridDim.x & 7) == 0 && (g.ntiles % (int)gridDim.x) == 0) tile0 = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
#endif
    const int ntl = (g.ntiles - tile0 + (int)gridDim.x - 1) / (int)gridDim.x;     // tiles of this workgroup (>= 1: grid <= ntiles)

    if (wave_u < 4) {
        // =========================================================== chain waves ===========================================================
        const int cw = wave_u, fb = 32 * cw;
        constexpr int D = OFFL ? P::PDO : P::PD;
        typedef Sched<P, YP, YENC, D, INFO> SC;
        typedef HSched<P, YP, YENC, DH, INFO> HS;
#ifndef R2_DEFER_WAUX
#define R2_DEFER_WAUX 0
#endif
        typedef WStream<P, SC, D, DEFER ? R2_DEFER_WAUX : R2_WAUX> WS;
        constexpr unsigned FBB = SC::FBB;
        WS ws;
        ws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
        ws.voff = lane * 16;
        ws.pl = g.wpl_bytes;
        {
            auto mo = [&](const void* Wp) { return (unsigned)((const char*)Wp - (const char*)g.wcopy); };
#ifdef R2_SAMEW
            const unsigned tw = 0;                                            // diagnostic: every chain wave streams the SAME fragments (3 of 4 loads hit L1)
#else
            const unsigned tw = (unsigned)cw * FBB;                          // this wave's row tile in the 4-tile matrices
#endif
            constexpr unsigned KB1 = (XP / KS) * 4 * FBB, KB3 = (ZD / KS) * 4 * FBB;   // label k-blocks behind the x / z blocks of W1 / W3
            ws.sb[G_W1X] = mo(g.W1s) + tw;  ws.sb[G_W1Y] = mo(g.W1s) + tw + KB1;
            ws.sb[G_W2] = mo(g.W2s) + tw;   ws.sb[G_WMV] = mo(g.Wmvs);
            ws.sb[G_W3Y] = mo(g.W3s) + tw + KB3;  ws.sb[G_W3Z] = mo(g.W3s) + tw;
            ws.sb[G_W4] = mo(g.W4s) + tw;
            ws.sb[G_W5A] = mo(g.W5s) + (unsigned)cw * FBB;        ws.sb[G_W5B] = mo(g.W5s) + (unsigned)(cw + 4) * FBB;
            ws.sb[G_W5C] = mo(g.W5s) + (unsigned)(cw + 8) * FBB;  ws.sb[G_W5D] = mo(g.W5s) + (unsigned)(cw + 12) * FBB;
            ws.sb[G_W5T] = mo(g.W5t) + tw;  ws.sb[G_W4T] = mo(g.W4t) + tw;  ws.sb[G_W3ZT] = mo(g.W3zt);
            ws.sb[G_WMVT] = mo(g.Wmvt) + tw;  ws.sb[G_W2T] = mo(g.W2t) + tw;  ws.sb[G_PAD] = mo(g.W1s);
            if constexpr (INFO) {
                ws.sb[G_A1] = mo(g.Wa1s) + tw;  ws.sb[G_A2] = mo(g.Wa2s) + tw;  ws.sb[G_A2T] = mo(g.Wa2t) + tw;  ws.sb[G_A1T] = mo(g.Wa1t);
            } else { ws.sb[G_A1] = ws.sb[G_A2] = ws.sb[G_A2T] = ws.sb[G_A1T] = 0; }
        }
        if constexpr (DEFER) {
            // ---- the previous step's optimizer update (apply_common.hpp): this wave's tasks, then the arrival of every wave of the grid
            if (g.defer.have) {
                T* const tile = reinterpret_cast<T*>(keep) + cw * DeferLds<T, NP>::wave_elems;      // `keep` is free until the first epilogue
                if (!(g.defer.diag & 1))
                for (int un = cw * (int)gridDim.x + (int)blockIdx.x; un < 4 * g.defer.ntasks; un += 4 * (int)gridDim.x)      // units: consecutive ones on different CUs
                    defer_unit<T, NP>(g.defer.a, g.defer.tasks[un >> 2], un & 3, tile, lane, g.defer.diag);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's write-through stores have left
                if (lane == 0)
                    __hip_atomic_fetch_add(g.defer.shard + 32 * (((int)blockIdx.x * 4 + cw) & (DEFER_SHARDS - 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cw == 0 && !(g.defer.diag & 2)) {
                    // waves (b, w) with (4 b + w) mod 32 == s add to shard s: its value after this launch is seq_arrive x that count
                    const int nw = 4 * (int)gridDim.x, sh = lane & (DEFER_SHARDS - 1);
                    const unsigned want = g.defer.seq_arrive * (unsigned)((nw - sh + DEFER_SHARDS - 1) / DEFER_SHARDS);
                    const unsigned long long deadline = wall_clock64() + g.defer.timeout_ticks;
                    bool ok = true;
                    for (unsigned it_ = 0;; ++it_) {
                        const unsigned have = __hip_atomic_load(g.defer.shard + 32 * sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__ballot((int)(have - want) < 0) == 0ull) break;
                        if ((it_ & 15u) == 0u && wall_clock64() >= deadline) { ok = false; break; }      // (the clock read is a scalar memory operation: not every lap)
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ok && lane == 0) __hip_atomic_store(g.defer.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            wg_barrier();                                                   // BARR: no weight fragment, no bias value is requested before this
        }
        ws.fill();                                                         // the first D k-steps of the stream, in flight under the x tile load
        const T* const Ur = U + l31 * LDU + h * E;
        const T* const Har = Ha + l31 * LDH + h * E;
        const T* const Hbr = Hb + l31 * LDH + h * E;
        const T* const Zbr = Zb + l31 * LDZ + h * E;
        const bool w0 = cw == 0;
        constexpr int mode = MODE;
        double tot_rec = 0.0, tot_kl = 0.0, tot_bc = 0.0, tot_ba = 0.0;

        for (int it = 0; it < ntl; ++it) {
            const int tile = tile0 + it * (int)gridDim.x;
            const int64_t b0 = (int64_t)tile * TB;
            const bool live = (b0 + l31) < g.B;
            const int64_t* const rsrc = rowsrc + (it & 1) * TB;
            float rec_lane = 0.f, kl_lane = 0.f;
            R2_STAMP(0);
            if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + 30] = clock64();
            // reparametrisation noise of this lane's frame (wave 0 owns the latent tile)
            float ep_r[8];
            if (w0) {
                int64_t br = b0 + l31; br = br < g.B ? br : g.B - 1;
                if (g.eps != nullptr) {
                    const f32x4 e0 = *reinterpret_cast<const f32x4*>(g.eps + br * ZD + 4 * h);
                    const f32x4 e1 = *reinterpret_cast<const f32x4*>(g.eps + br * ZD + 8 + 4 * h);
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) { ep_r[jq] = live ? e0[jq] : 0.f; ep_r[4 + jq] = live ? e1[jq] : 0.f; }
                } else {
                    frame_noise8(g.rng_seed, (unsigned long long)br, g.rng_step, h, ep_r);
#pragma unroll
                    for (int jq = 0; jq < 8; ++jq) ep_r[jq] = live ? ep_r[jq] : 0.f;
                }
            }
            if (it == 0) {
                if (gather) wg_barrier();                               // BROW
                wg_barrier();                                           // BX
            }
            R2_STAMP(1);
#if R2_STAGGER
            // diagnostic: workgroups of one XCD (b, b + 8, b + 16, ...) start their GEMM chain R2_STAGGER x 64 clocks apart (is the k-step
            // slowed by every CU of an XCD asking its L2 for the same weight lines at the same moment?)
            if (it == 0) { for (int q = 0; q < (int)((blockIdx.x >> 3) & 7); ++q) __builtin_amdgcn_s_sleep(R2_STAGGER); }
#endif
            // ---------------- encoder layer 1: [x | y] -> h1 ----------------
            f32x16 acc;
            zero_acc<P>(acc);
            constexpr bool XF = P::XF16 && NP == 2;                        // the x block of layer 1 in split fp16 (fused_tiles.hpp: struct X16)
            gemm_seg<P, SC, D, G_W1X, WS, XF>(acc, ws, Ur);
            if constexpr (XF) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] *= X16::ACC;            // x * 2^-3 and W * 2^6 (exact power-of-two scales)
            }
            R2_STAMP(2);
            bool ylo = false;
            if (YP > 0) {
                wg_barrier();                                           // BL1X: the x image of U has been consumed
                wg_barrier();                                           // BY: the y image is in U (it stays there until the loss epilogue writes da)
                ylo = NP == 2 && __builtin_amdgcn_readfirstlane(flags[0]) != 0;
                // (two copies of the segment instead of a branch around the hi * lo MFMA of every k-step)
                if (!R2_YLOSEG) gemm_seg<P, SC, D, G_W1Y>(acc, ws, Ur, ylo);
                else if (ylo) gemm_seg<P, SC, D, G_W1Y>(acc, ws, Ur, true);
                else gemm_seg<P, SC, D, G_W1Y>(acc, ws, Ur, false);
            } else if constexpr (HS::n(H_W1X) > 0) wg_barrier();        // BL1X: the helpers' share of the x block is in `keep`
            if constexpr (HS::n(H_W1X) > 0) {                           // the partner wave's partial tile (k-steps NX1 .. of the x block)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += keep[r * 256 + tid];
            }
            float hv[16], bv[16];
            R2_STAMP(3);
            bias16(Bias + OB1, fb, h, bv);
#pragma unroll
            for (int r = 0; r < 16; ++r) { hv[r] = P::tanh_(acc[r] + bv[r]); keep[r * 256 + tid] = hv[r]; }
            put_lds<P>(hv, Ha, LDH, fb, l31, h);
            wg_barrier();                                               // BH1
            R2_STAMP(4);
            // ---------------- encoder layer 2 ----------------
            zero_acc<P>(acc);
            gemm_seg<P, SC, D, G_W2>(acc, ws, Har);
            bias16(Bias + OB2, fb, h, bv);
#pragma unroll
            for (int r = 0; r < 16; ++r) { hv[r] = P::tanh_(acc[r] + bv[r]); keep[(16 + r) * 256 + tid] = hv[r]; }
            put_lds<P>(hv, Hb, LDH, fb, l31, h);
            wg_barrier();                                               // BH2
            R2_STAMP(5);
            // ---------------- heads + reparametrisation (wave 0): rows 0-15 mu, 16-31 log_var ----------------
            zero_acc<P>(acc);
            gemm_seg<P, SC, D, G_WMV>(acc, ws, Hbr, true, w0);
            if (w0) {
                float zv[16];
                bias16(Bias + OBMV, 0, h, bv);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float mu = acc[r] + bv[r], lv = acc[r + 8] + bv[r + 8];
                    keepz[r * 64 + lane] = mu; keepz[(8 + r) * 64 + lane] = lv;
                    zv[r] = fmaf(P::exp_(0.5f * lv), ep_r[r], mu);     // models.py:17, 20
                    zv[r + 8] = 0.f;
                    if (live) kl_lane += lv - mu * mu - P::exp_(lv);   // utils.py:75
                }
                put_lds<P>(zv, Zb, LDZ, 0, l31, h);
                if (mode == 1 && live) {                                  // feature (r & 3) + 8 (r >> 2) + 4 h of this lane's frame
                    const int64_t o = (b0 + l31) * ZD + 4 * h;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        *reinterpret_cast<f32x4*>(g.out_mu + o + 8 * q) = f32x4{acc[4 * q] + bv[4 * q], acc[4 * q + 1] + bv[4 * q + 1], acc[4 * q + 2] + bv[4 * q + 2], acc[4 * q + 3] + bv[4 * q + 3]};
                        *reinterpret_cast<f32x4*>(g.out_lv + o + 8 * q) = f32x4{acc[8 + 4 * q] + bv[8 + 4 * q], acc[9 + 4 * q] + bv[9 + 4 * q], acc[10 + 4 * q] + bv[10 + 4 * q], acc[11 + 4 * q] + bv[11 + 4 * q]};
                        if (g.out_z) *reinterpret_cast<f32x4*>(g.out_z + o + 8 * q) = f32x4{zv[4 * q], zv[4 * q + 1], zv[4 * q + 2], zv[4 * q + 3]};
                    }
                }
            }
            // label block of decoder layer 1: independent of z (three of the four waves have nothing else to do in this phase)
            // (513-label models: on the helper waves, during the L1 y GEMM -- SC::HELPY; they finish decoder layer 1 themselves)
            f32x16 accy;
            if constexpr (YP > 0 && !SC::HELPY) {
                zero_acc<P>(accy);
                if (!R2_YLOSEG) gemm_seg<P, SC, D, G_W3Y>(accy, ws, Ur, ylo);
                else if (ylo) gemm_seg<P, SC, D, G_W3Y>(accy, ws, Ur, true);
                else gemm_seg<P, SC, D, G_W3Y>(accy, ws, Ur, false);
            }
            wg_barrier();                                               // BZ
            R2_STAMP(6);
            float bce_a = 0.f;
            if constexpr (INFO) {
                // ---------------- auxiliary classifier on z (models.py:41-63, 419-420): forward, BCE against the frame label, backward down to
                // d BCE / d z.  Unit scale on chip; the stashed pre-activation gradients carry (gamma - beta) (quirk Q4: the -beta * dBCE that
                // enc_loss.backward() leaves in the auxiliary net's .grad is never zeroed before aux_loss.backward() adds gamma * dBCE).
                float a1r[16], a2r[16], w3v[16];
                zero_acc<P>(acc);
                gemm_seg<P, SC, D, G_A1>(acc, ws, Zbr);
                bias16(Binfo + OBA1, fb, h, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) a1r[r] = fmaxf(acc[r] + bv[r], 0.f);
                put_lds<P>(a1r, Ha, LDH, fb, l31, h);
                wg_barrier();                                           // BA1
                zero_acc<P>(acc);
                gemm_seg<P, SC, D, G_A2>(acc, ws, Har);
                bias16(Binfo + OBA2, fb, h, bv);
                bias16(Binfo + OWA3, fb, h, w3v);
                float pd = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { a2r[r] = fmaxf(acc[r] + bv[r], 0.f); pd = fmaf(w3v[r], a2r[r], pd); }
                put_lds<P>(a2r, Hb, LDH, fb, l31, h);
                pd += __shfl_xor(pd, 32, 64);
                if (h == 0) red2[cw * 32 + l31] = pd;
                wg_barrier();                                           // BA2
                {
                    int64_t rowy = gather ? rsrc[l31] : (b0 + l31 < g.B ? b0 + l31 : g.B - 1);
                    const float y_l = g.y[rowy * g.ldy];
                    const float logit = red2[l31] + red2[32 + l31] + red2[64 + l31] + red2[96 + l31] + Binfo[OS3 + 1];
                    const float p = 1.f / (1.f + P::exp_(-logit));
                    const float lp = P::log_(p + g.elbo_eps), lq = P::log_(1.f - p + g.elbo_eps);
                    bce_a = (live && h == 0 && w0) ? -(y_l * lp + (1.f - y_l) * lq) : 0.f;                        // utils.py:55-56, this frame's term
                    const float u = live ? -g.invB * (y_l / (p + g.elbo_eps) - (1.f - y_l) / (1.f - p + g.elbo_eps)) : 0.f;
                    const float dpre3 = u * p * (1.f - p);
                    float dv2[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) dv2[r] = a2r[r] > 0.f ? w3v[r] * dpre3 : 0.f;                     // dpre2 (unit scale)
                    put_lds<P>(dv2, Ha, LDH, fb, l31, h);
                }
                wg_barrier();                                           // BA3
                zero_acc<P>(acc);
                gemm_seg<P, SC, D, G_A2T>(acc, ws, Har);
                {
                    float dv1[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) dv1[r] = a1r[r] > 0.f ? acc[r] : 0.f;                              // dpre1 (unit scale)
                    put_lds<P>(dv1, Hb, LDH, fb, l31, h);
                }
                wg_barrier();                                           // BA4
                zero_acc<P>(acc);
                gemm_seg<P, SC, D, G_A1T>(acc, ws, Hbr, true, w0);     // d BCE_aux / d z: the latent tile belongs to wave 0
                if (w0) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) dzs[r * 64 + lane] = acc[r];
                }
            }
            // ---------------- decoder layer 1: [z | y] -> d1 ----------------
            if constexpr (!SC::HELPY) {
                zero_acc<P>(acc);
                gemm_seg<P, SC, D, G_W3Z>(acc, ws, Zbr);
                bias16(Bias + OB3, fb, h, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) hv[r] = P::tanh_(acc[r] + ((YP > 0) ? accy[r] : 0.f) + bv[r]);
                put_lds<P>(hv, Ha, LDH, fb, l31, h);
            }
            wg_barrier();                                               // BD1
            R2_STAMP(7);
            // ---------------- decoder layer 2 ----------------
            zero_acc<P>(acc);
            gemm_seg<P, SC, D, G_W4>(acc, ws, Har);
            // loss epilogue input: x[frame][32 t + 8 gq + 4 h .. + 3] of this lane's frame straight from global memory
            // (the tile was read a few microseconds ago: L2 / MALL), one output tile ahead
            int64_t rowx;
            if (gather) rowx = rsrc[l31];
            else { rowx = b0 + l31; rowx = rowx < g.B ? rowx : g.B - 1; }
            int64_t rowb = b0 + l31; rowb = rowb < g.B ? rowb : g.B - 1;     // batch-order row (outputs, upstream gradients)
            const float* const xrow = mode == 2 ? (g.g_r ? g.g_r + rowb * g.ld_gr + 4 * h : nullptr) : g.x + rowx * g.ldx + 4 * h;
            float* const orow = g.out_r + rowb * g.ld_r + 4 * h;
            f32x4 xq[4], xn[4];
            auto xload = [&](int t, f32x4 (&q)[4]) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (mode == 1 || xrow == nullptr) q[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    else q[gq] = reinterpret_cast<const F4U*>(xrow + 32 * t + 8 * gq)->v;
                }
            };
            if constexpr (!OFFL) xload(cw, xq);
            // bin 512 (wave 3's dot-product tile), requested a phase early
            const float xv512 = (cw == 3 && mode != 1 && xrow != nullptr) ? (mode == 2 ? g.g_r[rowb * g.ld_gr + XD - 1] : g.x[rowx * g.ldx + XD - 1]) : 0.f;
            bias16(Bias + OB4, fb, h, bv);
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = P::tanh_(acc[r] + bv[r]);
            put_lds<P>(hv, Hb, LDH, fb, l31, h);
            wg_barrier();                                               // BD2
            R2_STAMP(8);
            // ---------------- output layer a = W5 d2 + b5, Itakura-Saito terms, da -> U ----------------
            const float invB_l = live ? g.invB : 0.f;                      // frames past B contribute nothing
            // 16 full tiles, 4 per chain wave (tile cw + 4 i = stream segment G_W5A + i); the 17th tile holds ONE real
            // feature (bin 512): chain wave 3 does it as a 128-term dot product
            if constexpr (OFFL) {
                static_for<0, 4>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    const int t = cw + 4 * I;
                    zero_acc<P>(acc);
                    R2_FSTAMP(16 + 3 * I);
                    gemm_seg<P, SC, D, G_W5A + I>(acc, ws, Hbr);
                    R2_FSTAMP(17 + 3 * I);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 b5q = *reinterpret_cast<const f32x4*>(Bias + OB5 + 32 * t + 8 * gq + 4 * h);
                        const float a4[4] = {acc[4 * gq] + b5q[0], acc[4 * gq + 1] + b5q[1], acc[4 * gq + 2] + b5q[2], acc[4 * gq + 3] + b5q[3]};
                        put_raw4<P>(a4, U, LDU, 32 * t + 8 * gq + 4 * h, l31);
                    }
                    R2_FSTAMP(18 + 3 * I);
                    wg_barrier();                                       // RB0 .. RB3: the round's four tiles go to the helpers
                });
            } else
            static_for<0, 4>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                const int t = cw + 4 * I;
                zero_acc<P>(acc);
                if constexpr (I < 3) xload(t + 4, xn);
                gemm_seg<P, SC, D, G_W5A + I>(acc, ws, Hbr);
                float da[16], b5v[16];
                bias16(Bias + OB5, 32 * t, h, b5v);
                if (mode == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float xs = xq[r >> 2][r & 3];
                        const float a = acc[r] + b5v[r];
                        const float xe = xs * P::exp_(-a);                   // x / r,  r = exp(a)  (models.py:122)
                        rec_lane += xe - P::log_(xs + g.elbo_eps) + a - 1.f;   // utils.py:74 (log r = a)
                        da[r] = (1.f - xe) * invB_l;                         // d recon / d a
                    }
                    put_lds<P>(da, U, LDU, 32 * t, l31, h);
                } else if (mode == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) da[r] = live ? xq[r >> 2][r & 3] * P::exp_(acc[r] + b5v[r]) : 0.f;   // d a = (d L / d r) r
                    put_lds<P>(da, U, LDU, 32 * t, l31, h);
                } else if (live) {                                       // mode 1: the reconstruction itself
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        F4U o;
                        o.v = f32x4{P::exp_(acc[4 * gq] + b5v[4 * gq]), P::exp_(acc[4 * gq + 1] + b5v[4 * gq + 1]), P::exp_(acc[4 * gq + 2] + b5v[4 * gq + 2]), P::exp_(acc[4 * gq + 3] + b5v[4 * gq + 3])};
                        *reinterpret_cast<F4U*>(orow + 32 * t + 8 * gq) = o;
                    }
                }
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) xq[gq] = xn[gq];
            });
            if (cw == 3) {
                const float* wl = Bias + OB5 + NO + 64 * h;                 // this half's 64 weights (LDS broadcast reads)
                const T* drow = Hb + l31 * LDH + 64 * h;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 64 / E; ++c) {
                    Frag dv[NP];
                    bloadp<P>(dv, drow + c * E);
#pragma unroll
                    for (int j = 0; j < E; j += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + c * E + j);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            float dd = (float)dv[0][j + jj];
                            if constexpr (NP == 2) dd += (float)dv[1][j + jj];
                            s = fmaf(dd, wv[jj], s);
                        }
                    }
                }
                s += __shfl_xor(s, 32, 64);
                const float a = s + Bias[OB5 + XD - 1];
                const float xe = xv512 * P::exp_(-a);
                if (h == 0 && mode == 0) rec_lane += xe - P::log_(xv512 + g.elbo_eps) + a - 1.f;
                if (mode == 1 && live && h == 0) g.out_r[rowb * g.ld_r + XD - 1] = P::exp_(a);
                const float da512 = mode == 0 ? (1.f - xe) * invB_l : (mode == 2 && live ? xv512 * P::exp_(a) : 0.f);
                // columns 512 .. 543 of this frame's da row: the value, then 31 zeros (16 per lane half)
                float dz16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dz16[r] = 0.f;
                if (h == 0) dz16[0] = da512;
                T* const urow = U + l31 * LDU + (XD - 1) + 16 * h;
#pragma unroll
                for (int c = 0; c < 16 / E; ++c) {
                    Frag fh, fl;
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        fh[j] = P::cvt(dz16[c * E + j]);
                        fl[j] = P::cvt(dz16[c * E + j] - (float)fh[j]);
                    }
                    *reinterpret_cast<Frag*>(urow + c * E) = fh;
                    if constexpr (NP == 2) *reinterpret_cast<Frag*>(urow + Pl<P>::lds + c * E) = fl;
                }
            }
            R2_FSTAMP(28);
            wg_barrier();                                               // BDA
            R2_STAMP(9);
            if (mode == 1) {                                               // forward only: the helpers stage the next tile, then the stream restarts at position 0
                wg_barrier();                                           // BDD2'
                ws.fill();
                wg_barrier();                                           // BRED'
                continue;
            }
            // ---------------- backward: d2 <- da ----------------
            zero_acc<P>(acc);
            gemm_seg<P, SC, D, G_W5T>(acc, ws, Ur);
            float dv[16];
            get_lds<P>(hv, Hb, LDH, fb, l31, h);                          // d2 of this lane's elements (every reader of Hb has passed BDA)
#pragma unroll
            for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - hv[r] * hv[r]);
            put_lds<P>(dv, Hb, LDH, fb, l31, h);                          // in place
            wg_barrier();                                               // BDD2
            R2_STAMP(10);
            // ---------------- backward: d1 <- dpre_d2 ----------------
            zero_acc<P>(acc);
            gemm_seg<P, SC, D, G_W4T>(acc, ws, Hbr);
            get_lds<P>(hv, Ha, LDH, fb, l31, h);                          // d1: Ha has not been written since decoder layer 1
#pragma unroll
            for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - hv[r] * hv[r]);
            put_lds<P>(dv, Ha, LDH, fb, l31, h);                          // in place
            wg_barrier();                                               // BDD1
            R2_STAMP(11);
            // ---------------- backward: z <- dpre_d1 (wave 0), then dmu / dlogvar ----------------
            zero_acc<P>(acc);
            float gu[24];                                                  // mode 2: upstream d z | d mu | d log_var of this lane's 8 latent features
#pragma unroll
            for (int r = 0; r < 24; ++r) gu[r] = 0.f;
            if (w0 && mode == 2) {
                const int64_t o = (b0 + l31 < g.B ? b0 + l31 : g.B - 1) * ZD + 4 * h;
                const float* srcs[3] = {g.g_z, g.g_mu, g.g_lv};
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (srcs[q] != nullptr) {
                        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(srcs[q] + o), hi4 = *reinterpret_cast<const f32x4*>(srcs[q] + o + 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { gu[8 * q + j] = lo4[j]; gu[8 * q + 4 + j] = hi4[j]; }
                    }
            }
            gemm_seg<P, SC, D, G_W3ZT>(acc, ws, Har, true, w0);
            if (w0) {
                float dml[16];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float mu = keepz[r * 64 + lane], lv = keepz[(8 + r) * 64 + lane];
                    float dz = acc[r] + gu[r];
                    if constexpr (INFO) dz -= g.beta * dzs[r * 64 + lane];                                          // enc_loss = ELBO + alpha clf - beta BCE(aux(z), y)
                    const float kmu = mode == 0 ? mu * g.invB : gu[8 + r];                                         // KL term of the fused step / upstream d mu
                    const float klv = mode == 0 ? -0.5f * g.invB * (1.f - P::exp_(lv)) : gu[16 + r];
                    dml[r] = live ? dz + kmu : 0.f;                                                                 // dmu
                    dml[r + 8] = live ? dz * ep_r[r] * (0.5f * P::exp_(0.5f * lv)) + klv : 0.f;                     // dlogvar
                }
                put_lds<P>(dml, Zb, LDZ, 0, l31, h);
            }
            wg_barrier();                                               // BDML
            R2_STAMP(12);
            // ---------------- backward: h2 <- [dmu | dlogvar] ----------------
            zero_acc<P>(acc);
            gemm_seg<P, SC, D, G_WMVT>(acc, ws, Zbr);
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float hk = keep[(16 + r) * 256 + tid]; dv[r] = acc[r] * (1.f - hk * hk); }
            put_lds<P>(dv, Hb, LDH, fb, l31, h);
            wg_barrier();                                               // BDH2
            R2_STAMP(13);
            // ---------------- backward: h1 <- dpre_h2 (inputs are data: stop here) ----------------
            zero_acc<P>(acc);
            gemm_seg<P, SC, D, G_W2T>(acc, ws, Hbr);
            // the padding positions of the schedule (none for most shapes) keep the ring phase tile-invariant
            { f32x16 dummy; zero_acc<P>(dummy); gemm_seg<P, SC, D, G_PAD>(dummy, ws, Hbr, true, false); }
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float hk = keep[r * 256 + tid]; dv[r] = acc[r] * (1.f - hk * hk); }
            put_lds<P>(dv, Ha, LDH, fb, l31, h);
            wg_barrier();                                               // BDH1
            R2_STAMP(14);
            // ---------------- per-tile loss sums ----------------
            if (!live) rec_lane = 0.f;
            const float rs = wave_sum(rec_lane), ks = wave_sum(kl_lane);
            if (lane == 0) { red[cw] = rs; red[4 + cw] = ks; }
            if constexpr (INFO) { if (w0) { const float bas = wave_sum(bce_a); if (lane == 0) red2[129] = bas; } }
            wg_barrier();                                               // BRED (the next tile's x image is in U)
            if (tid == 0) {
                if constexpr (INFO) { tot_bc += (double)red2[128]; tot_ba += (double)red2[129]; }
                tot_rec += (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
                if constexpr (OFFL)                                        // the helpers' share: sum (x / r + a - 1), and ln 2 * sum log2(x + eps)
                    tot_rec += (double)red[8] + (double)red[9] + (double)red[10] + (double)red[11]
                             - 0.6931471805599453 * ((double)red[12] + (double)red[13] + (double)red[14] + (double)red[15]);
                tot_kl += -0.5 * (double)red[4];
            }
        }
        R2_STAMP(15);
        if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + 31] = clock64();
        if (tid == 0) {
            g.partials[4 * blockIdx.x] = tot_rec;
            g.partials[4 * blockIdx.x + 1] = tot_kl;
            g.partials[4 * blockIdx.x + 2] = tot_bc;
            g.partials[4 * blockIdx.x + 3] = tot_ba;
        }
    } else {
        // =========================================================== helper waves ===========================================================
        const int hw = wave_u - 4, ht = tid - 256;
        constexpr bool XFH = P::XF16 && NP == 2;                           // the x image in split fp16 (fused_tiles.hpp: struct X16)
        // (An L2 warm-up of the weight copies by the helpers -- one dword per 128-byte line, 1/32 of the buffer per workgroup of an
        // XCD -- changed nothing: the weight stream runs at the ~34 B/clk/CU of an L2-resident table shared by every CU, not at miss latency.)
        // fp32 bias table -> LDS once; the loads are issued here (clamped addresses instead of branches)
        constexpr int NBT = Ld<T>::nbias;
        constexpr int NB = (NBT + 255) / 256;
        float bvv[NB];
        auto load_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                int i = ht + 256 * q;
                i = i < NBT ? i : NBT - 1;
                const float* src;
                int k;
                if (i < OB2) { src = g.b1; k = i; }
                else if (i < OBMV) { src = g.b2; k = i - OB2; }
                else if (i < OBMV + ZD) { src = g.bmu; k = i - OBMV; }
                else if (i < OB3) { src = g.blv; k = i - OBMV - ZD; }
                else if (i < OB4) { src = g.b3; k = i - OB3; }
                else if (i < OB5) { src = g.b4; k = i - OB4; }
                else if (i < OB5 + NO) { src = g.b5; k = i - OB5; k = k < XD ? k : XD - 1; }
                else { src = g.w5last; k = i - OB5 - NO; }
                // deferred step: the values were written by other workgroups of THIS launch (write-through stores): sc1 loads, after BARR
                if constexpr (DEFER) bvv[q] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else bvv[q] = src[k];
            }
        };
        auto store_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int i = ht + 256 * q;
                if (i < NBT) Bias[i] = (i >= OB5 + XD && i < OB5 + NO) ? 0.f : bvv[q];
            }
        };
        if constexpr (!DEFER) load_bias();
        // gather table of tile `tl_` into half `half` (threads 0..31 of the helper group); bad indices are clamped and counted
        auto fill_rows = [&](int tl_, int half) {
            if (ht < TB) {
                const int64_t bf = (int64_t)tl_ * TB + ht;
                const int64_t br = bf < g.B ? bf : g.B - 1;                 // frames past the batch repeat its last row (masked out later)
                int64_t r = g.rows[br];
                if (r < 0 || r >= g.n_rows) { r = 0; if (g.bad_rows && bf < g.B) atomicAdd(g.bad_rows, 1); }
                rowsrc[half * TB + ht] = r;
            }
        };
        // the helper waves' own weight stream (wstream.hpp: HSched): the rest of the x block of encoder layer 1 and, for 513-label
        // models, decoder layer 1.  Row tile of helper wave hw = row tile of its partner chain wave.
        typedef Sched<P, YP, YENC, (OFFL ? P::PDO : P::PD), INFO> SCc;
        typedef HSched<P, YP, YENC, DH, INFO> HS;
        typedef WStream<P, HS, DH> HWS;
        HWS hws;
        if constexpr (HS::any) {
            hws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
            hws.voff = lane * 16;
            hws.pl = g.wpl_bytes;
            auto mo = [&](const void* Wp) { return (unsigned)((const char*)Wp - (const char*)g.wcopy); };
            constexpr unsigned FBB = HS::FBB;
            const unsigned tw = (unsigned)hw * FBB;
            hws.sb[H_W1X] = mo(g.W1s) + tw + (unsigned)SCc::NX1 * 4u * FBB;
            hws.sb[H_W3Y] = mo(g.W3s) + tw + (unsigned)(ZD / KS) * 4u * FBB;
            hws.sb[H_W3Z] = mo(g.W3s) + tw;
            hws.sb[H_PAD] = mo(g.W1s);
            if constexpr (INFO) { hws.sb[H_C1] = mo(g.Wc1s) + tw;  hws.sb[H_C2] = mo(g.Wc2s) + tw;  hws.sb[H_C2T] = mo(g.Wc2t) + tw; }
            else { hws.sb[H_C1] = hws.sb[H_C2] = hws.sb[H_C2T] = 0; }
            if constexpr (INFO) hws.fill();                               // the classifier's first fragments: requested BEFORE the x tile (vmcnt retires in order)
        }
        const T* const Urh = U + l31 * LDU + h * E;
        const T* const Zbrh = Zb + l31 * LDZ + h * E;
        float lsum2 = 0.f;            // sum of log2(x + eps) over this thread's share of the x tiles it committed (loss epilogue, OFFL)
        f32x4 xv[NQ513], yv[NQ513];
        bool x_in_regs = false;       // the dense fast path holds the tile in registers between issue and commit
        bool y_in_regs = false;       // persistent loop: the NEXT tile's label tile is requested during this tile's backward phases too
        // forward-only launches stash nothing; stash_inputs: bit 0 = the x tile, bit 1 = the label tile (a cleared bit: the weight-gradient
        // kernel reads that input from its fp32 matrix)
        const bool st1 = MODE != 1 && !(g.ablate & 1), st2 = MODE != 1 && !(g.ablate & 2) && (g.stash_inputs & 2), st2x = MODE != 1 && !(g.ablate & 2) && (g.stash_inputs & 1);
        for (int it = 0; it < ntl; ++it) {
            const int tile = tile0 + it * (int)gridDim.x;
            const int64_t b0 = (int64_t)tile * TB;
            const bool full = (b0 + TB) <= g.B;
            const int64_t* const rsrc = rowsrc + (it & 1) * TB;
            auto rowof = [&](int r) -> int64_t {
                if (gather) return rsrc[r];
                const int64_t br = b0 + r;
                return br < g.B ? br : g.B - 1;
            };
            // per-iteration opaque copy of the thread id: keeps the per-thread staging addresses out of loop-invariant hoisting
            int tl = ht;
            asm volatile("" : "+v"(tl));
            const bool yfast = Y513 && g.fasty && full;
            bool y_early = it > 0 && y_in_regs;
            if (it == 0) {
                if (gather) { fill_rows(tile, 0); wg_barrier(); }       // BROW
                if (g.fastx && full) {
                    tile513_issue(g.x, rowof, xv, tl);
                    if constexpr (!DEFER) store_bias();
#if R2_EARLY_Y
                    if (YP > 0 && Y513 && g.fasty) {
                        // The label tile is requested BEHIND the x tile, a quarter at a time, each quarter followed by the commit of
                        // the x quarter that has arrived by then.  (A CU pulls ~30 GB/s from HBM while every CU does the same: the
                        // 17 label requests of a thread take ~2.5 us to ISSUE.  Issued in one go in front of the x commit they delay
                        // it by that much; issued after the BX barrier -- the previous form -- they arrive 3 us after the L1 x GEMM
                        // has finished.  Interleaved, x is committed as it lands and the labels land during the L1 x GEMM.)
                        unsigned long long lb = 0ull;
                        static_for<0, 4>([&](auto qc) {
                            constexpr int Q = decltype(qc)::value;
                            tile513_issue_part<4 * Q, 4 * Q + 4>(g.y, rowof, yv, tl);
                            __builtin_amdgcn_sched_barrier(0);
                            tile513_commit_part<P, 4 * Q, 4 * Q + 4, XFH>(xv, U, LDU, tl, nullptr, lb);
                            __builtin_amdgcn_sched_barrier(0);
                        });
                        tile513_issue_last(g.y, rowof, yv, tl);
                        __builtin_amdgcn_sched_barrier(0);
                        tile513_commit_last<P, XP, XFH>(xv, U, LDU, tl, nullptr, lb);
                        y_early = true;
                    } else
#endif
                    {
                        tile513_commit<P, XP, XFH>(xv, U, LDU, tl);
#pragma unroll
                        for (int i = 0; i < NQ513; ++i) yv[i] = f32x4{0.f, 0.f, 0.f, 0.f};     // a full definition on every path (see xv below)
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NQ513; ++i) { xv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; yv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    if constexpr (!DEFER) store_bias();
                    load_rows_to_lds<P, XFH>(g.x, g.ldx, XD, XP, b0, g.B, U, LDU, tl, rowof, nullptr, OFFL ? &lsum2 : nullptr, g.elbo_eps);
                }
                if constexpr (INFO) {
                    for (int i = ht; i < 6 * HD + 2; i += 256) {
                        const int q = i / HD, k = i - q * HD;
                        float v;
                        if (q == 0) v = g.bc1[k]; else if (q == 1) v = g.bc2[k]; else if (q == 2) v = g.wc3[k];
                        else if (q == 3) v = g.ba1[k]; else if (q == 4) v = g.ba2[k]; else if (q == 5) v = g.wa3[k];
                        else v = k == 0 ? g.bc3[0] : g.ba3[0];
                        Binfo[i] = v;
                    }
                }
                if (ht == 0) flags[0] = 0;
                if constexpr (DEFER) {
                    wg_barrier();                                       // BARR: the optimizer update of the previous step has arrived from every workgroup
                    load_bias();
                    store_bias();
                }
                wg_barrier();                                           // BX
                if constexpr (OFFL) { if (g.fastx && full) lsum2 += tile513_log2sum(xv, g.elbo_eps, tl); }   // the tile is still in registers
            }
            // ---- chain: L1 x GEMM (its NX1 k-steps); here: the other k-steps of the same row tile, partial tile -> keep
            if constexpr (HS::n(H_W1X) > 0) {
                f32x16 hacc;
                zero_acc<P>(hacc);
                hws.fill();
                gemm_seg<P, HS, DH, H_W1X, HWS, XFH>(hacc, hws, Urh + SCc::NX1 * 2 * E);
#pragma unroll
                for (int r = 0; r < 16; ++r) keep[r * 256 + ht] = XFH ? hacc[r] * X16::ACC : hacc[r];
            }
            // ---- M2_info: the classifier on x (models.py:41-63, 418), on the helper waves beside the chain's encoder.  Wave hw owns hidden
            // features 32 hw .. + 31 of both 128-wide layers; the layer images sit in columns IMA / IMB / IMC of U.
            float c1r[16], c2r[16], w3c[16];
            float bce_c = 0.f, ylab = 0.f;
            if constexpr (INFO) {
                ylab = g.y[rowof(l31) * g.ldy];
                if (it > 0) hws.fill();                                    // (first tile: requested in front of the x tile)
                f32x16 cacc;
                zero_acc<P>(cacc);
                gemm_seg<P, HS, DH, H_C1, HWS, XFH>(cacc, hws, Urh);
                float bvc[16];
                bias16(Binfo + OBC1, 32 * hw, h, bvc);
#pragma unroll
                for (int r = 0; r < 16; ++r) c1r[r] = fmaxf((XFH ? cacc[r] * X16::ACC : cacc[r]) + bvc[r], 0.f);
            }
            f32x16 accy;                                                   // label block of decoder layer 1 (HS::HELPY)
            int yplanes = NP;                                              // planes of the label stash this tile writes (RowsArgs::ylo_skip)
            int ydirty = 0;                                                // the tile slot's lo plane holds an earlier launch's non-zero values
            if constexpr (NP == 2 && YP > 0) { if (g.ylo_skip && st2) ydirty = g.ylo_dirty[tile]; }
            // ---- y loads in flight, x -> stash
            if (YP > 0) {
                if (Y513 && yfast && !y_early) tile513_issue(g.y, rowof, yv, tl);
                R2_HSTAMP(16);
                if (st2x && !XFH) stash_from_lds<P>(U, LDU, XP, NO, (T*)g.xT, g.spl, g.Bp, b0, tl);      // (split-fp16 image: the x stash is re-read column-wise after BH1 / BH2, see xstash_reload)
                R2_HSTAMP(17);
                wg_barrier();                                           // BL1X
                if constexpr (HS::HELPY && HS::n(H_W1X) == 0) hws.fill();  // decoder layer 1's first fragments arrive under the label commit
                R2_HSTAMP(18);
                bool any = false;                                          // does the label tile need its lo plane?  (found while committing)
                if (Y513 && yfast) tile513_commit<P, XP>(yv, U, LDU, tl, nullptr, NP == 2 ? &any : nullptr);
                else load_rows_to_lds<P>(g.y, g.ldy, g.ydim, YP, b0, g.B, U, LDU, tl, rowof);
                R2_HSTAMP(19);
                if constexpr (NP == 2) {
                    if (Y513 && yfast) {
                    } else {
                        constexpr int YPD = YP > 0 ? YP : 1;
                        for (int idx = tl; idx < TB * YP; idx += 256) any |= (float)U[Pl<P>::lds + (idx / YPD) * LDU + idx % YPD] != 0.f;
                    }
                    if (__ballot(any) != 0ull && lane == 0) atomicOr(&flags[0], 1);
                }
                R2_HSTAMP(20);
                if constexpr (INFO) put_lds<P>(c1r, U + IMA, LDU, 32 * hw, l31, h);      // the x image is dead (BL1X)
                wg_barrier();                                           // BY
                if constexpr (NP == 2) {
                    const bool ylo_t = __builtin_amdgcn_readfirstlane(flags[0]) != 0;
                    if (g.ylo_skip && st2) {
                        ydirty = __builtin_amdgcn_readfirstlane(ydirty);
                        yplanes = (ylo_t || ydirty != 0) ? 2 : 1;
                        if (ht == 0) {
                            if (ylo_t) *g.ylo_epoch = g.launch_id;                           // every flagged tile stores the same value
                            if ((ylo_t ? 1 : 0) != ydirty) g.ylo_dirty[tile] = ylo_t ? 1 : 0;
                        }
                    }
                }
#if !R2_LATE_Y
                if (st2) stash_from_lds<P>(U, LDU, YP, (YP + 31) / 32 * 32, (T*)g.yT, g.spl, g.Bp, b0, tl, 0, 1 << 30, yplanes);
#endif
                if constexpr (HS::HELPY) {
                    // chain: L1 y GEMM (33 k-steps).  Here: the label block of decoder layer 1 (33 k-steps, independent of z), this
                    // wave's row tile; it stays in registers until the z block joins it after BZ
                    const bool ylo = NP == 2 && __builtin_amdgcn_readfirstlane(flags[0]) != 0;
                    zero_acc<P>(accy);
                    if (!R2_YLOSEG) gemm_seg<P, HS, DH, H_W3Y>(accy, hws, Urh, ylo);
                    else if (ylo) gemm_seg<P, HS, DH, H_W3Y>(accy, hws, Urh, true);
                    else gemm_seg<P, HS, DH, H_W3Y>(accy, hws, Urh, false);
                }
                if constexpr (INFO) {                                   // classifier layer 2 + this wave's share of the output dot product
                    f32x16 a2c;
                    zero_acc<P>(a2c);
                    gemm_seg<P, HS, DH, H_C2>(a2c, hws, Urh + IMA);
                    float bvc[16];
                    bias16(Binfo + OBC2, 32 * hw, h, bvc);
                    bias16(Binfo + OWC3, 32 * hw, h, w3c);
                    float pd = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { c2r[r] = fmaxf(a2c[r] + bvc[r], 0.f); pd = fmaf(w3c[r], c2r[r], pd); }
                    put_lds<P>(c2r, U + IMB, LDU, 32 * hw, l31, h);
                    pd += __shfl_xor(pd, 32, 64);
                    if (h == 0) red2[hw * 32 + l31] = pd;
                }
            } else {
                if (st2x && !XFH) stash_from_lds<P>(U, LDU, XP, NO, (T*)g.xT, g.spl, g.Bp, b0, tl);      // (split-fp16 image: the x stash is re-read column-wise after BH1 / BH2, see xstash_reload)
                if constexpr (HS::n(H_W1X) > 0) wg_barrier();            // BL1X (models without labels): the partial tile is in `keep`
            }
            wg_barrier();                                               // BH1
            if constexpr (INFO) {
                // sigmoid output, binary_cross_entropy against the frame label (utils.py:55-56) and the unit-scale backward to dpre2; the
                // stashed pre-activation gradients carry alpha (classif_loss = alpha * BCE, training_M2_info_vad.py:165)
                const bool live_h = (b0 + l31) < g.B;
                const float logit = red2[l31] + red2[32 + l31] + red2[64 + l31] + red2[96 + l31] + Binfo[OS3];
                const float p = 1.f / (1.f + P::exp_(-logit));
                const float lp = P::log_(p + g.elbo_eps), lq = P::log_(1.f - p + g.elbo_eps);
                bce_c = (live_h && h == 0 && hw == 0) ? -(ylab * lp + (1.f - ylab) * lq) : 0.f;
                const float u = live_h ? -g.invB * (ylab / (p + g.elbo_eps) - (1.f - ylab) / (1.f - p + g.elbo_eps)) : 0.f;   // d BCE / d p
                const float dpre3 = u * p * (1.f - p);
                float dv2[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dv2[r] = c2r[r] > 0.f ? w3c[r] * dpre3 : 0.f;                 // dpre2 (unit scale)
                put_lds<P>(dv2, U + IMC, LDU, 32 * hw, l31, h);
                if (st1) {
                    stash_tile<P>(U + IMA, LDU, 32 * hw, (T*)g.c1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
                    stash_tile<P>(U + IMB, LDU, 32 * hw, (T*)g.c2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
                    if (hw == 0 && h == 0) {                               // output pre-activation gradient: feature row 0 of a 32-row stash tile
                        T* d3 = (T*)g.dc3T + (b0 / KS) * (64 * E) + (l31 / E) * 32 * E + (l31 % E);
                        const T d3h = P::cvt(dpre3 * g.alpha);
                        *d3 = d3h;
                        if constexpr (NP == 2) d3[g.spl] = P::cvt(dpre3 * g.alpha - (float)d3h);
                    }
                }
            }
            if (st1) stash_tile<P>(Ha, LDH, 32 * hw, (T*)g.h1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
#ifndef R2_XSP
#define R2_XSP 1      // placement of the x stash re-read (split-fp16 image): 0 = groups 0-3 after BH1, 4-8 after BH2; 1 = all after BH2 (the heads phase, 4.9 us: measured +0.2 us per step against the round-3 LDS transposition, tools/r04/exp_xsp.sh; 0: +1 .. 3, 2: +0.6); 2 = 0-4 after BH2, 5-8 after BZ
#endif
            // (M2_info: the helpers run the classifier beside the encoder in these phases; its x stash waits for the auxiliary-net phases)
            if constexpr (XFH && !INFO && R2_XSP == 0) { if (st2x) xstash_reload<P, 0, 4>(g.x, g.ldx, rowof, b0, g.B, (T*)g.xT, g.spl, g.Bp, tl); }
#if R2_LATE_Y
            // the label tile stays in U until the output layer: its stash waits for these two phases, away from the window in which every
            // CU reads x and y and writes the x stash (the first 10 us of the kernel move 68 MB: HBM-bound); split over the L2 and the
            // heads phase (feature tiles below / from R2_YSPREAD) so that neither phase waits for it
            if (YP > 0 && st2) stash_from_lds<P>(U, LDU, YP, (YP + 31) / 32 * 32, (T*)g.yT, g.spl, g.Bp, b0, tl, 0, R2_YSPREAD, yplanes);
#endif
            wg_barrier();                                               // BH2
            if constexpr (INFO) {                                          // classifier: backward through layer 2, dpre1 -> image A (c1 is stashed)
                f32x16 a3c;
                zero_acc<P>(a3c);
                gemm_seg<P, HS, DH, H_C2T>(a3c, hws, Urh + IMC);
                float dv1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dv1[r] = c1r[r] > 0.f ? a3c[r] : 0.f;                          // dpre1 (unit scale)
                put_lds<P>(dv1, U + IMA, LDU, 32 * hw, l31, h);
                if (st1) stash_tile<P>(U + IMC, LDU, 32 * hw, (T*)g.dc2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h, g.alpha);
            }
            if (st1) stash_tile<P>(Hb, LDH, 32 * hw, (T*)g.h2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
            if constexpr (XFH && !INFO) {
                if (st2x) {
                    if constexpr (R2_XSP == 0) xstash_reload<P, 4, 9>(g.x, g.ldx, rowof, b0, g.B, (T*)g.xT, g.spl, g.Bp, tl);
                    else if constexpr (R2_XSP == 1) xstash_reload<P, 0, 9>(g.x, g.ldx, rowof, b0, g.B, (T*)g.xT, g.spl, g.Bp, tl);
                    else xstash_reload<P, 0, 5>(g.x, g.ldx, rowof, b0, g.B, (T*)g.xT, g.spl, g.Bp, tl);
                }
            }
#if R2_LATE_Y
            if (YP > 0 && st2) stash_from_lds<P>(U, LDU, YP, (YP + 31) / 32 * 32, (T*)g.yT, g.spl, g.Bp, b0, tl, R2_YSPREAD, 1 << 30, yplanes);
#endif
            wg_barrier();                                               // BZ
            if constexpr (HS::HELPY) {
                // decoder layer 1 on the helper waves: the z block (one k-step) joins the label block; d1 -> Ha
                f32x16 az;
                zero_acc<P>(az);
                gemm_seg<P, HS, DH, H_W3Z>(az, hws, Zbrh);
                float hv[16], bv[16];
                bias16(Bias + OB3, 32 * hw, h, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) hv[r] = P::tanh_(az[r] + accy[r] + bv[r]);
                put_lds<P>(hv, Ha, LDH, 32 * hw, l31, h);
            }
            if (hw == 0 && st1) stash_tile<P>(Zb, LDZ, 0, (T*)g.zT, g.spl, b0, l31, h);
            if constexpr (XFH && !INFO && R2_XSP == 2) { if (st2x) xstash_reload<P, 5, 9>(g.x, g.ldx, rowof, b0, g.B, (T*)g.xT, g.spl, g.Bp, tl); }
            if (ht == 0) flags[0] = 0;                                     // read by the chain before BH1 of this tile; next written after BL1X of the next
            if constexpr (INFO) {
                if (st1) stash_tile<P>(U + IMA, LDU, 32 * hw, (T*)g.dc1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h, g.alpha);
                // the auxiliary net on the chain (four extra phases): its layer tiles go to the stash from here; pre-activation gradients
                // carry gamma - beta (quirk Q4)
                const float sa = g.gamma - g.beta;
                wg_barrier();                                           // BA1: a1 is in Ha
                if (st1) stash_tile<P>(Ha, LDH, 32 * hw, (T*)g.a1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
                if constexpr (XFH) { if (st2x) xstash_reload<P, 0, 5>(g.x, g.ldx, rowof, b0, g.B, (T*)g.xT, g.spl, g.Bp, tl); }
                wg_barrier();                                           // BA2: a2 is in Hb, the output partials in red2
                if (st1) {
                    stash_tile<P>(Hb, LDH, 32 * hw, (T*)g.a2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
                    if (hw == 0 && h == 0) {
                        const bool live_h = (b0 + l31) < g.B;
                        const float logit = red2[l31] + red2[32 + l31] + red2[64 + l31] + red2[96 + l31] + Binfo[OS3 + 1];
                        const float p = 1.f / (1.f + P::exp_(-logit));
                        const float u = live_h ? -g.invB * (ylab / (p + g.elbo_eps) - (1.f - ylab) / (1.f - p + g.elbo_eps)) : 0.f;
                        const float dpre3 = u * p * (1.f - p);
                        T* d3 = (T*)g.da3T + (b0 / KS) * (64 * E) + (l31 / E) * 32 * E + (l31 % E);
                        const T d3h = P::cvt(dpre3 * sa);
                        *d3 = d3h;
                        if constexpr (NP == 2) d3[g.spl] = P::cvt(dpre3 * sa - (float)d3h);
                    }
                }
                wg_barrier();                                           // BA3: dpre2 is in Ha
                if (st1) stash_tile<P>(Ha, LDH, 32 * hw, (T*)g.da2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h, sa);
                if constexpr (XFH) { if (st2x) xstash_reload<P, 5, 9>(g.x, g.ldx, rowof, b0, g.B, (T*)g.xT, g.spl, g.Bp, tl); }
                wg_barrier();                                           // BA4: dpre1 is in Hb
                if (st1) stash_tile<P>(Hb, LDH, 32 * hw, (T*)g.da1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h, sa);
            }
            wg_barrier();                                               // BD1
            // loss epilogue of the output layer (OFFL): x of the first round's tile is requested HERE, a phase early.  Requested at BD2
            // by every CU at once (the tile has long left the L2: 16.8 MB from the Infinity Cache / HBM within a microsecond) the burst
            // doubled the time of the first epilogue round AND of the chain's GEMM beside it (its weight loads queue behind the misses).
            // per-iteration opaque copy of the lane id: keeps this block's 40-odd LDS / global addresses out of loop-invariant hoisting
            // (hoisted to the tile loop's preheader they stay live across every phase and spill)
            int lo_ = lane;
            asm volatile("" : "+v"(lo_));
            const int l31o = lo_ & 31, ho = lo_ >> 5;
            const float* const xrow_o = g.x + rowof(l31o) * g.ldx + 4 * ho;
            f32x4 xo[4], xnx[4];                                            // this round's x, the next round's (requested a round ahead)
            if constexpr (OFFL) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) xo[gq] = reinterpret_cast<const F4U*>(xrow_o + 32 * hw + 8 * gq)->v;
            }
            if (st1) stash_tile<P>(Ha, LDH, 32 * hw, (T*)g.d1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
            wg_barrier();                                               // BD2
            if (st1) stash_tile<P>(Hb, LDH, 32 * hw, (T*)g.d2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
            float rec_h = 0.f;
            if constexpr (OFFL) {
                // tile hw + 4 I of round I
                const bool live = (b0 + l31o) < g.B;
                const float invB_l = live ? g.invB : 0.f;                  // frames past B contribute nothing
                const float* const xrow = xrow_o;
#pragma unroll 1
                for (int I = 0; I < 4; ++I) {                               // rolled: unrolled, the scheduler interleaves the rounds and the block spills
                    const int t = hw + 4 * I;
                    if (I < 3) {
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) xnx[gq] = reinterpret_cast<const F4U*>(xrow + 32 * (t + 4) + 8 * gq)->v;
                    }
                    R2_HSTAMP(21 + 2 * I);
                    wg_barrier();                                       // RB0 .. RB3
                    R2_HSTAMP(22 + 2 * I);
#if R2_HPRIO
                    __builtin_amdgcn_s_setprio(R2_HPRIO);
#endif
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float a4[4], da4[4];
                        get_raw4<P>(a4, U, LDU, 32 * t + 8 * gq + 4 * ho, l31o);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // recon term x / r - log(x + eps) + log r - 1 with r = exp(a) (models.py:122, utils.py:74): x e^{-a} + a is
                            // summed here; the log(x + eps) terms were summed while the x tile sat in registers (lsum2) and the -1's are a
                            // count -- 6 VALU instructions per element instead of 13 (this loop is what the output layer waits for)
                            const float xe = xo[gq][j] * __builtin_amdgcn_exp2f(a4[j] * -1.44269504088896341f);
                            rec_h += xe;
                            rec_h += a4[j];
                            da4[j] = fmaf(-xe, invB_l, invB_l);                  // d recon / d a = (1 - x / r) / B
                        }
                        typename P::Pack4 ph, pl;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { ph[j] = P::cvt(da4[j]); pl[j] = P::cvt(da4[j] - (float)ph[j]); }
                        *reinterpret_cast<typename P::Pack4*>(U + l31o * LDU + 32 * t + 8 * gq + 4 * ho) = ph;
                        *reinterpret_cast<typename P::Pack4*>(U + Pl<P>::lds + l31o * LDU + 32 * t + 8 * gq + 4 * ho) = pl;
                    }
#if R2_HPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) xo[gq] = xnx[gq];
                }
                rec_h = live ? rec_h - 64.f : 0.f;                          // the -1 of each of this lane's 64 elements; frames past B contribute nothing
            }
            wg_barrier();                                               // BDA
            for (int t = hw; t < NT_OUT; t += 4) if (st1) stash_tile<P>(U, LDU, 32 * t, (T*)g.daT + (int64_t)t * 32 * g.Bp, g.spl, b0, l31, h);
            const bool more = it + 1 < ntl;
            const int ntile = tile + (int)gridDim.x;
            if (more && gather) fill_rows(ntile, (it + 1) & 1);
            wg_barrier();                                               // BDD2: da consumed, U is free
            if (MODE == 1) {                                             // forward only: stage the next tile's x right away
                if (more) {
                    const int64_t fb0 = (int64_t)ntile * TB;
                    const int64_t* const fsrc = rowsrc + ((it + 1) & 1) * TB;
                    auto frowof = [&](int r) -> int64_t {
                        if (gather) return fsrc[r];
                        const int64_t br = fb0 + r;
                        return br < g.B ? br : g.B - 1;
                    };
                    if (g.fastx && (fb0 + TB) <= g.B) { tile513_issue(g.x, frowof, xv, tl); tile513_commit<P, XP, XFH>(xv, U, LDU, tl); }
                    else load_rows_to_lds<P, XFH>(g.x, g.ldx, XD, XP, fb0, g.B, U, LDU, tl, frowof);
                }
                wg_barrier();                                           // BRED'
                continue;
            }
            if (st1) stash_tile<P>(Hb, LDH, 32 * hw, (T*)g.dd2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
            // next tile's x: requested now, committed three phases later
            const int64_t nb0 = (int64_t)ntile * TB;
            const bool nfull = more && (nb0 + TB) <= g.B;
            const int64_t* const nsrc = rowsrc + ((it + 1) & 1) * TB;
            auto nrowof = [&](int r) -> int64_t {
                if (gather) return nsrc[r];
                const int64_t br = nb0 + r;
                return br < g.B ? br : g.B - 1;
            };
            x_in_regs = more && g.fastx && nfull;
            if (x_in_regs) tile513_issue(g.x, nrowof, xv, tl);
            else {                                                         // a full redefinition: without it the register image of the PREVIOUS tile
#pragma unroll                                                             // stays live around the whole loop (conditional definition) -- 68 registers
                for (int i = 0; i < NQ513; ++i) xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            wg_barrier();                                               // BDD1
            if (st1) stash_tile<P>(Ha, LDH, 32 * hw, (T*)g.dd1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
            wg_barrier();                                               // BDML
            if constexpr (YP > 0 && Y513) {
                // the next tile's labels: in flight from here to the next tile's BL1X (requested there, the 64 KB per CU take 4 us to
                // arrive beside the weight stream and the chain waits for them)
                y_in_regs = more && g.fasty && nfull;
                if (y_in_regs) tile513_issue(g.y, nrowof, yv, tl);
                else {
#pragma unroll
                    for (int i = 0; i < NQ513; ++i) yv[i] = f32x4{0.f, 0.f, 0.f, 0.f};     // full redefinition (see xv above)
                }
            }
            if (hw == 0 && st1) stash_tile<P>(Zb, LDZ, 0, (T*)g.dmlvT, g.spl, b0, l31, h);
            wg_barrier();                                               // BDH2
            if (st1) stash_tile<P>(Hb, LDH, 32 * hw, (T*)g.dh2T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
            wg_barrier();                                               // BDH1
            if (st1) stash_tile<P>(Ha, LDH, 32 * hw, (T*)g.dh1T + (int64_t)hw * 32 * g.Bp, g.spl, b0, l31, h);
            if constexpr (OFFL) {                                          // log terms of the tile(s) whose x this thread has committed so far
                const float rs = wave_sum(rec_h), ls = wave_sum(lsum2);
                if (lane == 0) { red[8 + hw] = rs; red[12 + hw] = ls; }
                lsum2 = 0.f;
            }
            if constexpr (INFO) { if (hw == 0) { const float bcs = wave_sum(bce_c); if (lane == 0) red2[128] = bcs; } }
            if (more) {
                if (x_in_regs) {
                    tile513_commit<P, XP, XFH>(xv, U, LDU, tl);
                    if constexpr (OFFL) lsum2 = tile513_log2sum(xv, g.elbo_eps, tl);      // counted with the next tile's sums
                } else load_rows_to_lds<P, XFH>(g.x, g.ldx, XD, XP, nb0, g.B, U, LDU, tl, nrowof, nullptr, OFFL ? &lsum2 : nullptr, g.elbo_eps);
            }
            wg_barrier();                                               // BRED
        }
    }
}

template <typename P, int YP, bool YENC, int MODE, bool INFO = false, bool DEFER = false>
static int launch_rows2_m(const RowsArgs& a, int grid, hipStream_t s) {
    const size_t lds = Lds2<P, INFO>::bytes;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)vae_rows2_kernel<P, YP, YENC, MODE, INFO, DEFER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(rows2 kernel, %zu B LDS): %s", lds, hipGetErrorString(e)); return (int)e; }
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((vae_rows2_kernel<P, YP, YENC, MODE, INFO, DEFER>), dim3(grid), dim3(512), lds, s, a);
    DVAE_LAUNCH_OK("vae_rows2_kernel");
    return 0;
}

template <typename P, int YP, bool YENC>
static int launch_rows2_t(const RowsArgs& a, int grid, hipStream_t s) {
    if (a.mode == 1) return launch_rows2_m<P, YP, YENC, 1>(a, grid, s);
    if (a.mode == 2) return launch_rows2_m<P, YP, YENC, 2>(a, grid, s);
#ifdef DVAE_DIAG
    if (a.defer.on) return launch_rows2_m<P, YP, YENC, 0, false, true>(a, grid, s);
#else
    if (a.defer.on) { set_error("rows2 kernel: the deferred optimizer step exists in the diagnostic build only (build.py --diag)"); return DVAE_E_UNSUPPORTED; }
#endif
    return launch_rows2_m<P, YP, YENC, 0>(a, grid, s);
}

// model: DVAE_MODEL_M1 / DVAE_MODEL_M2 / DVAE_MODEL_M2_DEC / DVAE_MODEL_M2_INFO (train step only); precision: DVAE_PREC_BF16 / DVAE_PREC_BF16X3
int launch_rows2(int precision, int model, int y_dim, const RowsArgs& a, int grid, hipStream_t s) {
    const bool m2 = model == DVAE_MODEL_M2;
    if (model == DVAE_MODEL_M2_INFO) {
        if (a.mode != 0) { set_error("rows2 kernel: M2_info runs the fused train step only (mode %d)", a.mode); return DVAE_E_UNSUPPORTED; }
        if (precision == DVAE_PREC_BF16X3) return launch_rows2_m<PolX3v2, 16, false, 0, true>(a, grid, s);
        if (precision == DVAE_PREC_BF16) return launch_rows2_m<PolBF16v2, 16, false, 0, true>(a, grid, s);
    }
    if (model == DVAE_MODEL_M2_DEC) {          // labels in the decoder only (y_dim 1): the plain VAE kernel with a zero-length encoder label segment
        if (precision == DVAE_PREC_BF16X3) return launch_rows2_t<PolX3v2, 16, false>(a, grid, s);
        if (precision == DVAE_PREC_BF16) return launch_rows2_t<PolBF16v2, 16, false>(a, grid, s);
    }
    if (precision == DVAE_PREC_BF16X3) {
        if (!m2) return launch_rows2_t<PolX3v2, 0, false>(a, grid, s);
        if (y_dim == 1) return launch_rows2_t<PolX3v2, 16, true>(a, grid, s);
        return launch_rows2_t<PolX3v2, 528, true>(a, grid, s);
    }
    if (precision == DVAE_PREC_BF16) {
        if (!m2) return launch_rows2_t<PolBF16v2, 0, false>(a, grid, s);
        if (y_dim == 1) return launch_rows2_t<PolBF16v2, 16, true>(a, grid, s);
        return launch_rows2_t<PolBF16v2, 528, true>(a, grid, s);
    }
    set_error("rows2 kernel: unsupported precision %d", precision);
    return DVAE_E_UNSUPPORTED;
}

bool rows2_supported(int precision, int model) {
    return (precision == DVAE_PREC_BF16 || precision == DVAE_PREC_BF16X3) &&
           (model == DVAE_MODEL_M1 || model == DVAE_MODEL_M2 || model == DVAE_MODEL_M2_INFO || model == DVAE_MODEL_M2_DEC);
}

}  // namespace fused
}  // namespace dvae
