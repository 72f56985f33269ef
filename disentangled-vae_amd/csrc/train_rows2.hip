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
#include "rows_shared.hpp"
#include "../../include/dvae_train.h"

namespace dvae {
namespace fused {

// MODE (RowsArgs::mode, compile time so the train-step instantiation carries none of the other modes' code or registers):
// 0 fused train step, 1 forward outputs only, 2 backward from upstream gradients
// DEFER: the deferred optimizer step (apply_common.hpp) -- the previous step's Adam update on the chain waves at the top of the launch,
// an arrival barrier in front of the first weight / bias load (which are sc1 loads then), the loss scalars by the last workgroup at the end
template <typename P, int YP, bool YENC, int MODE, bool INFO = false, bool DEFER = false>
__global__ __launch_bounds__(512) void vae_rows2_kernel(const RowsArgs g) {
    static_assert(!INFO || (MODE == 0 && YP == 16 && !YENC), "M2_info: train step, 1-dim label, encoder on x only");
    static_assert(!DEFER || (MODE == 0 && !INFO && sizeof(typename P::T) == 2), "deferred optimizer step: M1 / M2 train step, bf16 copies");
#define ROWS_XP XP
#define ROWS_PREFETCH 1
#define ROWS_CHAIN_WAVES 4
#include "rows_prologue.inc"
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
#include "rows_helper.inc"
    }
}

#undef ROWS_XP
#undef ROWS_CHAIN_WAVES
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
