// rows kernel, third generation (round 4): one 768-thread workgroup (12 waves, three per SIMD) per 32-frame tile.
//
//   GEMM waves 0-7   : the serial layer chain of the train step, N-SPLIT -- wave w owns 16 of a layer's 128 output features (tile t of the
//                      output layer's 34) for all 32 frames: out^T[16 x 32] = W[16 x K] * in^T on v_mfma_f32_16x16x32_bf16 (two MFMAs per
//                      weight fragment: frames 0-15 and 16-31), weights streamed from L2 into a register ring exactly as in the 8-wave kernel
//                      (wstream.hpp), activations from LDS.  The same fragment bytes and the same MFMA work per CU as four chain waves on
//                      32 x 32 tiles -- but TWO waves per SIMD issue the weight loads.  Why: tools/r03/kstep_bench.hip -- one wave gets one
//                      1 KB buffer_load_dwordx4 through every ~70 clocks however many it keeps in flight (141 clocks per 2 KB k-step beside
//                      97 clocks of MFMA: the chain of train_rows2.hip is bound by its own load issue), a second wave on the same SIMD gets
//                      its own 70.
//   helper waves 8-11: the helper block of train_rows2.hip, unchanged (rows_helper.inc): tiles in, stash out, loss epilogue.
// Phase list and barrier sequence: train_rows2.hip.  M1 / M2 train step (mode 0) under the split-bf16 policy; every other model, mode and
// policy stays on the 8-wave kernel.  <= 168 registers per wave.
//
// MEASURED (round 4, M2 y513 B=8192 bf16x3, tools/stamp_rows.py, profiles/r04_rows3_stamps.txt): parity-identical to the 8-wave kernel, and
// SLOWER -- 70.5 us beside 45.3.  The L1 x GEMM, the one phase bound by load issue, does go 1.41x faster (2.16 us beside 3.04); every short phase
// is ~0.5 us slower (the MFMA pipe of a SIMD is shared by its waves, so N-splitting a 128-wide layer over two waves buys no MFMA time and doubles
// the LDS reads of the activations), and the helper block -- written for 256 registers -- spills under 168 (x + label tiles in flight are 136 of
// them): the opening takes 16.9 us beside 6.1.  Kept as a DIAGNOSTIC build only (-DDVAE_DIAG, DVAE_ROWS=3): the product library has no rows3.
//
// C-tile ownership (16 x 16 x 32 MFMA): lane = (q = lane >> 4, n = lane & 15) holds features fb + 4 q + i (i < 4) of frames n (accumulator 0)
// and n + 16 (accumulator 1): an epilogue is 8 values per lane, the next layer's operand leaves as 8-byte LDS stores ([frame][feature] rows,
// hi and lo planes), and the in-place backward tiles (d1 / d2 re-read from their own planes) keep "same lane, same elements".
// Latent heads: waves 0 and 1 own one 16-row tile each whose rows are INTERLEAVED (weight-copy row map, apply_types.hpp: rowmap) so that lane q
// holds mu_k, mu_k+1, log_var_k, log_var_k+1 for k = 8 w + 2 q -- reparametrisation, KL terms and the backward through the sample stay in-lane.
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

#ifdef DVAE_DIAG
constexpr int XP3 = 544;          // x / label image width: 17 k-steps of 32
constexpr int ZP3 = 32;           // latent block of decoder layer 1: one k-step

#ifndef R3_D
#define R3_D 6
#endif
#ifndef R3_BD
#define R3_BD 2
#endif

// segments of a GEMM wave's weight stream in consumption order (wstream.hpp: WStream walks them)
enum { S_W1X, S_W1Y, S_W2, S_WMV, S_W3Y, S_W3Z, S_W4, S_W5A, S_W5B, S_W5C, S_W5D, S_W5T, S_W4T, S_W3ZT, S_WMVT, S_W2T, S_PAD, S_N };
template <typename P, int YP, bool YENC, int D> struct Sched3 {
    static constexpr int NSEG = S_N;
    static constexpr unsigned FBB = 1024u;                                     // bytes of one (16-row tile, 32-deep k-step) fragment block per plane
    static constexpr int raw(int s) {
        return s == S_W1X ? XP3 / 32 : s == S_W1Y ? (YENC ? YP / 32 : 0) : s == S_W3Y ? YP / 32 : s == S_W3Z ? ZP3 / 32
             : s == S_W5T ? NO / 32 : s == S_WMVT ? 1 : s == S_PAD ? 0 : HD / 32;
    }
    static constexpr int sum_raw() { int t = 0; for (int s = 0; s < S_PAD; ++s) t += raw(s); return t; }
    static constexpr int pad = (D - sum_raw() % D) % D;
    static constexpr int total = sum_raw() + pad;
    static constexpr bool WRAP = true;
    static constexpr int n(int s) { return s == S_PAD ? pad : raw(s); }
    static constexpr int start(int s) { int t = 0; for (int i = 0; i < s; ++i) t += n(i); return t; }
    static constexpr int seg_of(int q) { int s = 0; while (q >= start(s) + n(s)) ++s; return s; }
    // k-step stride: 128-row matrices are [k-step][8 tiles], the heads and the backward-z matrix [k-step][2], the output layer [k-step][34]
    static constexpr unsigned stride(int s) { return (s == S_WMV || s == S_W3ZT) ? 2u * FBB : (s >= S_W5A && s <= S_W5D) ? 34u * FBB : 8u * FBB; }
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// acc{0,1} += W(segment S)[16 rows of this wave] * act^T for frames n / n + 16.  brow = &act[n][8 q] (hi plane); half = elements between frame n and
// frame n + 16.  F16: both operands hold split-fp16 planes (struct X16).  blo (wave-uniform): the activations have a non-zero lo plane.
template <typename P, typename SC, int D, int S, bool F16, typename WS>
__device__ __forceinline__ void gemm_seg16(f32x4v& acc0, f32x4v& acc1, WS& w, const typename P::T* brow, int half, bool blo = true, bool active = true) {
    typedef typename P::Frag Frag;
    constexpr int N = SC::n(S), Q0 = SC::start(S);
    if constexpr (N > 0) {
        constexpr int BD = N < R3_BD ? N : R3_BD;
        Frag b0[BD][2], b1[BD][2];
        auto bload = [&](int slot, int ks) __attribute__((always_inline)) {
            b0[slot][0] = *reinterpret_cast<const Frag*>(brow + 32 * ks);
            b0[slot][1] = *reinterpret_cast<const Frag*>(brow + 32 * ks + Pl<P>::lds);
            b1[slot][0] = *reinterpret_cast<const Frag*>(brow + half + 32 * ks);
            b1[slot][1] = *reinterpret_cast<const Frag*>(brow + half + 32 * ks + Pl<P>::lds);
        };
        if (active) {
#pragma unroll
            for (int i = 0; i < BD; ++i) bload(i, i);
        }
        static_for<0, N>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if (active) {
                const Frag& ah = w.r[(Q0 + I) % D][0];
                const Frag& al = w.r[(Q0 + I) % D][1];
                if constexpr (F16) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, b0[I % BD][0]), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, b1[I % BD][0]), acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b0[I % BD][1]), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b1[I % BD][1]), acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b0[I % BD][0]), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b1[I % BD][0]), acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b0[I % BD][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b1[I % BD][0], acc1, 0, 0, 0);
                    if (blo) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b0[I % BD][1], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b1[I % BD][1], acc1, 0, 0, 0);
                    }
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b0[I % BD][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b1[I % BD][0], acc1, 0, 0, 0);
                }
                if constexpr (I + BD < N) bload(I % BD, I + BD);
            }
            w.template req<(Q0 + I + D) % SC::total>();
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

template <typename P, int YP, bool YENC>
__global__ __launch_bounds__(768) void vae_rows3_kernel(const RowsArgs g) {
    static_assert(P::NP == 2 && sizeof(typename P::T) == 2 && P::XF16, "rows3: split-bf16 policy with the split-fp16 x block");
    constexpr int MODE = 0;
    constexpr bool INFO = false, DEFER = false;
#define ROWS_XP XP3
#define ROWS_PREFETCH 0      // 168 VGPRs per wave at 12 waves: the next tile is not held in registers across the backward phases
#define ROWS_CHAIN_WAVES 8
#include "rows_prologue.inc"
    static_assert(OFFL, "rows3: the loss epilogue runs on the helper waves");
    (void)KS; (void)l31; (void)h;

    if (wave_u < 8) {
        // =========================================================== GEMM waves ===========================================================
        typedef typename P::Pack4 Pack4;
        const int cw = wave_u, fb = 16 * cw;
        const int n = lane & 15, q = lane >> 4;
        constexpr int D = R3_D;
        typedef Sched3<P, YP, YENC, D> SC;
        typedef WStream<P, SC, D> WS;
        constexpr unsigned FBB = SC::FBB;
        WS ws;
        ws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
        ws.voff = lane * 16;
        ws.pl = g.wpl_bytes;
        {
            auto mo = [&](const void* Wp) { return (unsigned)((const char*)Wp - (const char*)g.wcopy); };
            const unsigned tw = (unsigned)cw * FBB, t2 = (unsigned)(cw & 1) * FBB;      // (the two-tile matrices: waves >= 2 stream a tile they do not use)
            ws.sb[S_W1X] = mo(g.W1s) + tw;   ws.sb[S_W1Y] = mo(g.W1s) + tw + (XP3 / 32) * 8u * FBB;
            ws.sb[S_W2] = mo(g.W2s) + tw;    ws.sb[S_WMV] = mo(g.Wmvs) + t2;
            ws.sb[S_W3Z] = mo(g.W3s) + tw;   ws.sb[S_W3Y] = mo(g.W3s) + tw + (ZP3 / 32) * 8u * FBB;
            ws.sb[S_W4] = mo(g.W4s) + tw;
            ws.sb[S_W5A] = mo(g.W5s) + (unsigned)cw * FBB;        ws.sb[S_W5B] = mo(g.W5s) + (unsigned)(cw + 8) * FBB;
            ws.sb[S_W5C] = mo(g.W5s) + (unsigned)(cw + 16) * FBB; ws.sb[S_W5D] = mo(g.W5s) + (unsigned)(cw + 24) * FBB;
            ws.sb[S_W5T] = mo(g.W5t) + tw;   ws.sb[S_W4T] = mo(g.W4t) + tw;   ws.sb[S_W3ZT] = mo(g.W3zt) + t2;
            ws.sb[S_WMVT] = mo(g.Wmvt) + tw; ws.sb[S_W2T] = mo(g.W2t) + tw;   ws.sb[S_PAD] = mo(g.W1s);
        }
        ws.fill();
        const T* const Ur = U + n * LDU + 8 * q;
        const T* const Har = Ha + n * LDH + 8 * q;
        const T* const Hbr = Hb + n * LDH + 8 * q;
        const T* const Zbr = Zb + n * LDZ + 8 * q;
        constexpr int HU = 16 * LDU, HH = 16 * LDH, HZ = 16 * LDZ;
        const bool wz = cw < 2;                                                    // the latent tiles belong to waves 0 and 1
        const int ka = 8 * cw + 2 * q;                                             // (waves 0, 1) this lane's latent features ka, ka + 1
        double tot_rec = 0.0, tot_kl = 0.0;

        // next layer's operand: 4 consecutive features of frames n and n + 16 as (hi, lo) planes, [frame][feature] rows
        auto put8 = [&](const float (&v0)[4], const float (&v1)[4], T* buf, int ld, int col) __attribute__((always_inline)) {
            Pack4 h0, l0, h1, l1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h0[j] = P::cvt(v0[j]); l0[j] = P::cvt(v0[j] - (float)h0[j]);
                h1[j] = P::cvt(v1[j]); l1[j] = P::cvt(v1[j] - (float)h1[j]);
            }
            *reinterpret_cast<Pack4*>(buf + n * ld + col) = h0;
            *reinterpret_cast<Pack4*>(buf + Pl<P>::lds + n * ld + col) = l0;
            *reinterpret_cast<Pack4*>(buf + (n + 16) * ld + col) = h1;
            *reinterpret_cast<Pack4*>(buf + Pl<P>::lds + (n + 16) * ld + col) = l1;
        };
        auto get8 = [&](float (&v0)[4], float (&v1)[4], const T* buf, int ld, int col) __attribute__((always_inline)) {
            const Pack4 h0 = *reinterpret_cast<const Pack4*>(buf + n * ld + col), l0 = *reinterpret_cast<const Pack4*>(buf + Pl<P>::lds + n * ld + col);
            const Pack4 h1 = *reinterpret_cast<const Pack4*>(buf + (n + 16) * ld + col), l1 = *reinterpret_cast<const Pack4*>(buf + Pl<P>::lds + (n + 16) * ld + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = (float)h0[j] + (float)l0[j]; v1[j] = (float)h1[j] + (float)l1[j]; }
        };
        auto zero = [](f32x4v& a) __attribute__((always_inline)) { a = f32x4v{0.f, 0.f, 0.f, 0.f}; };

        for (int it = 0; it < ntl; ++it) {
            const int tile = tile0 + it * (int)gridDim.x;
            const int64_t b0 = (int64_t)tile * TB;
            const bool live0 = (b0 + n) < g.B, live1 = (b0 + n + 16) < g.B;
            const int64_t* const rsrc = rowsrc + (it & 1) * TB;
            float rec_lane = 0.f, kl_lane = 0.f;
            R2_STAMP(0);
            if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + 30] = clock64();
            // reparametrisation noise of this lane's two latent features, frames n and n + 16
            float ep[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
            if (wz) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    int64_t br = b0 + n + 16 * hf; br = br < g.B ? br : g.B - 1;
                    const bool lv_ = hf ? live1 : live0;
                    if (g.eps != nullptr) {
                        ep[hf][0] = lv_ ? g.eps[br * ZD + ka] : 0.f;
                        ep[hf][1] = lv_ ? g.eps[br * ZD + ka + 1] : 0.f;
                    } else {
                        // frame_noise8's numbering (rows_common.hpp): features 4 h' .. of draw 2 h', features 8 + 4 h' .. of draw 2 h' + 1
                        float e4[4];
                        const unsigned draw = ka < 8 ? 2u * (unsigned)(ka >> 2) : 2u * (unsigned)((ka - 8) >> 2) + 1u;
                        philox_normal4(g.rng_seed, (unsigned long long)br, g.rng_step, draw, e4);
                        ep[hf][0] = lv_ ? e4[ka & 3] : 0.f;
                        ep[hf][1] = lv_ ? e4[(ka & 3) + 1] : 0.f;
                    }
                }
            }
            if (it == 0) {
                if (gather) wg_barrier();                               // BROW
                wg_barrier();                                           // BX
            }
            R2_STAMP(1);
            // ---------------- encoder layer 1: [x | y] -> h1 ----------------
            f32x4v a0, a1;
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W1X, true>(a0, a1, ws, Ur, HU);
            a0 *= X16::ACC; a1 *= X16::ACC;                                        // x * 2^-3 and W * 2^6 (exact power-of-two scales)
            R2_STAMP(2);
            bool ylo = false;
            if (YP > 0) {
                wg_barrier();                                           // BL1X: the x image of U has been consumed
                wg_barrier();                                           // BY: the y image is in U
                ylo = __builtin_amdgcn_readfirstlane(flags[0]) != 0;
                gemm_seg16<P, SC, D, S_W1Y, false>(a0, a1, ws, Ur, HU, ylo);
            }
            float v0[4], v1[4];
            R2_STAMP(3);
            {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(Bias + OB1 + fb + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v0[j] = P::tanh_(a0[j] + bq[j]); v1[j] = P::tanh_(a1[j] + bq[j]);
                    keep[j * 512 + tid] = v0[j]; keep[(4 + j) * 512 + tid] = v1[j];
                }
            }
            put8(v0, v1, Ha, LDH, fb + 4 * q);
            wg_barrier();                                               // BH1
            R2_STAMP(4);
            // ---------------- encoder layer 2 ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W2, false>(a0, a1, ws, Har, HH);
            {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(Bias + OB2 + fb + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v0[j] = P::tanh_(a0[j] + bq[j]); v1[j] = P::tanh_(a1[j] + bq[j]);
                    keep[(8 + j) * 512 + tid] = v0[j]; keep[(12 + j) * 512 + tid] = v1[j];
                }
            }
            put8(v0, v1, Hb, LDH, fb + 4 * q);
            wg_barrier();                                               // BH2
            R2_STAMP(5);
            // ---------------- heads + reparametrisation (waves 0, 1): tile rows 4 q + {0, 1} = mu_ka, mu_ka+1, 4 q + {2, 3} = log_var_ka, log_var_ka+1 ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_WMV, false>(a0, a1, ws, Hbr, HH, true, wz);
            if (wz) {
                const float bm0 = Bias[OBMV + ka], bm1 = Bias[OBMV + ka + 1], bl0 = Bias[OBMV + ZD + ka], bl1 = Bias[OBMV + ZD + ka + 1];
                typedef typename P::T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f32x4v& a = hf ? a1 : a0;
                    const bool lv_ = hf ? live1 : live0;
                    const float mu0 = a[0] + bm0, mu1 = a[1] + bm1, lv0 = a[2] + bl0, lv1 = a[3] + bl1;
                    keepz[(4 * hf + 0) * 128 + tid] = mu0; keepz[(4 * hf + 1) * 128 + tid] = mu1;
                    keepz[(4 * hf + 2) * 128 + tid] = lv0; keepz[(4 * hf + 3) * 128 + tid] = lv1;
                    const float z0 = fmaf(P::exp_(0.5f * lv0), ep[hf][0], mu0), z1 = fmaf(P::exp_(0.5f * lv1), ep[hf][1], mu1);      // models.py:17, 20
                    if (lv_) kl_lane += (lv0 - mu0 * mu0 - P::exp_(lv0)) + (lv1 - mu1 * mu1 - P::exp_(lv1));                             // utils.py:75
                    T* const zr = Zb + (n + 16 * hf) * LDZ;
                    T2 zh, zl, zz;
                    zh[0] = P::cvt(z0); zh[1] = P::cvt(z1); zl[0] = P::cvt(z0 - (float)zh[0]); zl[1] = P::cvt(z1 - (float)zh[1]);
                    zz[0] = P::cvt(0.f); zz[1] = P::cvt(0.f);
                    *reinterpret_cast<T2*>(zr + ka) = zh;          *reinterpret_cast<T2*>(zr + Pl<P>::lds + ka) = zl;
                    *reinterpret_cast<T2*>(zr + ZD + ka) = zz;     *reinterpret_cast<T2*>(zr + Pl<P>::lds + ZD + ka) = zz;       // columns 16 .. 31 of the k-step: zeros
                }
            }
            // label block of decoder layer 1: independent of z
            f32x4v y0, y1;
            zero(y0); zero(y1);
            if constexpr (YP > 0) gemm_seg16<P, SC, D, S_W3Y, false>(y0, y1, ws, Ur, HU, ylo);
            wg_barrier();                                               // BZ
            R2_STAMP(6);
            // ---------------- decoder layer 1: [z | y] -> d1 ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W3Z, false>(a0, a1, ws, Zbr, HZ);
            {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(Bias + OB3 + fb + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = P::tanh_(a0[j] + y0[j] + bq[j]); v1[j] = P::tanh_(a1[j] + y1[j] + bq[j]); }
            }
            put8(v0, v1, Ha, LDH, fb + 4 * q);
            wg_barrier();                                               // BD1
            R2_STAMP(7);
            // ---------------- decoder layer 2 ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W4, false>(a0, a1, ws, Har, HH);
            // bin 512 (wave 7's dot product), requested a phase early; mapping of that block: lane = (frame lane & 31, half lane >> 5)
            int64_t rowx512;
            if (gather) rowx512 = rsrc[lane & 31];
            else { rowx512 = b0 + (lane & 31); rowx512 = rowx512 < g.B ? rowx512 : g.B - 1; }
            const float xv512 = cw == 7 ? g.x[rowx512 * g.ldx + XD - 1] : 0.f;
            {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(Bias + OB4 + fb + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = P::tanh_(a0[j] + bq[j]); v1[j] = P::tanh_(a1[j] + bq[j]); }
            }
            put8(v0, v1, Hb, LDH, fb + 4 * q);
            wg_barrier();                                               // BD2
            R2_STAMP(8);
            // ---------------- output layer a = W5 d2 + b5: four rounds of 8 x 16 rows; the raw pre-activations go to the helpers (put_raw4) ----------------
            static_for<0, 4>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                const int col = 16 * (cw + 8 * I) + 4 * q;
                zero(a0); zero(a1);
                gemm_seg16<P, SC, D, S_W5A + I, false>(a0, a1, ws, Hbr, HH);
                const f32x4 b5q = *reinterpret_cast<const f32x4*>(Bias + OB5 + col);
                const float r0[4] = {a0[0] + b5q[0], a0[1] + b5q[1], a0[2] + b5q[2], a0[3] + b5q[3]};
                const float r1[4] = {a1[0] + b5q[0], a1[1] + b5q[1], a1[2] + b5q[2], a1[3] + b5q[3]};
                put_raw4<P>(r0, U, LDU, col, n);
                put_raw4<P>(r1, U, LDU, col, n + 16);
                wg_barrier();                                           // RB0 .. RB3
            });
            if (cw == 7) {
                // the 33rd / 34th row tiles hold ONE real feature (bin 512): a 128-term dot product per frame
                const int f31 = lane & 31, hh = lane >> 5;
                const bool live = (b0 + f31) < g.B;
                const float invB_l = live ? g.invB : 0.f;
                const float* wl = Bias + OB5 + NO + 64 * hh;
                const T* drow = Hb + f31 * LDH + 64 * hh;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 64 / E; ++c) {
                    Frag dv[NP];
                    bloadp<P>(dv, drow + c * E);
#pragma unroll
                    for (int j = 0; j < E; j += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + c * E + j);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) s = fmaf((float)dv[0][j + jj] + (float)dv[1][j + jj], wv[jj], s);
                    }
                }
                s += __shfl_xor(s, 32, 64);
                const float a = s + Bias[OB5 + XD - 1];
                const float xe = xv512 * P::exp_(-a);
                if (hh == 0 && live) rec_lane += xe - P::log_(xv512 + g.elbo_eps) + a - 1.f;
                const float da512 = (1.f - xe) * invB_l;
                float dz16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dz16[r] = 0.f;
                if (hh == 0) dz16[0] = da512;
                T* const urow = U + f31 * LDU + (XD - 1) + 16 * hh;
#pragma unroll
                for (int c = 0; c < 16 / E; ++c) {
                    Frag fh, fl;
#pragma unroll
                    for (int j = 0; j < E; ++j) { fh[j] = P::cvt(dz16[c * E + j]); fl[j] = P::cvt(dz16[c * E + j] - (float)fh[j]); }
                    *reinterpret_cast<Frag*>(urow + c * E) = fh;
                    *reinterpret_cast<Frag*>(urow + Pl<P>::lds + c * E) = fl;
                }
            }
            wg_barrier();                                               // BDA
            R2_STAMP(9);
            // ---------------- backward: d2 <- da ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W5T, false>(a0, a1, ws, Ur, HU);
            get8(v0, v1, Hb, LDH, fb + 4 * q);                               // d2 of this lane's elements
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = a0[j] * (1.f - v0[j] * v0[j]); v1[j] = a1[j] * (1.f - v1[j] * v1[j]); }
            put8(v0, v1, Hb, LDH, fb + 4 * q);                               // in place
            wg_barrier();                                               // BDD2
            R2_STAMP(10);
            // ---------------- backward: d1 <- dpre_d2 ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W4T, false>(a0, a1, ws, Hbr, HH);
            get8(v0, v1, Ha, LDH, fb + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = a0[j] * (1.f - v0[j] * v0[j]); v1[j] = a1[j] * (1.f - v1[j] * v1[j]); }
            put8(v0, v1, Ha, LDH, fb + 4 * q);
            wg_barrier();                                               // BDD1
            R2_STAMP(11);
            // ---------------- backward: z <- dpre_d1 (waves 0, 1: tile rows 4 q + {0, 1} = dz_ka, dz_ka+1), then dmu / dlogvar ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W3ZT, false>(a0, a1, ws, Har, HH, true, wz);
            if (wz) {
                typedef typename P::T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f32x4v& a = hf ? a1 : a0;
                    const bool lv_ = hf ? live1 : live0;
                    float dm[2], dl[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float mu = keepz[(4 * hf + e) * 128 + tid], lv = keepz[(4 * hf + 2 + e) * 128 + tid];
                        const float dz = a[e];
                        dm[e] = lv_ ? dz + mu * g.invB : 0.f;                                                            // dmu: KL term of the fused step
                        dl[e] = lv_ ? dz * ep[hf][e] * (0.5f * P::exp_(0.5f * lv)) - 0.5f * g.invB * (1.f - P::exp_(lv)) : 0.f;   // dlogvar
                    }
                    T* const zr = Zb + (n + 16 * hf) * LDZ;
                    T2 mh, ml, lh, ll;
                    mh[0] = P::cvt(dm[0]); mh[1] = P::cvt(dm[1]); ml[0] = P::cvt(dm[0] - (float)mh[0]); ml[1] = P::cvt(dm[1] - (float)mh[1]);
                    lh[0] = P::cvt(dl[0]); lh[1] = P::cvt(dl[1]); ll[0] = P::cvt(dl[0] - (float)lh[0]); ll[1] = P::cvt(dl[1] - (float)lh[1]);
                    *reinterpret_cast<T2*>(zr + ka) = mh;        *reinterpret_cast<T2*>(zr + Pl<P>::lds + ka) = ml;
                    *reinterpret_cast<T2*>(zr + ZD + ka) = lh;   *reinterpret_cast<T2*>(zr + Pl<P>::lds + ZD + ka) = ll;
                }
            }
            wg_barrier();                                               // BDML
            R2_STAMP(12);
            // ---------------- backward: h2 <- [dmu | dlogvar] ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_WMVT, false>(a0, a1, ws, Zbr, HZ);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float k0 = keep[(8 + j) * 512 + tid], k1 = keep[(12 + j) * 512 + tid];
                v0[j] = a0[j] * (1.f - k0 * k0); v1[j] = a1[j] * (1.f - k1 * k1);
            }
            put8(v0, v1, Hb, LDH, fb + 4 * q);
            wg_barrier();                                               // BDH2
            R2_STAMP(13);
            // ---------------- backward: h1 <- dpre_h2 (inputs are data: stop here) ----------------
            zero(a0); zero(a1);
            gemm_seg16<P, SC, D, S_W2T, false>(a0, a1, ws, Hbr, HH);
            { f32x4v d0, d1; zero(d0); zero(d1); gemm_seg16<P, SC, D, S_PAD, false>(d0, d1, ws, Hbr, HH, true, false); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float k0 = keep[j * 512 + tid], k1 = keep[(4 + j) * 512 + tid];
                v0[j] = a0[j] * (1.f - k0 * k0); v1[j] = a1[j] * (1.f - k1 * k1);
            }
            put8(v0, v1, Ha, LDH, fb + 4 * q);
            wg_barrier();                                               // BDH1
            R2_STAMP(14);
            // ---------------- per-tile loss sums: red[0] = bin 512's term (wave 7), red[4], red[5] = KL (waves 0, 1) ----------------
            const float rs = wave_sum(rec_lane), ks = wave_sum(kl_lane);
            if (lane == 0) { if (cw == 7) red[0] = rs; if (cw < 2) red[4 + cw] = ks; }
            wg_barrier();                                               // BRED (the next tile's x image is in U)
            if (tid == 0) {
                tot_rec += (double)red[0] + (double)red[8] + (double)red[9] + (double)red[10] + (double)red[11]
                         - 0.6931471805599453 * ((double)red[12] + (double)red[13] + (double)red[14] + (double)red[15]);
                tot_kl += -0.5 * ((double)red[4] + (double)red[5]);
            }
        }
        R2_STAMP(15);
        if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + 31] = clock64();
        if (tid == 0) {
            g.partials[4 * blockIdx.x] = tot_rec;
            g.partials[4 * blockIdx.x + 1] = tot_kl;
            g.partials[4 * blockIdx.x + 2] = 0.0;
            g.partials[4 * blockIdx.x + 3] = 0.0;
        }
    } else {
#include "rows_helper.inc"
    }
}
#undef ROWS_XP
#undef ROWS_CHAIN_WAVES

template <typename P, int YP, bool YENC>
static int launch_rows3_t(const RowsArgs& a, int grid, hipStream_t s) {
    const size_t lds = Lds2<P, false>::bytes;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)vae_rows3_kernel<P, YP, YENC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(rows3 kernel, %zu B LDS): %s", lds, hipGetErrorString(e)); return (int)e; }
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((vae_rows3_kernel<P, YP, YENC>), dim3(grid), dim3(768), lds, s, a);
    DVAE_LAUNCH_OK("vae_rows3_kernel");
    return 0;
}

// M1 / M2 (y 1 or 513), train step (mode 0), split-bf16 policy
int launch_rows3(int model, int y_dim, const RowsArgs& a, int grid, hipStream_t s) {
    if (a.mode != 0) { set_error("rows3 kernel: train step only (mode %d)", a.mode); return DVAE_E_UNSUPPORTED; }
    if (model == DVAE_MODEL_M1) return launch_rows3_t<PolX3v2, 0, false>(a, grid, s);
    if (model == DVAE_MODEL_M2 && y_dim == 1) return launch_rows3_t<PolX3v2, 32, true>(a, grid, s);
    if (model == DVAE_MODEL_M2 && y_dim == XD) return launch_rows3_t<PolX3v2, XP3, true>(a, grid, s);
    set_error("rows3 kernel: M1 / M2 (y 1 or 513) only");
    return DVAE_E_UNSUPPORTED;
}

bool rows3_supported(int precision, int model) {
    return precision == DVAE_PREC_BF16X3 && (model == DVAE_MODEL_M1 || model == DVAE_MODEL_M2);
}
#else
int launch_rows3(int, int, const RowsArgs&, int, hipStream_t) {
    set_error("rows3 kernel: diagnostic builds only (build.py --diag)");
    return DVAE_E_UNSUPPORTED;
}
bool rows3_supported(int, int) { return false; }
#endif

}  // namespace fused
}  // namespace dvae
