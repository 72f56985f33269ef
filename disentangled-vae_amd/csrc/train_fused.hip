// Fused train step for the reference geometry (x 513, h [128,128], z 16, y 0/1/513):
// see include/dvae_train.h for the three-launch structure.
//
// rows kernel (one 256-thread workgroup = 4 waves per 32-frame tile).  Every layer is computed
// TRANSPOSED: out^T[features x frames] = W[features x K] * in^T[K x frames] on 32x32 MFMA tiles, so
//   * the A operand is the weight matrix in its natural nn.Linear [out][in] order: each lane reads
//     16 contiguous bytes of one weight row straight from L2 into VGPRs (a weight element is used
//     by exactly one wave of the workgroup, so staging it in LDS would buy nothing);
//   * the B operand is the previous layer's activations, kept in LDS as [frame][feature] rows whose
//     stride is an odd number of 16-byte slots (conflict-free ds_read_b128);
//   * in the C tile the lane is the frame and the 16 registers are features, so the tanh / exp /
//     loss epilogues, the [frame][feature] LDS write for the next layer and the coalesced
//     [feature][frame] stash store for the weight-gradient kernel all come out without shuffles;
//   * the four waves split the output features; fp32 copies of the tanh outputs stay in registers
//     for the backward pass of the same tile.
// Two operand policies share the code: exact fp32 (v_mfma_f32_32x32x2_f32, parity mode) and bf16
// operands with fp32 accumulation (v_mfma_f32_32x32x16_bf16, throughput mode).
//
// wgrad kernel: dW tile[32 out x 32 in] = sum over frames of dPre^T * In, both operands read from
// the [feature][frame] stash with 16-byte loads (frame = MFMA k index), 4 tiles per workgroup,
// the frame axis cut into `ksplit` slabs that the apply kernel sums in a fixed order
// (deterministic: no atomics anywhere).  Bias gradients ride along as one extra MFMA against a
// constant-one fragment.
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <map>
#include <algorithm>
#include <tuple>
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "fused_tiles.hpp"
#include "rows_common.hpp"
#include "apply_common.hpp"
#include "../../include/dvae_train.h"

namespace dvae {
namespace fused {


// x[32 frames][f0 .. f0+127] (fp32) for the loss epilogue: 16 coalesced dwords per thread, addresses clamped
template <typename RowOf>
__device__ __forceinline__ void xt_issue(const float* __restrict__ x, int ldx, RowOf rowof, int f0, float (&xr)[16], int tid) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 7, col = idx & 127;
        int cg = f0 + col; cg = cg < XD ? cg : XD - 1;
        xr[i] = x[rowof(row) * ldx + cg];          // rowof clamps rows past the batch
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void xt_commit(const float (&xr)[16], float* Xt, int ldxt, int64_t b0, int64_t B, int f0, int tid) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 7, col = idx & 127;
        Xt[row * ldxt + col] = (f0 + col < XD && b0 + row < B) ? xr[i] : 0.f;
    }
}

// "Classifier" (packages/models/models.py:41-63): 128-128-1 relu / relu / sigmoid MLP on the current tile, its
// binary_cross_entropy against the frame label (utils.py:55-56) and the unit-scale backward, all on chip.
// Used twice by M2_info (scripts/training_M2_info_vad.py:159-183): classifier on x, auxiliary net on z.
// Each wave owns 32 hidden features; the 1-wide output layer is a VALU dot product reduced through LDS.
struct SideArgs {
    WRef W1, W2, W2t, W1t;          // W1t only when the gradient wrt the input is needed
    unsigned s1, s1t;               // k-step strides of W1 / W1t
    const float *b1, *b2, *w3;      // LDS tables
    float b3;
    float y, invB, eps;
    bool live, need_dx;
    float scale;                    // factor applied to the stashed pre-activation gradients (loss weight)
    void *h1T, *h2T, *d1T, *d2T, *d3T;
    int64_t Bp, b0, spl;
};

template <typename P, int K1STEPS>
__device__ __forceinline__ void side_mlp(__amdgpu_buffer_rsrc_t wrs, const SideArgs& a, const typename P::T* in_row,
                                         typename P::T* Ha, typename P::T* Hb, float* redbuf, int wave, int l31, int h,
                                         unsigned S4, float& bce_frame, float& p_out, f32x16& dx) {
    typedef typename P::T T;
    constexpr int E = P::E, KS = P::KSTEP, LDH = Ld<T>::hh;
    const int fb = 32 * wave;
    const T* const Har = Ha + l31 * LDH + h * E;
    const T* const Hbr = Hb + l31 * LDH + h * E;
    f32x16 acc;
    float bv[16], w3v[16], c1r[16], c2r[16], dv[16];
    // layer 1
    WPre<P, K1STEPS> w1;
    wprefetch<P, K1STEPS>(w1, wrs, a.W1, a.s1);
    zero_acc<P>(acc);
    gemm_block<P, K1STEPS>(acc, w1, wrs, a.W1, in_row, a.s1);
    WPre<P, HD / KS, P::PRE128> w2;
    wprefetch<P, HD / KS>(w2, wrs, a.W2, S4);
    bias16(a.b1, fb, h, bv);
#pragma unroll
    for (int r = 0; r < 16; ++r) c1r[r] = fmaxf(acc[r] + bv[r], 0.f);
    put_lds<P>(c1r, Ha, LDH, fb, l31, h);
    __syncthreads();
    // layer 2 + output dot product
    zero_acc<P>(acc);
    gemm_block<P, HD / KS>(acc, w2, wrs, a.W2, Har, S4, [&]() { stash_tile<P>(Ha, LDH, fb, (T*)a.h1T + (int64_t)wave * 32 * a.Bp, a.spl, a.b0, l31, h); });
    WPre<P, HD / KS, P::PRE128> w2t;
    wprefetch<P, HD / KS>(w2t, wrs, a.W2t, S4);
    bias16(a.b2, fb, h, bv);
    bias16(a.w3, fb, h, w3v);
    float pd = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c2r[r] = fmaxf(acc[r] + bv[r], 0.f); pd = fmaf(w3v[r], c2r[r], pd); }
    put_lds<P>(c2r, Hb, LDH, fb, l31, h);
    pd += __shfl_xor(pd, 32, 64);
    if (h == 0) redbuf[wave * 32 + l31] = pd;
    __syncthreads();
    const float logit = redbuf[l31] + redbuf[32 + l31] + redbuf[64 + l31] + redbuf[96 + l31] + a.b3;
    const float p = 1.f / (1.f + P::exp_(-logit));
    p_out = p;
    const float lp = P::log_(p + a.eps), lq = P::log_(1.f - p + a.eps);
    bce_frame = a.live ? -(a.y * lp + (1.f - a.y) * lq) : 0.f;                              // utils.py:55-56, this frame's term
    const float u = a.live ? -a.invB * (a.y / (p + a.eps) - (1.f - a.y) / (1.f - p + a.eps)) : 0.f;   // d BCE / d p
    const float dpre3 = u * p * (1.f - p);
    stash_tile<P>(Hb, LDH, fb, (T*)a.h2T + (int64_t)wave * 32 * a.Bp, a.spl, a.b0, l31, h);
    if (wave == 0 && h == 0) {       // output pre-activation gradient: feature row 0 of a 32-row stash tile
        T* d3 = (T*)a.d3T + (a.b0 / KS) * (64 * E) + (l31 / E) * 32 * E + (l31 % E);
        const T d3h = P::cvt(dpre3 * a.scale);
        *d3 = d3h;
        if constexpr (P::NP == 2) d3[a.spl] = P::cvt(dpre3 * a.scale - (float)d3h);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = c2r[r] > 0.f ? w3v[r] * dpre3 : 0.f;            // dpre2 (unit scale)
    put_lds<P>(dv, Ha, LDH, fb, l31, h);
    __syncthreads();
    // backward through layer 2
    zero_acc<P>(acc);
    gemm_block<P, HD / KS>(acc, w2t, wrs, a.W2t, Har, S4, [&]() { stash_tile<P>(Ha, LDH, fb, (T*)a.d2T + (int64_t)wave * 32 * a.Bp, a.spl, a.b0, l31, h, a.scale); });
    WPre<P, HD / KS> w1t;
    if (a.need_dx && wave == 0) wprefetch<P, HD / KS>(w1t, wrs, a.W1t, a.s1t);
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = c1r[r] > 0.f ? acc[r] : 0.f;                      // dpre1 (unit scale)
    put_lds<P>(dv, Hb, LDH, fb, l31, h);
    __syncthreads();
    zero_acc<P>(dx);
    if (a.need_dx && wave == 0) {
        gemm_block<P, HD / KS>(dx, w1t, wrs, a.W1t, Hbr, a.s1t, [&]() { stash_tile<P>(Hb, LDH, fb, (T*)a.d1T + (int64_t)wave * 32 * a.Bp, a.spl, a.b0, l31, h, a.scale); });
    } else {
        stash_tile<P>(Hb, LDH, fb, (T*)a.d1T + (int64_t)wave * 32 * a.Bp, a.spl, a.b0, l31, h, a.scale);
    }
    __syncthreads();
}

template <typename P, int YP, bool YENC, bool INFO>
__global__ __launch_bounds__(256, 1) void vae_rows_kernel(const RowsArgs g) {
    typedef typename P::T T;
    constexpr int E = P::E;
    constexpr int KS = P::KSTEP;
    constexpr int LDU = Ld<T>::u, LDH = Ld<T>::hh, LDZ = Ld<T>::z, LDX = Ld<T>::xt;
    constexpr int LD1 = XP + (YENC ? YP : 0);           // W1 shadow row length
    constexpr int LD3 = ZD + YP;                        // W3 shadow row length
    constexpr bool Y513 = (YP == XP);                   // IBM labels: y has the same 513-column shape as x
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* U = reinterpret_cast<T*>(smem);
    T* Ha = U + TB * LDU;
    T* Hb = Ha + TB * LDH;
    T* Zb = Hb + TB * LDH;
    float* Xt = reinterpret_cast<float*>(U + P::NP * Ld<T>::act_elems);     // behind the operand plane(s).  XFULL: dense [32][513] fp32 x tile; else [32][129] slice
    float* Bias = Xt + (P::XFULL ? Ld<T>::xf_floats : Ld<T>::xt_floats);
    constexpr int OB1 = 0, OB2 = HD, OBMV = 2 * HD, OB3 = 2 * HD + 32, OB4 = 3 * HD + 32, OB5 = 4 * HD + 32;
    // M2_info tables behind the VAE biases: bc1 bc2 wc3 ba1 ba2 wa3 (128 each), then bc3, ba3
    constexpr int OI = Ld<T>::nbias, OBC1 = OI, OBC2 = OI + HD, OWC3 = OI + 2 * HD, OBA1 = OI + 3 * HD, OBA2 = OI + 4 * HD, OWA3 = OI + 5 * HD, OS3 = OI + 6 * HD;
    __shared__ float red[16];
    __shared__ float red2[128];
    __shared__ int64_t rowsrc[TB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int fb = 32 * wave;                           // this wave's feature block in 128-wide layers
    // fragment-major weight copies: [k-step][row tile][lane][E] (k-step major: the fragments a wave keeps in
    // flight then sit >= 4 KB apart and spread over the L2 channels); this wave's tile = wave in 128-row layers
    constexpr int FB = 64 * E;                          // elements per (tile, k-step) block
    // weight copies through ONE buffer descriptor: per-lane byte offset + wave-uniform (matrix, tile) offset
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned SZ = sizeof(T);
    auto wbase = [&](const void* Wp, int tile, int ld) -> WRef {
        const unsigned m = (unsigned)((const char*)Wp - (const char*)g.wcopy);
        if (WFRAG) return WRef{lane * 16, m + (unsigned)tile * (FB * SZ), g.wpl_bytes};
        return WRef{(int)((l31 * ld + h * E) * SZ), m + (unsigned)(32 * tile * ld) * SZ, g.wpl_bytes};
    };
    // byte strides between k-steps (4-tile, 1-tile and 17-tile matrices), between row tiles of W5s, and the
    // byte offsets of the y k-blocks inside W1 / W3
    constexpr unsigned S4 = (WFRAG ? 4 * FB : 2 * E) * SZ, S1 = (WFRAG ? FB : 2 * E) * SZ, S17 = (WFRAG ? NT_OUT * FB : 2 * E) * SZ;
    constexpr unsigned TSTEP = (WFRAG ? FB : 32 * HD) * SZ;
    constexpr unsigned KB1 = (WFRAG ? (XP / KS) * 4 * FB : XP) * SZ;
    constexpr unsigned KB3 = (WFRAG ? (ZD / KS) * 4 * FB : ZD) * SZ;
    const WRef W1r = wbase(g.W1s, wave_u, LD1);
    const WRef W2r = wbase(g.W2s, wave_u, HD);
    const WRef Wmvr = wbase(g.Wmvs, 0, HD);
    const WRef W3r = wbase(g.W3s, wave_u, LD3);
    const WRef W4r = wbase(g.W4s, wave_u, HD);
    const WRef W5s = wbase(g.W5s, 0, HD);
    const WRef W5tr = wbase(g.W5t, wave_u, NO);
    const WRef W4tr = wbase(g.W4t, wave_u, HD);
    const WRef W3ztr = wbase(g.W3zt, 0, HD);
    const WRef Wmvtr = wbase(g.Wmvt, wave_u, 32);
    const WRef W2tr = wbase(g.W2t, wave_u, HD);
    auto woff = [](WRef r, unsigned bytes) { return WRef{r.voff, r.soff + bytes, r.pl}; };
    const T* const Ur = U + l31 * LDU + h * E;
    const T* const Har = Ha + l31 * LDH + h * E;
    const T* const Hbr = Hb + l31 * LDH + h * E;
    const T* const Zbr = Zb + l31 * LDZ + h * E;

    double tot_rec = 0.0, tot_kl = 0.0, tot_bc = 0.0, tot_ba = 0.0;

    // fp32 bias table -> LDS once (epilogues must not queue global loads behind the weight prefetch).
    // The loads are issued here (clamped addresses instead of branches); the LDS stores wait until the first
    // tile's x loads are in flight, so the kernel's first HBM round trip carries both.
    constexpr int NB = (Ld<T>::nbias + 255) / 256;
    float bvv[NB];
#pragma unroll
    for (int it = 0; it < NB; ++it) {
        int i = tid + 256 * it;
        i = i < Ld<T>::nbias ? i : Ld<T>::nbias - 1;
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
        bvv[it] = src[k];
    }
    auto store_bias_table = [&]() {
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int i = tid + 256 * it;
            if (i < Ld<T>::nbias) Bias[i] = (i >= OB5 + XD && i < OB5 + NO) ? 0.f : bvv[it];
        }
        if (INFO) {
            for (int i = tid; i < 6 * HD + 2; i += 256) {
                float v;
                const int q = i / HD, k = i - q * HD;
                if (q == 0) v = g.bc1[k]; else if (q == 1) v = g.bc2[k]; else if (q == 2) v = g.wc3[k];
                else if (q == 3) v = g.ba1[k]; else if (q == 4) v = g.ba2[k]; else if (q == 5) v = g.wa3[k];
                else v = k == 0 ? g.bc3[0] : g.ba3[0];
                Bias[OI + i] = v;
            }
        }
    };
    bool bias_pending = true;

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int64_t b0 = (int64_t)tile * TB;
        const bool live = (b0 + l31) < g.B;             // this lane's frame exists
        const bool full = (b0 + TB) <= g.B;
        float rec_lane = 0.f, kl_lane = 0.f, bce_c = 0.f, bce_a = 0.f;
        float y_l = 0.f;
        // source row of the tile's frame r (clamped to the batch): identity, or through the gather table
        if (g.rows != nullptr) {
            __syncthreads();                                 // previous tile's readers are done with rowsrc
            if (tid < TB) {
                const int64_t bf = b0 + tid;
                const int64_t br = bf < g.B ? bf : g.B - 1;
                int64_t rr = g.rows[br];
                if (rr < 0 || rr >= g.n_rows) { rr = 0; if (g.bad_rows && bf < g.B) atomicAdd(g.bad_rows, 1); }   // never dereference an out-of-range index
                rowsrc[tid] = rr;
            }
            __syncthreads();
        }
        auto rowof = [&](int r) -> int64_t {
            if (g.rows != nullptr) return rowsrc[r];
            const int64_t br = b0 + r;
            return br < g.B ? br : g.B - 1;
        };
        if (INFO) y_l = g.y[rowof(l31) * g.ldy];
        f32x16 dzu;                                      // M2_info: d BCE_aux / d z (unit scale), wave 0
        // per-iteration opaque copy of the thread id: stops the compiler from hoisting the ~70 per-thread
        // staging addresses out of the tile loop (they would live across the whole loop and spill)
        int tl = tid;
        asm volatile("" : "+v"(tl));

        DVAE_STAMP(0);
        if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + 30] = clock64();
        // reparametrisation noise of this lane's frame (wave 0 owns the latent tile): requested first,
        // long before it is needed
        float ep_r[8];
        if (wave == 0) {
            int64_t br = b0 + l31; br = br < g.B ? br : g.B - 1;
            if (g.eps != nullptr) {
                const f32x4 e0 = *reinterpret_cast<const f32x4*>(g.eps + br * ZD + 4 * h);
                const f32x4 e1 = *reinterpret_cast<const f32x4*>(g.eps + br * ZD + 8 + 4 * h);
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) { ep_r[jq] = live ? e0[jq] : 0.f; ep_r[4 + jq] = live ? e1[jq] : 0.f; }
            } else {                                             // drawn here: no noise tensor, no extra launch
                frame_noise8(g.rng_seed, (unsigned long long)br, g.rng_step, h, ep_r);
#pragma unroll
                for (int jq = 0; jq < 8; ++jq) ep_r[jq] = live ? ep_r[jq] : 0.f;
            }
        }
        // ---------------- encoder layer 1: [x | y] -> h1 ----------------
        WPre<P, XP / KS> w1x;
        wprefetch<P, XP / KS>(w1x, wrs, W1r, S4);
        const bool yfast = Y513 && g.fasty && full;
        f32x4 yv[NQ513];
        if (g.fastx && full) {
            f32x4 xv[NQ513];
            tile513_issue(g.x, rowof, xv, tl);
            if constexpr (Y513 && P::EARLY_Y) {
                if (yfast) tile513_issue(g.y, rowof, yv, tl);      // y tile in flight under the x commit and the x GEMM
            }
            if (bias_pending) { store_bias_table(); bias_pending = false; }
            tile513_commit<P, XP>(xv, U, LDU, tl, P::XFULL ? Xt : nullptr);
        } else {
            if (bias_pending) { store_bias_table(); bias_pending = false; }
            load_rows_to_lds<P>(g.x, g.ldx, XD, XP, b0, g.B, U, LDU, tl, rowof, P::XFULL ? Xt : nullptr);
            if constexpr (Y513 && P::EARLY_Y) {
                if (yfast) tile513_issue(g.y, rowof, yv, tl);
            }
        }
        __syncthreads();
        DVAE_STAMP(1);
        f32x16 acc;
        zero_acc<P>(acc);
        gemm_block<P, XP / KS>(acc, w1x, wrs, W1r, Ur, S4, [&]() {
            if (!(g.ablate & 2)) stash_from_lds<P>(U, LDU, XP, NO, (T*)g.xT, g.spl, g.Bp, b0, tl);
        });
        DVAE_STAMP(2);
        if (INFO) {
            const f32x16 acc_keep = acc;
            SideArgs sa;
            sa.W1 = wbase(g.Wc1s, wave_u, XP); sa.W2 = wbase(g.Wc2s, wave_u, HD); sa.W2t = wbase(g.Wc2t, wave_u, HD); sa.W1t = sa.W2t;
            sa.s1 = S4; sa.s1t = S4; sa.b1 = Bias + OBC1; sa.b2 = Bias + OBC2; sa.w3 = Bias + OWC3; sa.b3 = Bias[OS3];
            sa.y = y_l; sa.invB = g.invB; sa.eps = g.elbo_eps; sa.live = live; sa.need_dx = false; sa.scale = g.alpha;
            sa.h1T = g.c1T; sa.h2T = g.c2T; sa.d1T = g.dc1T; sa.d2T = g.dc2T; sa.d3T = g.dc3T; sa.Bp = g.Bp; sa.b0 = b0; sa.spl = g.spl;
            float pc; f32x16 dxc;
            side_mlp<P, XP / KS>(wrs, sa, Ur, Ha, Hb, red2, wave, l31, h, S4, bce_c, pc, dxc);
            acc = acc_keep;
        }
        WPre<P, HD / KS, P::PRE128> w2;
        WPre<P, (YENC ? YP : 0) / KS> w1y;
        if (YENC) wprefetch<P, (YENC ? YP : 0) / KS>(w1y, wrs, woff(W1r, KB1), S4);
        else wprefetch<P, HD / KS>(w2, wrs, W2r, S4);
        if (YP > 0) {
            __syncthreads();
            if (Y513 && yfast) {
                if constexpr (!P::EARLY_Y) tile513_issue(g.y, rowof, yv, tl);
                tile513_commit<P, XP>(yv, U, LDU, tl);
            } else {
                load_rows_to_lds<P>(g.y, g.ldy, g.ydim, YP, b0, g.B, U, LDU, tl, rowof);
            }
            __syncthreads();
            if (YENC) {
                gemm_block<P, (YENC ? YP : 0) / KS>(acc, w1y, wrs, woff(W1r, KB1), Ur, S4, [&]() {
                    if (!(g.ablate & 2)) stash_from_lds<P>(U, LDU, YP, (YP + 31) / 32 * 32, (T*)g.yT, g.spl, g.Bp, b0, tl);
                });
                wprefetch<P, HD / KS>(w2, wrs, W2r, S4);
            } else {
                if (!(g.ablate & 2)) stash_from_lds<P>(U, LDU, YP, (YP + 31) / 32 * 32, (T*)g.yT, g.spl, g.Bp, b0, tl);
            }
        }
        float h1r[16], bv[16];
        DVAE_STAMP(3);
        bias16(Bias + OB1, fb, h, bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) h1r[r] = P::tanh_(acc[r] + bv[r]);
        put_lds<P>(h1r, Ha, LDH, fb, l31, h);
        __syncthreads();

        DVAE_STAMP(4);
        // ---------------- encoder layer 2 ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, w2, wrs, W2r, Har, S4, [&]() { DVAE_FSTAMP(16); stash_tile<P>(Ha, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.h1T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h); DVAE_FSTAMP(17); });
        DVAE_FSTAMP(18);
        WPre<P, HD / KS, P::PRE128> wmv;
        WPre<P, ZD / KS> w3z;
        if (wave == 0) wprefetch<P, HD / KS>(wmv, wrs, Wmvr, S1);
        wprefetch<P, ZD / KS>(w3z, wrs, W3r, S4);
        float h2r[16];
        bias16(Bias + OB2, fb, h, bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) h2r[r] = P::tanh_(acc[r] + bv[r]);
        DVAE_FSTAMP(19);
        put_lds<P>(h2r, Hb, LDH, fb, l31, h);
        DVAE_FSTAMP(20);
        __syncthreads();

        DVAE_STAMP(5);
        // ---------------- heads + reparametrisation (wave 0): rows 0-15 mu, 16-31 log_var ----------------
        float mu_r[8], lv_r[8], sd_r[8];
        // the label block of decoder layer 1 (33 k-steps) does not depend on z: its first fragments are requested here,
        // a whole phase ahead (three of the four waves idle through the heads anyway)
        WPre<P, (YP > 0 ? YP : KS) / KS, P::PREBIG> w3y;
        if (YP > 0) wprefetch<P, (YP > 0 ? YP : KS) / KS>(w3y, wrs, woff(W3r, KB3), S4);
        if (wave == 0) {
            zero_acc<P>(acc);
            gemm_block<P, HD / KS>(acc, wmv, wrs, Wmvr, Hbr, S1, [&]() { stash_tile<P>(Hb, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.h2T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h); });
            float zv[16];
            bias16(Bias + OBMV, 0, h, bv);                          // rows 0-15 bmu, 16-31 blv
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                mu_r[r] = acc[r] + bv[r];
                lv_r[r] = acc[r + 8] + bv[r + 8];
                sd_r[r] = P::exp_(0.5f * lv_r[r]);                 // models.py:17
                zv[r] = fmaf(sd_r[r], ep_r[r], mu_r[r]);           // models.py:20
                zv[r + 8] = 0.f;
                if (live) kl_lane += lv_r[r] - mu_r[r] * mu_r[r] - P::exp_(lv_r[r]);   // utils.py:75
            }
            // z block of the decoder input: features 0..15 valid, 16..31 zero
            put_lds<P>(zv, Zb, LDZ, 0, l31, h);
        } else {
            stash_tile<P>(Hb, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.h2T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h);
        }
        __syncthreads();

        DVAE_STAMP(6);
        if (INFO) {
            SideArgs sa;
            sa.W1 = wbase(g.Wa1s, wave_u, ZD); sa.W2 = wbase(g.Wa2s, wave_u, HD); sa.W2t = wbase(g.Wa2t, wave_u, HD); sa.W1t = wbase(g.Wa1t, 0, HD);
            sa.s1 = S4; sa.s1t = S1; sa.b1 = Bias + OBA1; sa.b2 = Bias + OBA2; sa.w3 = Bias + OWA3; sa.b3 = Bias[OS3 + 1];
            sa.y = y_l; sa.invB = g.invB; sa.eps = g.elbo_eps; sa.live = live; sa.need_dx = true; sa.scale = g.gamma - g.beta;
            sa.h1T = g.a1T; sa.h2T = g.a2T; sa.d1T = g.da1T; sa.d2T = g.da2T; sa.d3T = g.da3T; sa.Bp = g.Bp; sa.b0 = b0; sa.spl = g.spl;
            float pa;
            if (wave == 0) stash_tile<P>(Zb, LDZ, 0, (T*)g.zT, g.spl, b0, l31, h);
            side_mlp<P, ZD / KS>(wrs, sa, Zbr, Ha, Hb, red2, wave, l31, h, S4, bce_a, pa, dzu);
        }
        // ---------------- decoder layer 1: [z | y] -> d1 ----------------
        zero_acc<P>(acc);
        gemm_block<P, ZD / KS>(acc, w3z, wrs, W3r, Zbr, S4, [&]() { if (!INFO && wave == 0) stash_tile<P>(Zb, LDZ, 0, (T*)g.zT, g.spl, b0, l31, h); });
        WPre<P, HD / KS, P::PRE128> w4;
        if (YP > 0) gemm_block<P, (YP > 0 ? YP : KS) / KS>(acc, w3y, wrs, woff(W3r, KB3), Ur, S4);
        wprefetch<P, HD / KS>(w4, wrs, W4r, S4);
        float d1r[16];
        bias16(Bias + OB3, fb, h, bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) d1r[r] = P::tanh_(acc[r] + bv[r]);
        put_lds<P>(d1r, Ha, LDH, fb, l31, h);
        __syncthreads();

        DVAE_STAMP(7);
        // ---------------- decoder layer 2 ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, w4, wrs, W4r, Har, S4, [&]() { stash_tile<P>(Ha, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.d1T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h); });
        WPre<P, HD / KS, P::PRE128> w5;
        wprefetch<P, HD / KS>(w5, wrs, woff(W5s, wave_u * TSTEP), S17);
        float xr[16];
        if (!P::XFULL) xt_issue(g.x, g.ldx, rowof, 0, xr, tl);
        float d2r[16];
        bias16(Bias + OB4, fb, h, bv);
#pragma unroll
        for (int r = 0; r < 16; ++r) d2r[r] = P::tanh_(acc[r] + bv[r]);
        put_lds<P>(d2r, Hb, LDH, fb, l31, h);
        __syncthreads();

        DVAE_STAMP(8);
        // ---------------- output layer a = W5 d2 + b5, Itakura-Saito terms, da -> U ----------------
        WPre<P, NO / KS> w5t;
        // one 32-feature tile t of the output layer for this wave: GEMM, loss terms, da
        auto out_tile = [&](int t, const float* xsrc, int xld, int xcol0, int xcmax) {
            if (t == 4) DVAE_FSTAMP(21);
            zero_acc<P>(acc);
            const WRef wr = woff(W5s, (unsigned)t * TSTEP);
            gemm_block<P, HD / KS>(acc, w5, wrs, wr, Hbr, S17, [&]() { if (t < 4) stash_tile<P>(Hb, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.d2T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h); });
            if (t == 4) DVAE_FSTAMP(22);
            if (t + 4 < (P::XFULL ? NT_OUT - 1 : NT_OUT)) wprefetch<P, HD / KS>(w5, wrs, woff(wr, 4 * TSTEP), S17);
            else wprefetch<P, NO / KS>(w5t, wrs, W5tr, S4);
            if (t == 4) DVAE_FSTAMP(23);
            float da[16], b5v[16], xs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                       // all x reads up front: one LDS wait, not sixteen
                int xc = xcol0 + feat_of(r, h); xc = xc < xcmax ? xc : xcmax;
                xs[r] = xsrc[l31 * xld + xc];
            }
            bias16(Bias + OB5, 32 * t, h, b5v);
            if (t == 4) DVAE_FSTAMP(24);
            const float invB_l = live ? g.invB : 0.f;            // frames past B contribute nothing
            if (t < NT_OUT - 1) {                                // all 32 features of the tile exist
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = acc[r] + b5v[r];
                    const float xe = xs[r] * P::exp_(-a);        // x / r,  r = exp(a)  (models.py:122)
                    rec_lane += xe - P::log_(xs[r] + g.elbo_eps) + a - 1.f;   // utils.py:74 (log r = a)
                    da[r] = (1.f - xe) * invB_l;                 // d recon / d a
                }
            } else {                                             // last tile: features >= 513 are padding
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = 32 * t + feat_of(r, h) < XD;
                    const float a = acc[r] + b5v[r];
                    const float xe = xs[r] * P::exp_(-a);
                    const float term = xe - P::log_(xs[r] + g.elbo_eps) + a - 1.f;
                    rec_lane += ok ? term : 0.f;
                    da[r] = ok ? (1.f - xe) * invB_l : 0.f;
                }
            }
            if (t == 4) DVAE_FSTAMP(25);
            put_lds<P>(da, U, LDU, 32 * t, l31, h);
            if (t == 4) DVAE_FSTAMP(26);
        };
        if (P::XFULL) {
            // the fp32 x tile is resident in LDS ([frame][513], odd stride: conflict-free): no barriers here.
            // 16 full tiles = 4 per wave.  The 17th tile holds ONE real feature (bin 512): a whole MFMA tile and
            // epilogue round for it would be a fifth round for wave 0; wave 3 does it as a 128-term dot product instead.
#pragma unroll 1
            for (int t = wave_u; t < NT_OUT - 1; t += 4) out_tile(t, Xt, XD, 32 * t, XD - 1);
            if (wave_u == 3) {
                const float* wl = Bias + OB5 + NO + 64 * h;                 // this half's 64 weights (LDS broadcast reads)
                const T* drow = Hb + l31 * LDH + 64 * h;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 64 / E; ++c) {
                    const typename P::Frag dv = *reinterpret_cast<const typename P::Frag*>(drow + c * E);
#pragma unroll
                    for (int j = 0; j < E; j += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + c * E + j);
                        s = fmaf((float)dv[j], wv[0], s); s = fmaf((float)dv[j + 1], wv[1], s);
                        s = fmaf((float)dv[j + 2], wv[2], s); s = fmaf((float)dv[j + 3], wv[3], s);
                    }
                }
                s += __shfl_xor(s, 32, 64);
                const float a = s + Bias[OB5 + XD - 1];
                const float xv512 = Xt[l31 * XD + XD - 1];
                const float xe = xv512 * P::exp_(-a);
                if (h == 0) rec_lane += xe - P::log_(xv512 + g.elbo_eps) + a - 1.f;
                const float da512 = live ? (1.f - xe) * g.invB : 0.f;
                // columns 512 .. 543 of this frame's da row: the value, then 31 zeros (16 bf16 = 2 fragments per half)
                typename P::Frag z0, z1;
#pragma unroll
                for (int j = 0; j < E; ++j) { z0[j] = P::cvt(0.f); z1[j] = P::cvt(0.f); }
                if (h == 0) z0[0] = P::cvt(da512);
                T* urow = U + l31 * LDU + (XD - 1) + 16 * h;
                *reinterpret_cast<typename P::Frag*>(urow) = z0;
                *reinterpret_cast<typename P::Frag*>(urow + E) = z1;
            }
            __syncthreads();
        } else {
#pragma unroll 1
            for (int it = 0; it < (NT_OUT + 3) / 4; ++it) {
                xt_commit(xr, Xt, LDX, b0, g.B, 128 * it, tl);
                __syncthreads();
                if (it + 1 < (NT_OUT + 3) / 4) xt_issue(g.x, g.ldx, rowof, 128 * (it + 1), xr, tl);
                const int t = 4 * it + wave_u;
                if (t < NT_OUT) out_tile(t, Xt, LDX, 32 * wave, 127);
                __syncthreads();
            }
        }

        DVAE_STAMP(9);
        // ---------------- backward: d2 <- da ----------------
        zero_acc<P>(acc);
        gemm_block<P, NO / KS>(acc, w5t, wrs, W5tr, Ur, S4, [&]() {
            for (int t = wave; t < NT_OUT; t += 4) stash_tile<P>(U, LDU, 32 * t, (g.ablate & 1) ? nullptr : (T*)g.daT + (int64_t)(t) * 32 * g.Bp, g.spl, b0, l31, h);
        });
        WPre<P, HD / KS, P::PRE128> w4t;
        wprefetch<P, HD / KS>(w4t, wrs, W4tr, S4);
        float dv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - d2r[r] * d2r[r]);
        put_lds<P>(dv, Ha, LDH, fb, l31, h);
        __syncthreads();

        DVAE_STAMP(10);
        // ---------------- backward: d1 <- dpre_d2 ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, w4t, wrs, W4tr, Har, S4, [&]() { stash_tile<P>(Ha, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.dd2T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h); });
        WPre<P, HD / KS, P::PRE128> w3zt;
        WPre<P, 32 / KS> wmvt;
        if (wave == 0) wprefetch<P, HD / KS>(w3zt, wrs, W3ztr, S1);
        wprefetch<P, 32 / KS>(wmvt, wrs, Wmvtr, S4);
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - d1r[r] * d1r[r]);
        put_lds<P>(dv, Hb, LDH, fb, l31, h);
        __syncthreads();

        DVAE_STAMP(11);
        // ---------------- backward: z <- dpre_d1 (wave 0), then dmu / dlogvar ----------------
        if (wave == 0) {
            zero_acc<P>(acc);
            gemm_block<P, HD / KS>(acc, w3zt, wrs, W3ztr, Hbr, S1, [&]() { stash_tile<P>(Hb, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.dd1T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h); });
            float dml[16];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float dz = INFO ? acc[r] - g.beta * dzu[r] : acc[r];     // enc_loss = ELBO + alpha*clf - beta*BCE(aux(z), y)
                dml[r] = live ? dz + mu_r[r] * g.invB : 0.f;                                                   // dmu
                dml[r + 8] = live ? dz * ep_r[r] * (0.5f * sd_r[r]) - 0.5f * g.invB * (1.f - P::exp_(lv_r[r])) : 0.f;   // dlogvar
            }
            put_lds<P>(dml, Zb, LDZ, 0, l31, h);
        } else {
            stash_tile<P>(Hb, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.dd1T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h);
        }
        __syncthreads();

        DVAE_STAMP(12);
        // ---------------- backward: h2 <- [dmu | dlogvar] ----------------
        zero_acc<P>(acc);
        gemm_block<P, 32 / KS>(acc, wmvt, wrs, Wmvtr, Zbr, S4, [&]() { if (wave == 0) stash_tile<P>(Zb, LDZ, 0, (T*)g.dmlvT, g.spl, b0, l31, h); });
        WPre<P, HD / KS, P::PRE128> w2t;
        wprefetch<P, HD / KS>(w2t, wrs, W2tr, S4);
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - h2r[r] * h2r[r]);
        put_lds<P>(dv, Ha, LDH, fb, l31, h);
        __syncthreads();

        DVAE_STAMP(13);
        // ---------------- backward: h1 <- dpre_h2 (inputs are data: stop here) ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, w2t, wrs, W2tr, Har, S4, [&]() { stash_tile<P>(Ha, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.dh2T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h); });
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - h1r[r] * h1r[r]);
        put_lds<P>(dv, Hb, LDH, fb, l31, h);
        stash_tile<P>(Hb, LDH, fb, (g.ablate & 1) ? nullptr : (T*)g.dh1T + (int64_t)(wave) * 32 * g.Bp, g.spl, b0, l31, h);

        DVAE_STAMP(14);
        // ---------------- per-tile loss sums ----------------
        if (!live) rec_lane = 0.f;
        const float rs = wave_sum(rec_lane), ks = wave_sum(kl_lane);
        if (lane == 0) { red[wave] = rs; red[4 + wave] = ks; }
        if (INFO && wave == 0) {
            const float bcs = wave_sum(h == 0 ? bce_c : 0.f), bas = wave_sum(h == 0 ? bce_a : 0.f);
            if (lane == 0) { red[8] = bcs; red[9] = bas; }
        }
        __syncthreads();
        if (tid == 0) {
            tot_rec += (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
            tot_kl += -0.5 * (double)red[4];
            if (INFO) { tot_bc += (double)red[8]; tot_ba += (double)red[9]; }
        }
        __syncthreads();
    }
    DVAE_STAMP(15);
    if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + 31] = clock64();
    if (tid == 0) {
        g.partials[4 * blockIdx.x] = tot_rec;
        g.partials[4 * blockIdx.x + 1] = tot_kl;
        g.partials[4 * blockIdx.x + 2] = tot_bc;
        g.partials[4 * blockIdx.x + 3] = tot_ba;
    }
}

// ---------------------------------------------------------------------------------------------
// One wave = one 2x2 group of 32x32 MFMA tiles (64 output features x 64 input features of one
// layer): per k-step it loads 2 + 2 operand fragments and issues 4 MFMAs, halving the bytes per
// FLOP of a single-tile wave.  Missing halves (odd tile counts, 16-row heads) are null.
struct GroupDesc {
    const void* A[2];        // stash rows of dPre^T (32 output features each); A[1] may be null
    const void* Bm[2];       // stash rows of In^T (32 input features each); Bm[1] may be null
    int64_t out_off[2][2];   // float offset of tile (i, j) element (0, 0) in a gradient slab
    int64_t bias_off[2];     // float offset of the bias gradient rows of A block i, -1 = none
    int32_t ldo[2];          // row stride of the destination tensor of A block i
    int32_t mvalid[2];
    int32_t nvalid[2];
    int32_t split16;         // A block 0 holds two 16-row tensors (mu | log_var heads): rows >= 16 go to the *_hi targets
    int32_t ldo_hi;
    int64_t out_off_hi[2];
    int64_t bias_off_hi;
};

template <typename P, bool A1, bool B1>
__device__ __forceinline__ void wgrad_body(const GroupDesc& d, int64_t kbeg, int64_t kend, int64_t Bp, int64_t spl, float* __restrict__ slab,
                                           int l31, int h) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    constexpr int E = P::E, KS = P::KSTEP, NP = P::NP;
    const int lane = h * 32 + l31;
    constexpr int FB = 64 * E;                              // elements per (feature tile, k-step) block
    const T* a0p = (const T*)d.A[0] + lane * E;
    const T* a1p = A1 ? (const T*)d.A[1] + lane * E : a0p;
    const T* b0p = (const T*)d.Bm[0] + lane * E;
    const T* b1p = B1 ? (const T*)d.Bm[1] + lane * E : b0p;
    const bool bias0 = d.bias_off[0] >= 0, bias1 = A1 && d.bias_off[1] >= 0;
    f32x16 c00, c01, c10, c11, cb0, cb1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c00[i] = 0.f; c01[i] = 0.f; c10[i] = 0.f; c11[i] = 0.f; cb0[i] = 0.f; cb1[i] = 0.f; }
    const Frag one = P::ones();
    const int64_t sbeg = kbeg / KS, send = kend / KS;       // k-steps of this frame slice
    // The stash was written once by the previous kernel: these are cold HBM/MALL reads (~2 us round trip).
    // A ring of RD k-steps per operand keeps 4 * RD (x planes) 1-KB loads in flight per wave; the slot an MFMA group has
    // consumed is re-requested RD steps ahead (clamped on the last lap: a harmless reload, no branch).
    constexpr int RD = P::WRING;
    Frag a0[RD][NP], a1[RD][NP], b0[RD][NP], b1[RD][NP];
    auto ldp = [&](Frag (&f)[NP], const T* p, int64_t sk) {
        f[0] = *reinterpret_cast<const Frag*>(p + sk * FB);
        if constexpr (NP == 2) f[1] = *reinterpret_cast<const Frag*>(p + spl + sk * FB);
    };
#pragma unroll
    for (int i = 0; i < RD; ++i) {
        int64_t sk = sbeg + i; sk = sk < send ? sk : send - 1;
        ldp(a0[i], a0p, sk);
        ldp(b0[i], b0p, sk);
        if (A1) ldp(a1[i], a1p, sk);
        if (B1) ldp(b1[i], b1p, sk);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int64_t sk = sbeg; sk < send; sk += RD) {
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            if (sk + i < send) {                            // wave-uniform: slices are multiples of RD steps except the tail
                mmap<P>(c00, a0[i], b0[i]);
                if (B1) mmap<P>(c01, a0[i], b1[i]);
                if (A1) mmap<P>(c10, a1[i], b0[i]);
                if (A1 && B1) mmap<P>(c11, a1[i], b1[i]);
                if (bias0) { P::mma(cb0, a0[i][0], one); if constexpr (NP == 2) P::mma(cb0, a0[i][1], one); }
                if (A1) { if (bias1) { P::mma(cb1, a1[i][0], one); if constexpr (NP == 2) P::mma(cb1, a1[i][1], one); } }
            }
            int64_t sn = sk + RD + i; sn = sn < send ? sn : send - 1;
            ldp(a0[i], a0p, sn);
            ldp(b0[i], b0p, sn);
            if (A1) ldp(a1[i], a1p, sn);
            if (B1) ldp(b1[i], b1p, sn);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = feat_of(r, h);
        if (row < d.mvalid[0]) {
            const bool hi = d.split16 && row >= 16;
            const int rr = hi ? row - 16 : row;
            const int ldo = hi ? d.ldo_hi : d.ldo[0];
            if (l31 < d.nvalid[0]) slab[(hi ? d.out_off_hi[0] : d.out_off[0][0]) + (int64_t)rr * ldo + l31] = c00[r];
            if (B1) { if (l31 < d.nvalid[1]) slab[(hi ? d.out_off_hi[1] : d.out_off[0][1]) + (int64_t)rr * ldo + l31] = c01[r]; }
            if (bias0 && l31 == 0) slab[(hi ? d.bias_off_hi : d.bias_off[0]) + rr] = cb0[r];
        }
        if (A1) {
            if (row < d.mvalid[1]) {
                if (l31 < d.nvalid[0]) slab[d.out_off[1][0] + (int64_t)row * d.ldo[1] + l31] = c10[r];
                if (B1) { if (l31 < d.nvalid[1]) slab[d.out_off[1][1] + (int64_t)row * d.ldo[1] + l31] = c11[r]; }
                if (bias1 && l31 == 0) slab[d.bias_off[1] + row] = cb1[r];
            }
        }
    }
}

// grid.x = workgroups * ksplit with the k-slice as the FAST index: consecutive workgroups (dealt
// round-robin to the 8 XCDs) work on different frame slices, so each XCD's L2 mostly holds one
// slice of the stash.  blockDim.x / 64 groups per workgroup.
#ifndef DVAE_WGRAD_OCC
#define DVAE_WGRAD_OCC 1
#endif
template <typename P>
__global__ __launch_bounds__(256, DVAE_WGRAD_OCC) void wgrad_kernel(const GroupDesc* __restrict__ groups, int ngroups, int ksplit, int64_t Bp,
                                                    int64_t spl, int64_t kper, float* __restrict__ slabs, int64_t slab_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int slice = blockIdx.x % ksplit, wg = blockIdx.x / ksplit;
    const int gi = wg * (blockDim.x >> 6) + wave;
    if (gi >= ngroups) return;
    const GroupDesc d = groups[gi];
    const int64_t kbeg = (int64_t)slice * kper;
    int64_t kend = kbeg + kper;
    if (kend > Bp) kend = Bp;
    float* slab = slabs + (int64_t)slice * slab_stride;
    const bool a1 = d.A[1] != nullptr, b1 = d.Bm[1] != nullptr;
    if (a1 && b1) wgrad_body<P, true, true>(d, kbeg, kend, Bp, spl, slab, l31, h);
    else if (a1) wgrad_body<P, true, false>(d, kbeg, kend, Bp, spl, slab, l31, h);
    else if (b1) wgrad_body<P, false, true>(d, kbeg, kend, Bp, spl, slab, l31, h);
    else wgrad_body<P, false, false>(d, kbeg, kend, Bp, spl, slab, l31, h);
}

// ---------------------------------------------------------------------------------------------
// Weight gradients, workgroup-blocked: one 256-thread workgroup owns a 4 x 4 block of 32 x 32 tiles (128 output x 128 input
// features of one layer), wave (wr, wc) the 2 x 2 group {2wr, 2wr+1} x {2wc, 2wc+1} of it.  The block's 4 + 4 operand tiles
// are staged ONCE per k-step in LDS and shared by the four waves: the fragment-major stash tile of one k-step is 1 KB in
// exactly the lane-linear order a direct-to-LDS load writes (LDS address = wave-uniform base + lane * 16), so a fragment
// costs one `global_load_lds_dwordx4` and no registers.  A ring of NSTG stages x 2 k-steps keeps (NSTG - 1) stages in flight
// across one raw workgroup barrier per stage (counted vmcnt, never 0: cdna_hip_programming.md "Pipelining across barriers").
// Against the register-ring kernel above (every wave loads its own 2 + 2 fragments: each stash line crosses L2 -> CU 3.7
// times) the operand traffic halves and the in-flight bytes no longer cost registers.
struct BlockDesc {
    const void* At[4];       // the block's A tiles (null = absent)
    const void* Bt[4];
    GroupDesc g[4];          // per wave (wr * 2 + wc): destinations of its 2 x 2 group; A[0] == null: nothing to do
};

#ifdef DVAE_DIAG
template <typename P> struct WgLds {
    static constexpr int KPS = 2;                                   // k-steps per stage
    static constexpr int NSTG = 4;
    static constexpr int FRAG = 1024;                               // bytes of one (tile, plane, k-step) fragment block
    static constexpr int STAGE = 8 * P::NP * KPS * FRAG;
    static constexpr int BYTES = NSTG * STAGE;
    static constexpr int LOADS = 2 * P::NP * KPS;                   // direct-to-LDS loads per wave and stage (one A slot + one B slot)
};

template <typename P>
__global__ __launch_bounds__(256, 1) void wgrad_lds_kernel(const BlockDesc* __restrict__ blocks, int nblocks, int ksplit, int64_t Bp,
                                                           int64_t spl, int64_t kper, float* __restrict__ slabs, int64_t slab_stride) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    typedef WgLds<P> W;
    constexpr int E = P::E, KS = P::KSTEP, NP = P::NP, KPS = W::KPS, NSTG = W::NSTG;
    constexpr int FB = 64 * E;
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int slice = blockIdx.x % ksplit, bi = blockIdx.x / ksplit;
    if (bi >= nblocks) return;
    const BlockDesc& bd = blocks[bi];
    const GroupDesc d = bd.g[wave];
    const int64_t kbeg = (int64_t)slice * kper;
    int64_t kend = kbeg + kper;
    if (kend > Bp) kend = Bp;
    const int64_t sbeg = kbeg / KS, send = kend / KS;                 // k-steps of this frame slice
    const int nst = (int)((send - sbeg + KPS - 1) / KPS);             // stages
    float* slab = slabs + (int64_t)slice * slab_stride;
    // this wave stages tile slots `wave` (an A tile) and 4 + `wave` (a B tile); absent tiles reload the block's first A tile so
    // that every wave issues the same number of loads per stage (the vmcnt counts below are immediates)
    const T* src[2];
    src[0] = (const T*)(bd.At[wave] ? bd.At[wave] : bd.At[0]);
    src[1] = (const T*)(bd.Bt[wave] ? bd.Bt[wave] : bd.At[0]);
    auto issue = [&](int st) {                                        // stage st -> ring slot st % NSTG
        char* base = wsm + (st % NSTG) * W::STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    int64_t sk = sbeg + (int64_t)st * KPS + kk; sk = sk < send ? sk : send - 1;   // past the slice: a harmless reload into a free slot
                    const T* gp = src[q] + pl * spl + sk * FB + lane * E;
                    char* lp = base + (((q * 4 + wave) * NP + pl) * KPS + kk) * W::FRAG;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
                }
    };
    f32x16 c00, c01, c10, c11, cb0, cb1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c00[i] = 0.f; c01[i] = 0.f; c10[i] = 0.f; c11[i] = 0.f; cb0[i] = 0.f; cb1[i] = 0.f; }
    const bool have = d.A[0] != nullptr;
    const bool A1 = d.A[1] != nullptr, B1 = d.Bm[1] != nullptr;
    const bool bias0 = have && d.bias_off[0] >= 0, bias1 = A1 && d.bias_off[1] >= 0;
    const Frag one = P::ones();
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int st = 0; st < NSTG - 1; ++st) issue(st);
    for (int st = 0; st < nst; ++st) {
        // stage st has landed for this wave's loads once at most (NSTG - 2) later stages are outstanding; the barrier extends that
        // to every wave's loads and says that everybody has finished reading stage st - 1, whose slot the next issue overwrites
        if constexpr (W::LOADS * (NSTG - 2) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (W::LOADS * (NSTG - 2) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        issue(st + NSTG - 1);
        const char* base = wsm + (st % NSTG) * W::STAGE;
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk) {
            if (sbeg + (int64_t)st * KPS + kk >= send) break;         // wave-uniform tail
            auto frag = [&](int slot, int pl) -> Frag {
                return *reinterpret_cast<const Frag*>(base + ((slot * NP + pl) * KPS + kk) * W::FRAG + lane * 16);
            };
            Frag a0[NP], a1[NP], b0[NP], b1[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) { a0[pl] = frag(2 * wr, pl); a1[pl] = frag(2 * wr + 1, pl); b0[pl] = frag(4 + 2 * wc, pl); b1[pl] = frag(5 + 2 * wc, pl); }
            if (have) {
                mmap<P>(c00, a0, b0);
                if (B1) mmap<P>(c01, a0, b1);
                if (A1) mmap<P>(c10, a1, b0);
                if (A1 && B1) mmap<P>(c11, a1, b1);
                if (bias0) { P::mma(cb0, a0[0], one); if constexpr (NP == 2) P::mma(cb0, a0[1], one); }
                if (bias1) { P::mma(cb1, a1[0], one); if constexpr (NP == 2) P::mma(cb1, a1[1], one); }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the clamped tail loads must land before the LDS is released
    if (!have) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = feat_of(r, h);
        if (row < d.mvalid[0]) {
            const bool hi = d.split16 && row >= 16;
            const int rr = hi ? row - 16 : row;
            const int ldo = hi ? d.ldo_hi : d.ldo[0];
            if (l31 < d.nvalid[0]) slab[(hi ? d.out_off_hi[0] : d.out_off[0][0]) + (int64_t)rr * ldo + l31] = c00[r];
            if (B1) { if (l31 < d.nvalid[1]) slab[(hi ? d.out_off_hi[1] : d.out_off[0][1]) + (int64_t)rr * ldo + l31] = c01[r]; }
            if (bias0 && l31 == 0) slab[(hi ? d.bias_off_hi : d.bias_off[0]) + rr] = cb0[r];
        }
        if (A1) {
            if (row < d.mvalid[1]) {
                if (l31 < d.nvalid[0]) slab[d.out_off[1][0] + (int64_t)row * d.ldo[1] + l31] = c10[r];
                if (B1) { if (l31 < d.nvalid[1]) slab[d.out_off[1][1] + (int64_t)row * d.ldo[1] + l31] = c11[r]; }
                if (bias1 && l31 == 0) slab[d.bias_off[1] + row] = cb1[r];
            }
        }
    }
}
#endif  // DVAE_DIAG

// ---------------------------------------------------------------------------------------------
// Weight gradients, third form (default): one 256-thread workgroup = one 4 x 4 block of 32 x 32 tiles (128 output x 128 input
// features of a layer) x one frame slice, ONE workgroup per CU.  EVERY wave owns the whole 4 x 4 block (256 accumulator
// registers: the kernel runs at one wave per SIMD and has 512) on a QUARTER of the slice's frames: per k-step a wave loads 4 + 4
// operand fragments and issues 16 tile products, so a byte pulled into the CU feeds twice the MFMAs of the 2 x 2 register-ring
// kernel above (the measured bound there: ~55 GB/s of operand fragments per CU, 335 MB per launch, at 2.6 x the MFMA time).
// The four partial blocks meet in LDS as a reduce-scatter in a fixed order (deterministic): wave w finishes and stores A row w.  Bias gradients are in-lane sums of the A fragments (a frame
// sum needs no MFMA: one fp32 register per A tile instead of a 16-register accumulator against a constant-one operand).
struct Block4 {
    const void* At[4];       // stash rows of dPre^T, 32 output features each (null = absent; tiles are contiguous from 0)
    const void* Bt[4];       // stash rows of In^T, 32 input features each
    int64_t a_off[4];        // float offset, in a gradient slab, of (row 0 of A tile i, column 0) of its tensor
    int64_t bias_off[4];     // float offset of the bias-gradient rows of A tile i, -1 = none (only a layer's first B column carries them)
    int32_t ldo[4];          // row stride of A tile i's tensor
    int32_t mvalid[4];
    int32_t bcol[4];         // column of B tile j in the tensor
    int32_t nvalid[4];
    int32_t split16;         // A tile 0 holds two 16-row tensors (mu | log_var heads): rows >= 16 go to the *_hi targets
    int32_t ldo_hi;
    int64_t a_off_hi;
    int64_t bias_off_hi;
    // B tiles that are columns of the step's INPUTS (x: raw = 1, labels: raw = 2): the kernel can take them straight from the fp32
    // input matrix (the rows kernel then writes no stash for them); blocks never mix input tiles with stash tiles
    int32_t raw, rncols;     // rncols: columns of the input matrix (513 / y_dim)
    int32_t rcol[4];         // first column of B tile j in the input matrix
    int32_t wt[4], bt[4], wt_hi, bt_hi;   // tensor numbers of A tile i's weight / bias rows (and of the rows >= 16 of a split tile): fold_tail
    int32_t layer, pad_;                   // host side (w4_schedule): blocks of one emit() call share their A or their B tiles
};

// One workgroup of wgrad4_kernel = one item: block `block` over frames [kbeg, kend) into gradient slab `slice`.  The table is built on the
// host (w4_build_items): which slices a block is cut into, and which XCD a workgroup index lands on, are scheduling decisions the kernel
// only reads.  block < 0: an empty slot of the grid.
struct W4Item { int32_t block, slice; int64_t kbeg, kend; };
constexpr int W4_MAX_ITEMS = 4096;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for_w(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_w<I + 1, N>(f); }
}

// gradient-slab stores of the workgroup k-split kernel as buffer stores with cache-policy bits (W4_SLAB_AUX: gfx950 buffer aux, 0 = plain,
// 16 = sc1 = write-through -- the slabs are read by the apply kernel: 73.1 -> 72.9 us per step, same box, alternating; non-temporal
// loads of the once-read B fragments, also tried: 26.8 -> 31.6 us for the kernel)
#ifndef W4_SLAB_AUX
#define W4_SLAB_AUX 16
#endif
// the ragged-tile and bias stores of the slabs: write-through like the full-tile buffer stores (the folded optimizer tail reads the
// slabs of other workgroups of the same launch and relies on every slab store being one)
__device__ __forceinline__ void slab_store(float* p, float v) {
#if W4_SLAB_AUX
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p = v;
#endif
}
template <typename P> struct Wg4 {
#ifndef DVAE_W4RING_X3
#define DVAE_W4RING_X3 2      // round 5, same box, alternating, three rounds: 25.6 us (2) against 27.2 (3) by hipEvent -- 32 instead of 48 KB per wave in flight
#endif
#ifndef DVAE_W4RING
#define DVAE_W4RING 4
#endif
    static constexpr int RD = P::NP == 2 ? DVAE_W4RING_X3 : DVAE_W4RING;      // k-steps of operand fragments in flight per wave
    static constexpr size_t BYTES = (size_t)(8 * 16) * 64 * 16 + 12 * 64 * 4;           // 8 exchange slots of one A row (4 tiles x 16 registers x 64 lanes x 4 B) + the bias sums
};

// NA x NB = tiles of the block this instantiation computes (absent tiles alias tile 0 and are masked at the store: their
// descriptors carry mvalid / nvalid 0).  Compile-time shapes keep every operand load unconditional: a load under a run-time
// branch makes hipcc's wait-count pass fall back to vmcnt(0) in front of the first MFMA of every k-step (the ring then holds one).
struct RawIn { const float* x; const float* y; int ldx, ldy; int64_t B; };

// BLO = false: the B tiles (labels) have no lo plane in this launch -- it is neither read nor multiplied
// BIAS = false: no A tile of the block carries bias rows (only a layer's first B column does): no frame sums of the A fragments
template <typename P, int NA, int NB, int RAW, bool BLO = true, bool BIAS = true>
__device__ __forceinline__ void wgrad4_body(const Block4& bd, const Block4* __restrict__ bdg, char* wsm, int slice, int64_t Bp, int64_t spl, int64_t kbeg, int64_t kend_,
                                            float* __restrict__ slabs, int64_t slab_stride, int lane, int wave, const RawIn& ri) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    typedef Wg4<P> W;
    typedef const __attribute__((address_space(1))) char* gptr;
    typedef const __attribute__((address_space(1))) Frag* gfrag;
    // RAW: 0 = B tiles from the stash; 1 = from the fp32 input matrix by dword loads (any shape); 2 = from the input matrix through
    // this wave's LDS staging rows (four full tiles): 16 frames x 128 columns per k-step arrive as eight 1 KB row loads, are split
    // into (hi, lo) bf16, written as [frame][column] rows and read back transposed (ds_read_b64_tr_b16) into MFMA fragments.
    // Mode 1 needs 32 loads per k-step and overruns the 6-bit vmcnt (at most 63 loads in flight: 1.5 k-steps); mode 2 needs 16.
    constexpr int E = P::E, KS = P::KSTEP, NP = P::NP, RD = RAW == 2 ? 2 : W::RD;
    constexpr int64_t FBB = 64 * 16;                                     // bytes of one (feature tile, k-step) fragment block
    constexpr int SLD = 128 + 8, SPL = 16 * SLD;                         // staging rows: elements per frame row (odd number of 16-byte slots), per plane
    const int l31 = lane & 31, h = lane >> 5;
    // this wave's quarter of the slice's k-steps
    int64_t kend = kend_;
    if (kend > Bp) kend = Bp;
    const int64_t s0 = kbeg / KS, s1 = kend / KS;
    const int64_t nq = (s1 - s0 + 3) / 4;
    int64_t sbeg = s0 + (int64_t)wave * nq, send = sbeg + nq;
    if (send > s1) send = s1;
    if (sbeg > send) sbeg = send;
    gptr ap[NA], bp[NB];
    // NA == 4: wave w holds the A tiles rotated by w (local row i = A tile (i + w) % 4), so that "the row this wave finishes and
    // stores" is local row 0 for every wave and the reduce-scatter below is ONE instruction stream with compile-time register indices
    const int rot = NA == 4 ? wave : 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) { const void* q = NA == 4 ? bdg->At[(k + rot) & 3] : bd.At[k]; ap[k] = (gptr)(uintptr_t)(q ? q : bd.At[0]); }   // dynamic index: from the global copy (a scalar load), not a private-memory copy of bd
#pragma unroll
    for (int k = 0; k < NB; ++k) bp[k] = (gptr)(uintptr_t)(bd.Bt[k] ? bd.Bt[k] : bd.Bt[0]);
    const int64_t plb = spl * (int64_t)sizeof(T);                          // bytes between the hi and lo planes
    const unsigned loff = (unsigned)lane * 16u;
    f32x16 c[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    Frag a[RD][NA][NP], b[RAW == 1 ? 1 : (RAW == 2 ? 1 : RD)][RAW == 1 ? 1 : NB][NP];   // RAW == 2: b[0] = the fragments of the k-step being multiplied
    // RAW: the B fragments come from the fp32 input matrix itself -- lane (feature l31, frame half h) of tile j needs E consecutive
    // frames of ONE column: E dword loads, each wave-instruction two 128-byte row segments; split into (hi, lo) when consumed.
    // Frames past the batch repeat its last row (their dPre operand is zero), pad columns repeat the last column (never stored).
    float braw[RAW == 1 ? RD : 1][RAW == 1 ? NB : 1][E];
    f32x4 rawq[RAW == 2 ? RD : 1][RAW == 2 ? 8 : 1];                        // RAW == 2: row 2u + (lane >> 5), columns 4 (lane & 31) .. + 3 of the k-step's block
    typedef const __attribute__((address_space(1))) float* gflt;
    const gflt rsrc = (gflt)(uintptr_t)(RAW ? (bd.raw == 1 ? ri.x : ri.y) : nullptr);
    const int rld = RAW ? (bd.raw == 1 ? ri.ldx : ri.ldy) : 0;
    int rcolv[RAW == 1 ? NB : 1];
    if constexpr (RAW == 1) {
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int cc = bd.rcol[k] + l31; rcolv[k] = cc < bd.rncols ? cc : bd.rncols - 1; }
    }
    T* const stg = reinterpret_cast<T*>(wsm) + wave * (SPL * NP);          // RAW == 2: this wave's staging rows
    const int rcol4 = RAW == 2 ? bd.rcol[0] + 4 * l31 : 0;
    auto load = [&](auto sc, int64_t sk) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const int64_t o = sk * FBB;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            a[s][k][0] = *(gfrag)(ap[k] + o + loff);
            if constexpr (NP == 2) a[s][k][1] = *(gfrag)(ap[k] + plb + o + loff);
        }
        if constexpr (RAW == 1) {
            const int64_t f0 = sk * KS + h * E;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                int64_t fr = f0 + e; fr = fr < ri.B ? fr : ri.B - 1;
                const gflt rowp = rsrc + fr * rld;
#pragma unroll
                for (int k = 0; k < NB; ++k) braw[s][k][e] = rowp[rcolv[k]];
            }
        } else if constexpr (RAW == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int64_t fr = sk * KS + 2 * u + h; fr = fr < ri.B ? fr : ri.B - 1;
                rawq[s][u] = reinterpret_cast<const __attribute__((address_space(1))) F4U*>(rsrc + fr * rld + rcol4)->v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                b[s][k][0] = *(gfrag)(bp[k] + o + loff);
                if constexpr (NP == 2 && BLO) b[s][k][1] = *(gfrag)(bp[k] + plb + o + loff);
            }
        }
    };
    auto fsum = [&](const Frag& f) __attribute__((always_inline)) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < E; ++q) t += (float)f[q];
        return t;
    };
    // RAW == 2: stage s of the raw ring -> (hi, lo) rows in LDS -> transposed fragments b[0][j]
    auto prepare = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        if constexpr (RAW == 2) {
            typedef typename P::Pack4 Pack4;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                Pack4 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ph[e] = P::cvt(rawq[s][u][e]); pl[e] = P::cvt(rawq[s][u][e] - (float)ph[e]); }
                T* const rowp = stg + (2 * u + h) * SLD + 4 * l31;
                *reinterpret_cast<Pack4*>(rowp) = ph;
                if constexpr (NP == 2) *reinterpret_cast<Pack4*>(rowp + SPL) = pl;
            }
            const int i16 = l31 & 15, q = i16 >> 2, pp = i16 & 3, cg = l31 >> 4;
            typedef short s16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int pln = 0; pln < NP; ++pln) {
                    const T* bpj = stg + pln * SPL + q * SLD + 32 * j + 16 * cg + 4 * pp;
                    const s16x4 r0 = lds_tr16(bpj + (8 * h) * SLD), r1 = lds_tr16(bpj + (8 * h + 4) * SLD);
                    const s16x8 raw8 = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                    b[0][j][pln] = __builtin_bit_cast(Frag, raw8);
                }
        }
    };
    // timing ablations of the main loop (tools/r05/ab_libs.sh on variants built by tools/r05/mkvariant.sh; results are wrong under any of them): W4_NOMFMA = loads + bias sums only,
    // W4_NOFSUM = no bias sums, W4_NOLOAD = the ring is never refilled (MFMAs on the prologue's fragments), W4_NOEPI = no reduce-scatter / stores
    auto compute = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
#ifdef W4_NOMFMA
        if constexpr (RAW == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                bs[i] += fsum(a[s][i][0]);
                if constexpr (NP == 2) bs[i] += fsum(a[s][i][1]);
#pragma unroll
                for (int j = 0; j < NB; ++j) { c[i][j][0] += (float)b[s][j][0][0]; if constexpr (NP == 2 && BLO) c[i][j][1] += (float)b[s][j][1][0]; }
            }
            return;
        }
#endif
        if constexpr (RAW == 2) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
#pragma unroll
                for (int j = 0; j < NB; ++j) mmap<P>(c[i][j], a[s][i], b[0][j]);
                if constexpr (BIAS) {
                    bs[i] += fsum(a[s][i][0]);
                    if constexpr (NP == 2) bs[i] += fsum(a[s][i][1]);
                }
            }
        } else if constexpr (RAW == 1) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                Frag bj[NP];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    bj[0][e] = P::cvt(braw[s][j][e]);
                    if constexpr (NP == 2) bj[1][e] = P::cvt(braw[s][j][e] - (float)bj[0][e]);
                }
#pragma unroll
                for (int i = 0; i < NA; ++i) mmap<P>(c[i][j], a[s][i], bj);
            }
            if constexpr (BIAS) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    bs[i] += fsum(a[s][i][0]);
                    if constexpr (NP == 2) bs[i] += fsum(a[s][i][1]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
#pragma unroll
                for (int j = 0; j < NB; ++j) mmap<P>(c[i][j], a[s][i], b[s][j], BLO);
#ifndef W4_NOFSUM
                if constexpr (BIAS) {
                    bs[i] += fsum(a[s][i][0]);                           // bias gradient: frame sum of the A fragment (VALU in the MFMAs' shadow)
                    if constexpr (NP == 2) bs[i] += fsum(a[s][i][1]);
                }
#endif
            }
        }
    };
    // No branch around the loop (an empty range runs zero laps; its clamped prologue loads re-read the slice's last k-step): a
    // conditional region here makes every accumulator a phi of (zero, loop result) and costs a 256-register copy.
    {
        const int64_t slast = (send > s0 ? send : s0 + 1) - 1;
        static_for_w<0, RD>([&](auto sc) {
            int64_t sk = sbeg + decltype(sc)::value; sk = sk < slast ? sk : slast;
            load(sc, sk);
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RAW == 2) prepare(std::integral_constant<int, 0>{});
#pragma unroll 1
        for (int64_t sk = sbeg; sk < send; sk += RD) {
            static_for_w<0, RD>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if (sk + s < send) compute(sc);                         // wave-uniform
                int64_t sn = sk + RD + s; sn = sn < slast ? sn : slast;     // last lap: a harmless reload, no branch around a load
                if constexpr (RAW == 2) {
                    // the MFMAs of this k-step are in the pipe (their operands are read): stage the NEXT k-step's B tiles in their shadow
                    // -- its raw values are consumed before this slot's reload below overwrites the ring
                    prepare(std::integral_constant<int, (s + 1) % RD>{});
                }
#ifndef W4_NOLOAD
                load(sc, sn);
#else
                (void)sn;
#endif
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        if constexpr (RAW == 2) __syncthreads();                        // every wave is done with its staging rows: the reduce-scatter below reuses the LDS
    }
    // ---- the four partial blocks meet in LDS: a reduce-scatter in a fixed order (deterministic).  Wave w ends up with A tile row
    // w of the block (its local row 0) and stores it: a single wave storing 8 tiles with per-element address arithmetic took
    // 17 us (issue-bound), more than the main loop.
#ifdef W4_NOEPI
    {
        float t = bs[0] + bs[1] + bs[2] + bs[3];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += c[i][j][r];
        if (t == 123.456f) slabs[0] = t;
        return;
    }
#endif
    constexpr int ROWQ = 4 * 4 * 64;                                      // f32x4 quads of one A row (4 tiles x 16 registers x 64 lanes)
    f32x4* const lds = reinterpret_cast<f32x4*>(wsm);
    float* const lbias = reinterpret_cast<float*>(lds + 8 * ROWQ);        // [dest wave][source order][lane]
    auto put_row = [&](auto ic, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                lds[slot * ROWQ + (j * 4 + q) * 64 + lane] = f32x4{c[i][j][4 * q], c[i][j][4 * q + 1], c[i][j][4 * q + 2], c[i][j][4 * q + 3]};
    };
    auto add_row0 = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = lds[slot * ROWQ + (j * 4 + q) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) c[0][j][4 * q + e] += v[e];
            }
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
    if constexpr (NA == 1) {
        // one A row: waves 1-3 hand it to wave 0
        if (wave != 0) { put_row(I0{}, wave); lbias[wave * 64 + lane] = bs[0]; }
        __syncthreads();
        if (wave != 0) return;
        add_row0(1); add_row0(2); add_row0(3);
        bs[0] += lbias[64 + lane]; bs[0] += lbias[128 + lane]; bs[0] += lbias[192 + lane];
    } else {
        // round A: local rows 1, 2 go to waves (w + 1) % 4, (w + 2) % 4 (slot = 2 * destination + source order); round B: local row 3
        const int d1 = (wave + 1) & 3, d2 = (wave + 2) & 3, d3 = (wave + 3) & 3;
        put_row(I1{}, 2 * d1); put_row(I2{}, 2 * d2 + 1);
        lbias[(3 * d1 + 0) * 64 + lane] = bs[1]; lbias[(3 * d2 + 1) * 64 + lane] = bs[2]; lbias[(3 * d3 + 2) * 64 + lane] = bs[3];
        __syncthreads();
        add_row0(2 * wave); add_row0(2 * wave + 1);                         // from wave (w - 1) % 4, then from wave (w - 2) % 4
        __syncthreads();
        put_row(I3{}, d3);
        __syncthreads();
        add_row0(wave);                                                   // from wave (w - 3) % 4
        bs[0] += lbias[(3 * wave + 0) * 64 + lane]; bs[0] += lbias[(3 * wave + 1) * 64 + lane]; bs[0] += lbias[(3 * wave + 2) * 64 + lane];
    }
    // ---- store local row 0 = A tile `rot` of the block (descriptor fields of that tile: wave-uniform scalar loads)
    float* const slab = slabs + (int64_t)slice * slab_stride;
    const int mv = bdg->mvalid[rot], ldo = bdg->ldo[rot];
    const int64_t a_off = bdg->a_off[rot], bias_off = bdg->bias_off[rot];
    const bool split = rot == 0 && bd.split16;
    if (mv == 32 && !split) {
        // full tile rows: one exec region per tile, scalar row address + a per-lane 32-bit offset.  (Round 3 tried 16-byte stores -- the
        // tile turned through this wave's LDS slot so that a lane holds four consecutive columns, 4 stores of 1 KB per tile instead of
        // 16 of 256 B: 28.1 us against 27.2 us for the kernel, same box, alternating.  The dword form stays.)
        const unsigned lo = (unsigned)(4 * h * ldo + l31);
#if W4_SLAB_AUX
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slab + a_off, 0, 0x7fffffff, 0x00020000);   // this A row's 32 tensor rows
#endif
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (l31 < bd.nvalid[j]) {
                float* const t0 = slab + a_off + bd.bcol[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#if W4_SLAB_AUX
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c[0][j][r]), srs, (int)(4u * lo), (int)(4u * (unsigned)(bd.bcol[j] + ((r & 3) + 8 * (r >> 2)) * ldo)), W4_SLAB_AUX);
#else
                    float* const rowp = t0 + (int64_t)((r & 3) + 8 * (r >> 2)) * ldo;     // wave-uniform
                    rowp[lo] = c[0][j][r];
#endif
                }
                (void)t0;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = feat_of(r, h);
                if (row < mv && l31 < bd.nvalid[j]) {
                    const bool hi = split && row >= 16;
                    const int rr = hi ? row - 16 : row;
                    const int ld = hi ? bd.ldo_hi : ldo;
                    slab_store(&slab[(hi ? bd.a_off_hi : a_off) + (int64_t)rr * ld + bd.bcol[j] + l31], c[0][j][r]);
                }
            }
        }
    }
    const float tot = bs[0] + __shfl_xor(bs[0], 32, 64);                  // the two frame halves of feature row l31
    if (h == 0 && l31 < mv && bias_off >= 0) {
        const bool hi = split && l31 >= 16;
        slab_store(&slab[hi ? bd.bias_off_hi + (l31 - 16) : bias_off + l31], tot);
    }
}

// ---------------------------------------------------------------------------------------------
// ---- the optimizer step folded into the tail of the weight-gradient kernel (a train step = two launches).
// Every (slice, block) workgroup, once its partial block is in its slab, arrives at the block's counter and waits until all `ksplit`
// slices of the block have arrived (the grid is one round of workgroups, all resident: the host folds only when the grid fits the CUs;
// the wait is bounded and raises the error word instead of hanging).  Then the ksplit * 4 waves share the block's parameters: a wave
// takes whole tensor rows (row u of the block's 128, u = wave id, + ksplit * 4, ...; units 128-131: the bias rows), a lane two elements
// of a row; each element = the slab sum in slab order (the very additions of apply_kernel), Adam, the weight-copy refresh.  Workgroup 0
// also turns the rows kernel's partial sums into the loss scalars.
// Counters (unsigned words of the flag header): [2] error (sticky), [16 + b] arrivals of block b -- never reset: launch number n of a
// workspace (counted by the host, fold_seq) waits for ksplit * n.  One fire-and-forget atomic and the polling loads are all the
// synchronisation a workgroup pays (returning atomics cost a device-scope round trip each: 2 us on the critical path).
constexpr int FOLD_MAXB = 120;
struct FoldArgs { unsigned* cnt; unsigned target; unsigned max_polls; };

#ifdef DVAE_DIAG
template <typename T, int NP>
__device__ __forceinline__ void fold_tail(const ApplyArgs& g, const FoldArgs& fa, const Block4& bd, const Block4* __restrict__ bdg, int bi, int slice,
                                          int ks, int lane, int wave, char* wsm) {
    int* const flag = reinterpret_cast<int*>(wsm);                     // the reduce-scatter is over: the exchange slots are free
    // This wave's slab stores are complete, i.e. visible device-wide: they are write-through (sc1) stores, so waiting for them is
    // enough.  (A release fence writes back the XCD's whole L2 and the matching acquire invalidates it -- 960 times per launch: the
    // kernel took 75 us instead of 28.)
#if W4_SLAB_AUX
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(fa.cnt + 16 + bi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned polls = 0;
        int ok = 1;
        while ((int)(__hip_atomic_load(fa.cnt + 16 + bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fa.target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > fa.max_polls) { ok = 0; break; }
        }
        if (!ok) __hip_atomic_store(fa.cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = ok;
    }
    __syncthreads();
    const int ok = flag[0];
#if !W4_SLAB_AUX
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
#ifndef FOLD_DIAG
#define FOLD_DIAG 0      // timing diagnostics (wrong results): 1 = wait only, 2 = wait + loads + stores of p only, 3 = no loss scalars
#endif
    if (ok && FOLD_DIAG != 1) {
        int nb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (bd.Bt[k] != nullptr) nb = k + 1;
        const int gw = slice * 4 + wave, nw = ks * 4;
        constexpr int UB = 4;                                          // units per batch: their loads are all in flight together
        for (int u0 = gw; u0 < 132; u0 += UB * nw) {
            int64_t idx[UB][2];
            float pi[UB][2], mo[UB][2], vo[UB][2], gi[UB][2];
            int tq[UB][2];                                             // wave-uniform: a unit is a row of ONE tensor (two for the heads' bias unit: one per q)
#pragma unroll
            for (int b = 0; b < UB; ++b) {
                const int u = u0 + b * nw;                             // wave-uniform
                idx[b][0] = idx[b][1] = -1;
                tq[b][0] = tq[b][1] = 0;
                if (u < 128) {
                    const int rot = u >> 5, rr = u & 31;
                    if (rr < bdg->mvalid[rot]) {
                        const bool hi = rot == 0 && bd.split16 && rr >= 16;
                        const int64_t base = hi ? bd.a_off_hi + (int64_t)(rr - 16) * bd.ldo_hi : bdg->a_off[rot] + (int64_t)rr * bdg->ldo[rot];
                        tq[b][0] = tq[b][1] = hi ? bd.wt_hi : bdg->wt[rot];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int j = (lane >> 5) + 2 * q, l = lane & 31;     // columns lane and lane + 64 of the block's 128
                            if (j < nb && l < bdg->nvalid[j]) idx[b][q] = base + bdg->bcol[j] + l;
                        }
                    }
                } else if (u < 132) {
                    const int rot = u - 128;
                    const int64_t bo = bdg->bias_off[rot];
                    const int mv = bdg->mvalid[rot];
                    if (bo >= 0) {
                        const bool split = rot == 0 && bd.split16;
                        tq[b][0] = bdg->bt[rot]; tq[b][1] = bd.bt_hi;
                        if (lane < (split ? 16 : 32) && lane < mv) idx[b][0] = bo + lane;
                        if (split && lane >= 16 && lane < 32 && lane < mv) idx[b][1] = bd.bias_off_hi + (lane - 16);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int64_t i = idx[b][q] >= 0 ? idx[b][q] : 0;  // masked lanes read element 0 (no branch around the loads)
                    pi[b][q] = g.p[i]; mo[b][q] = g.m[i]; vo[b][q] = g.v[i];
                    gi[b][q] = slab_sum_at<W4_SLAB_AUX != 0>(g, i);               // the other slices' slabs, not this XCD's stale L2 lines
                }
            }
#pragma unroll
            for (int b = 0; b < UB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const TensorDesc& d = g.tensors[tq[b][q]];                     // uniform address: scalar loads
                    if (FOLD_DIAG == 2) { if (idx[b][q] >= 0) g.p[idx[b][q]] = pi[b][q] + mo[b][q] + vo[b][q] + gi[b][q]; continue; }
                    if (idx[b][q] >= 0) apply_element<T, true, NP>(g, idx[b][q], d, pi[b][q], mo[b][q], vo[b][q], gi[b][q]);
                }
        }
    }
    if (blockIdx.x == 0 && g.losses3 != nullptr && FOLD_DIAG == 0) {   // workgroup 0 (always a participant): loss scalars
        __syncthreads();
        finalize_losses(g, reinterpret_cast<double (*)[4]>(wsm + 64));
        // a wait that ran out (error word set, sticky): parameters were not all updated -- the loss says so
        if (threadIdx.x == 0 && __hip_atomic_load(fa.cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) g.losses3[0] = __builtin_nanf("");
    }
    if (!ok && threadIdx.x == 0 && g.losses3 != nullptr) g.losses3[0] = __builtin_nanf("");
}
#endif  // DVAE_DIAG

template <typename P>
__global__ __launch_bounds__(256, 1) void wgrad4_kernel(const Block4* __restrict__ blocks, const W4Item* __restrict__ items, int ksplit, int64_t Bp,
                                                        int64_t spl, float* __restrict__ slabs, int64_t slab_stride,
                                                        const RawIn ri, int use_raw, const unsigned* __restrict__ ylo_epoch, unsigned launch_id,
                                                        const ApplyArgs fold_apply, const FoldArgs fold, int fin_block, const unsigned* fin_err) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // deferred optimizer step (apply_common.hpp): the step's loss scalars no longer come from an optimizer launch -- ONE extra workgroup of
    // this launch (it follows the rows kernel, whose partial sums are complete) reduces them, on a CU the weight-gradient blocks leave free
    if (fin_block >= 0 && (int)blockIdx.x == fin_block) {
        finalize_losses(fold_apply, reinterpret_cast<double (*)[4]>(wsm));
        if (threadIdx.x == 0 && fin_err != nullptr && __hip_atomic_load(fin_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            fold_apply.losses3[0] = __builtin_nanf("");              // a bounded wait of the rows kernel ran out: the update was not complete
        return;
    }
    // Workgroup -> (block, slice, frames): the host's item table (w4_build_items).  Workgroup i runs on XCD i % 8 (speed only): the table
    // keeps the items that read the same stash lines -- the blocks of a layer over the same frames -- on one XCD, and cuts every block into
    // as many slices as its cost per k-step asks for, so that all workgroups of the one round finish together.
    const W4Item it = items[blockIdx.x];
    if (it.block < 0) return;
    const int slice = it.slice, bi = it.block;
    const Block4 bd = blocks[bi];                                       // by value: wave-uniform, lives in SGPRs (a reference would be re-read after every slab store)
    int na = 0, nb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (bd.At[k] != nullptr) na = k + 1; if (bd.Bt[k] != nullptr) nb = k + 1; }
    // (BIAS = false bodies -- no frame sums of the A fragments in blocks that carry no bias rows -- exist as a template parameter and are NOT
    // instantiated: built in round 5, the kernel with them took 34.3 us against 25.3 without, same box, alternating (their loops are
    // tighter, 158 against 394 instructions per two k-steps, but hipcc spills 250 - 650 registers around them; tools/r05/w4_ab2.sh))
#define W4_GO_(NA_, NB_, RAW_, BLO_) wgrad4_body<P, NA_, NB_, RAW_, BLO_, true>(bd, blocks + bi, wsm, slice, Bp, spl, it.kbeg, it.kend, slabs, slab_stride, lane, wave, ri)
#define W4_GO(NA_, NB_, RAW_) W4_GO_(NA_, NB_, RAW_, true)
    bool raw = false;
    if constexpr (sizeof(typename P::T) == 2) raw = (use_raw & bd.raw) != 0;      // input-matrix B tiles (16-bit operand policies only); use_raw bit 0: x, bit 1: labels
    if (raw) {
        if constexpr (sizeof(typename P::T) == 2) {
            if (nb == 4 && bd.rcol[0] + 128 <= bd.rncols) W4_GO(4, 4, 2);      // four full tiles: through the LDS staging rows
            else if (nb == 1) W4_GO(4, 1, 1);
            else if (nb == 2) W4_GO(4, 2, 1);
            else W4_GO(4, 4, 1);
        }
    } else if (sizeof(typename P::T) == 2 && P::NP == 2 && bd.raw == 2 && ylo_epoch != nullptr && *ylo_epoch != launch_id) {
        // label-fed blocks of a launch whose label tiles all fit one bf16 plane (binary labels): hi plane only
        if (nb == 1) W4_GO_(4, 1, 0, false);
        else if (nb == 2) W4_GO_(4, 2, 0, false);
        else W4_GO_(4, 4, 0, false);
    } else if (na == 1) {
        if (nb == 1) W4_GO(1, 1, 0);
        else if (nb == 2) W4_GO(1, 2, 0);
        else W4_GO(1, 4, 0);
    } else {
        if (nb == 1) W4_GO(4, 1, 0);
        else if (nb == 2) W4_GO(4, 2, 0);
        else W4_GO(4, 4, 0);
    }
#undef W4_GO
#undef W4_GO_
#ifndef W4_FOLD
#ifdef DVAE_DIAG
#define W4_FOLD 1      // 0: the folded optimizer tail compiled out (A/B of what its presence costs the main loop)
#else
#define W4_FOLD 0      // product build: no folded tail
#endif
#endif
#if W4_FOLD
    if (fold.cnt != nullptr) fold_tail<typename P::T, P::NP>(fold_apply, fold, bd, blocks + bi, bi, slice, ksplit, lane, wave, wsm);
#endif
}

// sum of up to NS slabs at element i: every load issued before the first addition (a run-time loop makes each addition wait for
// its own load: ten dependent round trips), additions in slab order (deterministic)
template <int NS>
__device__ __forceinline__ float slab_total(const float* __restrict__ slabs, int64_t i, int nslabs, int64_t stride) {
    float part[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) part[k] = slabs[(int64_t)(k < nslabs ? k : 0) * stride + i];
    float t = part[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) if (k < nslabs) t += part[k];
    return t;
}
__device__ __forceinline__ float slab_total_any(const float* __restrict__ slabs, int64_t i, int nslabs, int64_t stride) {
    if (nslabs <= 8) return slab_total<8>(slabs, i, nslabs, stride);
    if (nslabs <= 16) return slab_total<16>(slabs, i, nslabs, stride);
    float s = slabs[i];
    for (int k = 1; k < nslabs; ++k) s += slabs[k * stride + i];
    return s;
}

// dst = (accumulate ? dst : 0) + sum of the slabs (fixed order: deterministic)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int64_t n, int nslabs, int64_t stride, float* __restrict__ dst, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = slab_total_any(slabs, i, nslabs, stride);
        dst[i] = accumulate ? dst[i] + s : s;
    }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(float* __restrict__ slabs, int64_t n, int nslabs, int64_t stride) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        slabs[i] = slab_total_any(slabs, i, nslabs, stride);
}

// One thread per parameter over the flat buffer (every load independent); chunk_tensor maps each
// 64-float chunk to its tensor (tensors start on 64-float boundaries), 255 = alignment padding.
// The block after the last parameter block finalises the loss scalars.
template <typename T, bool ADAM, int NP = 1>
__global__ __launch_bounds__(256) void apply_kernel(const ApplyArgs g) {
    if (blockIdx.x == gridDim.x - 1) {                    // loss finalisation block
        if (!ADAM || g.losses3 == nullptr) return;
        __shared__ double red[4][4];
        finalize_losses(g, red);
        return;
    }
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.n_params) return;
    // Everything this thread reads sits at flat index idx (alignment padding between tensors included, the buffers
    // are allocated whole): request it all FIRST, so the two dependent table lookups below (chunk -> tensor ->
    // descriptor) overlap the one HBM round trip instead of preceding it.
    const float pi = g.p[idx];
    float m_old = 0.f, v_old = 0.f, gi = 0.f;
    if (ADAM) {
        m_old = g.m[idx]; v_old = g.v[idx];
        gi = slab_sum_at(g, idx);
    }
    // a wave covers one 64-float chunk: its tensor and the descriptor are wave-uniform, fetched by scalar loads
    const int t = g.chunk_tensor[__builtin_amdgcn_readfirstlane((int)(idx >> 6))];
    if (t == 255) return;
    const TensorDesc d = g.tensors[t];
    apply_element<T, ADAM, NP>(g, idx, d, pi, m_old, v_old, gi);
}

#ifdef DVAE_DIAG
// The optimizer step by UNITS (apply_common.hpp: defer_unit -- 8 rows x 32 columns of one weight matrix per wave, whole-line loads of
// parameters / moments / slabs, the kernel-layout copies as whole 8- and 16-byte pieces through the transposing LDS read): the same
// element arithmetic as apply_kernel on the same slab sums (bit-identical, tested), a quarter of its instructions, no lone 2-byte stores.
// One unit per wave; the block after the last unit block finalises the loss scalars.  bf16 / bf16x3 copies.
template <typename T, int NP>
__global__ __launch_bounds__(256) void apply_units_kernel(const ApplyArgs g, const DeferTask* __restrict__ tasks, int nunits) {
    __shared__ __attribute__((aligned(16))) char sm[4 * DeferLds<T, NP>::wave_elems * sizeof(T) > 128 ? 4 * DeferLds<T, NP>::wave_elems * sizeof(T) : 128];
    if (blockIdx.x == gridDim.x - 1) {
        if (g.losses3 == nullptr) return;
        finalize_losses(g, reinterpret_cast<double (*)[4]>(sm));
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // unit un of the launch: consecutive units of a tile on different workgroups (a tile's four units share lines at odd row lengths)
    const int un = wave * ((int)gridDim.x - 1) + (int)blockIdx.x;
    if (un >= nunits) return;
    T* const tile = reinterpret_cast<T*>(sm) + wave * DeferLds<T, NP>::wave_elems;
    defer_unit<T, NP, false>(g, tasks[un >> 2], un & 3, tile, lane);
}
#endif  // DVAE_DIAG

// ---------------------------------------------------------------------------------------------
// host-side planning
struct Layout {
    // weight-copy buffer (elements of T)
    int64_t W1s, W2s, Wmvs, W3s, W4s, W5s, W5t, W4t, W3zt, Wmvt, W2t, wcopy_elems;
    int64_t Wc1s, Wc2s, Wc2t, Wa1s, Wa1t, Wa2s, Wa2t;                  // M2_info
    int64_t c1T, c2T, dc1T, dc2T, dc3T, a1T, a2T, da1T, da2T, da3T;     // M2_info stash
    bool info;
    int ld1, ld3, yp, ye, yd;
    // stash (rows of Bp elements)
    int64_t xT, yT, h1T, h2T, dh1T, dh2T, dmlvT, zT, d1T, d2T, dd1T, dd2T, daT, stash_rows;
    // workspace byte offsets
    int64_t o_tiles, o_blocks, o_blocks4, o_items4, o_tensors, o_chunks, o_partials, o_flags, o_defer, o_wcopy, o_stash, o_grads, total;
    int ntiles, nblocks, nblocks4;
};
// deferred optimizer step (apply_common.hpp): task table, then the arrival counters (DEFER_SHARDS lines of 128 bytes), then one line with
// the `done` counter (word 0) and the error word (word 1)
constexpr int DEFER_MAX_TASKS = 640;
constexpr int64_t DEFER_O_SHARD = (int64_t)DEFER_MAX_TASKS * (int64_t)sizeof(DeferTask);
constexpr int64_t DEFER_O_DONE = DEFER_O_SHARD + DEFER_SHARDS * 128;
constexpr int64_t DEFER_BYTES = DEFER_O_DONE + 128;

static inline int64_t al(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static inline bool is_bf(int precision) { return precision == DVAE_PREC_BF16 || precision == DVAE_PREC_BF16X3; }
static inline int planes_of(int precision) { return precision == DVAE_PREC_BF16X3 ? 2 : 1; }

static int make_layout(const dvae_train_plan_t& p, Layout& L) {
    const int esz = is_bf(p.precision) ? 2 : 4;
    const int np = planes_of(p.precision);
    L.yp = p.y_dim == 0 ? 0 : (p.y_dim + 15) / 16 * 16;
    L.ye = p.model == DVAE_MODEL_M2 ? L.yp : 0;
    L.info = p.model == DVAE_MODEL_M2_INFO;
    L.yd = L.yp;
    // K extents of the weight copies: 16-deep k-steps (x 513 -> 528, labels -> multiple of 16, z 16), or -- rows3 kernel, 32-deep
    // k-steps on 16 x 16 x 32 MFMA tiles -- x -> 544, labels -> multiple of 32, z -> 32
    const bool r3 = p.rows_kernel == 3;
    const int yk = p.y_dim == 0 ? 0 : (r3 ? (p.y_dim + 31) / 32 * 32 : L.yp);
    L.ld1 = (r3 ? 544 : XP) + (L.ye ? yk : 0);
    L.ld3 = (r3 ? 32 : ZD) + (L.yd ? yk : 0);
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += al(n, 128); return r; };
    L.W1s = take((int64_t)HD * L.ld1); L.W2s = take(HD * HD); L.Wmvs = take(32 * HD); L.W3s = take((int64_t)HD * L.ld3);
    L.W4s = take(HD * HD); L.W5s = take((int64_t)NO * HD); L.W5t = take((int64_t)HD * NO); L.W4t = take(HD * HD);
    L.W3zt = take(32 * HD); L.Wmvt = take(HD * 32); L.W2t = take(HD * HD);
    if (L.info) {
        L.Wc1s = take((int64_t)HD * XP); L.Wc2s = take(HD * HD); L.Wc2t = take(HD * HD);
        L.Wa1s = take(HD * ZD); L.Wa1t = take(32 * HD); L.Wa2s = take(HD * HD); L.Wa2t = take(HD * HD);
    }
    L.wcopy_elems = o;
    int64_t r = 0;
    auto rows = [&](int64_t n) { int64_t q = r; r += n; return q; };
    L.xT = rows(NO); L.yT = rows(L.yp ? al(L.yp, 32) : 0); L.h1T = rows(HD); L.h2T = rows(HD); L.dh1T = rows(HD); L.dh2T = rows(HD);
    L.dmlvT = rows(32); L.zT = rows(32); L.d1T = rows(HD); L.d2T = rows(HD); L.dd1T = rows(HD); L.dd2T = rows(HD); L.daT = rows(NO);
    if (L.info) {
        L.c1T = rows(HD); L.c2T = rows(HD); L.dc1T = rows(HD); L.dc2T = rows(HD); L.dc3T = rows(32);
        L.a1T = rows(HD); L.a2T = rows(HD); L.da1T = rows(HD); L.da2T = rows(HD); L.da3T = rows(32);
    }
    rows(32);   // slack
    L.stash_rows = r;
    // 2x2 groups of 32x32 tiles per job: (pairs of A blocks) x (pairs of B blocks)
    const int nty = L.yp ? (int)(al(L.yp, 32) / 32) : 0;
    auto pr = [](int n) { return (n + 1) / 2; };
    L.ntiles = 2 * pr(NT_OUT + (L.ye ? nty : 0)) + 2 * 2 + 1 * 2 + 2 * pr(1 + (L.yd ? nty : 0)) + 2 * 2 + pr(NT_OUT) * 2;
    if (L.info) L.ntiles += 2 * pr(NT_OUT) + 2 * 2 + 1 * 2 + 2 * 1 + 2 * 2 + 1 * 2;   // clf L1, L2, out; aux L1, L2, out
    // 4 x 4 blocks of the workgroup-blocked kernel: ceil(A tiles / 4) x ceil(B tiles / 4) per layer
    auto q4 = [](int n) { return (n + 3) / 4; };
    L.nblocks = q4(4) * q4(NT_OUT + (L.ye ? nty : 0)) + 1 + 1 + q4(4) * q4(1 + (L.yd ? nty : 0)) + 1 + q4(NT_OUT) * 1;
    if (L.info) L.nblocks += q4(NT_OUT) + 1 + 1 + 1 + 1 + 1;
    // blocks of the workgroup k-split kernel: as above, but a block never mixes input-matrix tiles (x, labels) with stash tiles
    L.nblocks4 = q4(4) * (q4(NT_OUT) + (L.ye ? q4(nty) : 0)) + 1 + 1 + q4(4) * (1 + (L.yd ? q4(nty) : 0)) + 1 + q4(NT_OUT) * 1;
    if (L.info) L.nblocks4 += q4(NT_OUT) + 1 + 1 + 1 + 1 + 1;
    int64_t b = 0;
    auto bytes = [&](int64_t n) { int64_t q = b; b += al(n, 256); return q; };
    L.o_tiles = bytes((int64_t)L.ntiles * sizeof(GroupDesc));
    L.o_blocks = bytes((int64_t)L.nblocks * sizeof(BlockDesc));
    L.o_blocks4 = bytes((int64_t)L.nblocks4 * sizeof(Block4));
    L.o_items4 = bytes((int64_t)3 * W4_MAX_ITEMS * sizeof(W4Item));      // table of the one launch, then the two tables of a grouped plan
    L.o_tensors = bytes(DVAE_TRAIN_MAX_TENSORS * sizeof(TensorDesc));
    L.o_chunks = bytes(p.n_params / 64 + 64);
    L.o_partials = bytes(p.rows_grid * 4 * sizeof(double));
    L.o_flags = bytes(1024 + 4 * (p.Bp / TB + 1));            // header of 256 words -- [0]: label-lo-plane epoch (RowsArgs::ylo_epoch), [1], [2], [16..]: the counters of the
                                                              // folded optimizer tail (fold_tail) -- then from byte 1024: ylo_dirty[tile]
    L.o_defer = bytes(DEFER_BYTES);
    L.o_wcopy = bytes(L.wcopy_elems * esz * np);              // PolX3: hi plane, then lo plane
    L.o_stash = bytes(L.stash_rows * p.Bp * esz * np);
    L.o_grads = bytes((int64_t)p.ksplit * p.n_params * sizeof(float));
    L.total = b;
    return 0;
}

static bool g_prof = false;
static thread_local bool g_eval_only = false;
// dvae_module_forward / dvae_module_backward: rows-kernel mode and its extra operands for the next dvae_train_grads call
struct ModeArgs {
    int mode = 0;
    float *out_r = nullptr, *out_mu = nullptr, *out_lv = nullptr, *out_z = nullptr;
    const float *g_r = nullptr, *g_mu = nullptr, *g_lv = nullptr, *g_z = nullptr;
    int ld_r = 0, ld_gr = 0;
};
static thread_local ModeArgs g_mode;
static thread_local long long g_rng_step_override = -1;   // dvae_train_step: its `step` argument numbers the noise draw   // dvae_train_eval: skip the wgrad launch
static unsigned long long* g_dbg = nullptr;   // set by dvae_train_debug_stamps
static double g_ms[4] = {0, 0, 0, 0};
static int64_t g_calls[4] = {0, 0, 0, 0};
struct PendingEv { hipEvent_t a, b; int which; };
static PendingEv g_pending[4096];
static int g_npending = 0;

struct ProfScope {
    hipStream_t s; int which; hipEvent_t a, b; bool on;
    ProfScope(hipStream_t s_, int w) : s(s_), which(w), on(g_prof && g_npending < 4096) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(b, s); g_pending[g_npending++] = PendingEv{a, b, which}; }
    }
};

}  // namespace fused
}  // namespace dvae

using namespace dvae;
using namespace dvae::fused;

// weight-gradient kernel form: 4 = workgroup k-split 4 x 4 blocks (default), 2 = 2 x 2 register ring (DVAE_WGRAD=ring),
// 1 = LDS-staged 4 x 4 blocks (DVAE_WGRAD=lds; bf16 policies only)
static int wgrad_form(const char* wk) {
    if (wk && strcmp(wk, "ring") == 0) return 2;
    if (wk && strcmp(wk, "lds") == 0) return 1;
    return 4;
}

static void w4_plan_classes(dvae_train_plan_t* plan, bool grouped);      // (defined behind fill_tables)
constexpr int W4_GROUPED = 0x40000000;      // plan->reserved0: two launches (decoder-side blocks, encoder-side blocks), see w4_build_items

extern "C" int dvae_train_plan(int model, int y_dim, int precision, int64_t B, int ksplit_hint, dvae_train_plan_t* plan) {
    DVAE_CHECK_ARG(plan != nullptr && B > 0, "train_plan: bad argument");
    if (!((model == DVAE_MODEL_M1 && y_dim == 0) || (model == DVAE_MODEL_M2 && (y_dim == 1 || y_dim == 513)) ||
          (model == DVAE_MODEL_M2_INFO && y_dim == 1) || (model == DVAE_MODEL_M2_DEC && y_dim == 1))) {
        set_error("train_plan: fused kernels cover M1 (y 0), M2 (y 1 or 513), M2_info and M2_DEC (y 1) at x 513 / h [128,128] / z 16; got model %d y_dim %d", model, y_dim);
        return DVAE_E_UNSUPPORTED;
    }
    DVAE_CHECK_ARG(precision == DVAE_PREC_F32 || precision == DVAE_PREC_BF16 || precision == DVAE_PREC_BF16X3, "train_plan: unknown precision %d", precision);
    memset(plan, 0, sizeof(*plan));
    plan->model = model; plan->y_dim = y_dim; plan->precision = precision; plan->B = B;
    plan->Bp = al(B, 128);
    const int ye = model == DVAE_MODEL_M2 ? y_dim : 0, yd = y_dim;
    const int rows[26] = {HD, HD, HD, HD, ZD, ZD, ZD, ZD, HD, HD, HD, HD, XD, XD,
                          HD, HD, HD, HD, 1, 1, HD, HD, HD, HD, 1, 1};
    const int cols[26] = {XD + ye, 1, HD, 1, HD, 1, HD, 1, ZD + yd, 1, HD, 1, HD, 1,
                          XD, 1, HD, 1, HD, 1, ZD, 1, HD, 1, HD, 1};
    int64_t off = 0;
    plan->n_tensors = model == DVAE_MODEL_M2_INFO ? 26 : 14;
    for (int i = 0; i < plan->n_tensors; ++i) {
        plan->tensor_offset[i] = off; plan->tensor_rows[i] = rows[i]; plan->tensor_cols[i] = cols[i];
        off += al((int64_t)rows[i] * cols[i], 64);
    }
    plan->n_params = off;
    const int64_t ntiles = (B + TB - 1) / TB;
    // rows kernel generation: the 8-wave chain + helper kernel wherever it exists (M1 / M2 with bf16 or bf16x3 operands: 48 vs 55 us
    // per 8192 frames under bf16x3, 31 vs 33 us under bf16); DVAE_ROWS=1 forces the 4-wave kernel
    const char* rk = getenv("DVAE_ROWS");
    int want = 2;
    if (rk && (atoi(rk) == 1 || atoi(rk) == 2 || atoi(rk) == 3)) want = atoi(rk);
    plan->rows_kernel = (want == 3 && rows3_supported(precision, model)) ? 3 : ((want >= 2 && rows2_supported(precision, model)) ? 2 : 1);
    if (!kDiagBuild && plan->rows_kernel == 1 && is_bf(precision)) {
        set_error("train_plan: DVAE_ROWS=1 (the 4-wave rows kernel) under the bf16 policies needs the diagnostic build (build.py --diag)");
        return DVAE_E_UNSUPPORTED;
    }
    if (model == DVAE_MODEL_M2_DEC && plan->rows_kernel != 2) {
        set_error("train_plan: M2_DEC exists in the 8-wave rows kernel only (bf16 / bf16x3 operands, DVAE_ROWS unset)");
        return DVAE_E_UNSUPPORTED;
    }
    // workgroups resident at once: the 8-wave kernel holds one per CU; the 4-wave bf16 kernel two.  Beyond that: persistent tile loop
    const int64_t maxg = plan->rows_kernel >= 2 ? 256 : 256 * (precision == DVAE_PREC_BF16 ? 2 : 1);
    plan->rows_grid = ntiles < maxg ? ntiles : maxg;
    int ks = ksplit_hint;
    // bf16: 8 slices up to 8192 frames (more slices = more slabs for the apply pass to sum); 12 beyond: 80 groups x 12 = 960
    // single-wave jobs fill the 1024 wave slots in one round (wgrad 164 -> 124 us at 65 536 frames, 2.49 -> 1.80 ms at 2^20).
    // fp32: the MFMA-bound wgrad needs a wave on every SIMD (>= 1024 wave jobs): 16.
    if (ks <= 0 && wgrad_form(getenv("DVAE_WGRAD")) == 4) {
        // workgroup k-split kernel: one workgroup per (4 x 4 tile block, frame slice) and per CU: as many slices as fill the 256 CUs
        // in one round (M2 y513: 22 blocks x 11), at least 128 frames (two k-steps per wave) each, at most 16 (the apply pass sums them)
        dvae_train_plan_t tmp = *plan;
        tmp.ksplit = 1;
        Layout L0;
        make_layout(tmp, L0);
        ks = 256 / L0.nblocks4;
        // as many slices as fill the CUs; the kernel's XCD-aware index map keeps 8 * (ks / 8) of them on one XCD each (the bf16 policies
        // are bound by the cold read of the stash) and deals the rest out block-wise
        if (ks > plan->Bp / 128) ks = (int)(plan->Bp / 128);
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
    }
    if (ks <= 0) {
        // bf16x3: a wave issues 16 MFMAs per k-step (3 per tile product), ~15.6 us of matrix time per 1024-frame slice on its own SIMD, and
        // 80 groups x 8 slices fill only 640 of the 1024 SIMDs: 12 slices (960 waves) take 3.3 us off the kernel and add 1.7 us of slab sums
        const int cap = precision == DVAE_PREC_BF16X3 ? 12 : (is_bf(precision) ? (plan->Bp > 12288 ? 12 : 8) : 16);
        ks = (int)(plan->Bp / (precision == DVAE_PREC_BF16X3 ? 640 : (is_bf(precision) ? 1024 : 512))); if (ks < 1) ks = 1; if (ks > cap) ks = cap;
    }
    if (ks > 64) ks = 64;
    plan->ksplit = ks;
    plan->reserved0 = 0;
    // Class-sliced schedule of the weight-gradient launch (round 5; w4_class_slices): every block is cut into as many frame slices as ITS
    // cost per k-step asks for, instead of one slice count for all.  Chosen when the library picks the slicing (no hint), for the workgroup
    // k-split kernel; the diagnostic variants whose protocols count `ksplit` arrivals per block (folded / deferred optimizer step) and
    // DVAE_W4_UNIFORM=1 keep the uniform table.  plan->reserved0 = workgroups of that launch (0: uniform slices), ksplit = slabs to sum.
    // Measured (tools/r05/w4_classes_ab.sh, w4_ab2.sh; M2 y 513, 8192 frames, same box, alternating): fp32 operands 60.3 -> 53.2 us (the launch is
    // bound by its matrix work: the model's quantity); bf16x3 25.3 -> 26.6 us and bf16 18.0 -> 19.8 us -- those launches are bound by the
    // 85 MB they pull from the stash (4.3 TB/s), not by the heaviest blocks' MFMAs, and more slices only add slab stores.  So: class-sliced
    // under the fp32 policy, uniform under the bf16 policies (DVAE_W4_CLASSES=1 forces it there, DVAE_W4_UNIFORM=1 switches it off).
    const bool want_classes = getenv("DVAE_W4_CLASSES") != nullptr ? atoi(getenv("DVAE_W4_CLASSES")) != 0 : !is_bf(precision);
    // DVAE_EXCHANGE_GROUPS=2 (opt-in, multi-GPU; read when the plan is made): the weight-gradient pass as TWO launches -- the blocks of the
    // decoder-side tensors (and M2_info's side nets), then those of the encoder -- each cut to fill the CUs by itself, so that the exchange
    // of the first group's gradient can run while the second launch computes (dvae_train_grads_group; Trainer, dp.py).  Always class-sliced.
    const bool want_groups = getenv("DVAE_EXCHANGE_GROUPS") != nullptr && atoi(getenv("DVAE_EXCHANGE_GROUPS")) == 2;
    if ((want_classes || want_groups) && ksplit_hint <= 0 && wgrad_form(getenv("DVAE_WGRAD")) == 4 && plan->Bp > 128 && getenv("DVAE_W4_UNIFORM") == nullptr &&
        getenv("DVAE_FOLD_APPLY") == nullptr && getenv("DVAE_DEFER_APPLY") == nullptr)
        w4_plan_classes(plan, want_groups);
    Layout L;
    make_layout(*plan, L);
    plan->workspace_bytes = L.total;
    plan->grad_offset_bytes = L.o_grads;
    const double mac = (double)HD * (XD + ye) + HD * HD + 2.0 * ZD * HD + (double)HD * (ZD + yd) + HD * HD + (double)XD * HD;
    const double dxm = (double)HD * HD + 2.0 * ZD * HD + (double)ZD * HD + HD * HD + (double)XD * HD;
    double macx = 0.0, dxx = 0.0;
    if (model == DVAE_MODEL_M2_INFO) {   // classifier fwd once; auxiliary fwd twice in the reference (z and z.detach())
        macx = ((double)HD * XD + HD * HD + HD) + 2.0 * ((double)HD * ZD + HD * HD + HD);
        dxx = ((double)HD * HD + HD) + ((double)HD * HD + HD + (double)ZD * HD);
    }
    plan->flops_per_step = 2.0 * (2.0 * mac + dxm) * (double)B + 2.0 * (2.0 * macx + dxx) * (double)B;
    plan->info_alpha = 0.0; plan->info_beta = 10.0; plan->info_gamma = 1.0;
    plan->min_hbm_bytes_per_step = 4.0 * (XD + y_dim + ZD) * (double)B;
    return 0;
}

static int used_slabs(const dvae_train_plan_t* plan);
static int64_t kper_of(const dvae_train_plan_t* p) {
    const int ks = is_bf(p->precision) ? 16 : 8;
    const int64_t unit = 4 * ks;
    return al((p->Bp + p->ksplit - 1) / p->ksplit, unit);
}

struct ABlock { int64_t row; int mvalid; int tensor; int m0; int bias_tensor; int tensor_hi; int bias_hi; };
struct BBlock { int64_t row; int nvalid; int col; int kind; int scol; int sncols; };   // kind 1 / 2: columns scol.. of the input matrix x / y (sncols wide)

template <typename T>
static void fill_tables(const dvae_train_plan_t* p, const Layout& L, char* ws_dev, GroupDesc* groups, BlockDesc* blocks, Block4* blocks4, TensorDesc* td) {
    const int64_t Bp = p->Bp;
    T* stash = (T*)(ws_dev + L.o_stash);
    auto S = [&](int64_t row) { return (const void*)(stash + row * Bp); };
    int n = 0;
    ABlock ab[64];
    BBlock bb[64];
    int na = 0, nb = 0;
    auto addA = [&](int64_t row0, int M, int tensor, int bias_tensor) {
        for (int m0 = 0; m0 < M; m0 += 32) ab[na++] = ABlock{row0 + m0, M - m0 < 32 ? M - m0 : 32, tensor, m0, bias_tensor, -1, -1};
    };
    auto addB = [&](int64_t row0, int N, int col0, int kind = 0) {
        for (int n0 = 0; n0 < N; n0 += 32) bb[nb++] = BBlock{row0 + n0, N - n0 < 32 ? N - n0 : 32, col0 + n0, kind, n0, N};
    };
    int nblk = 0, nblk4 = 0, layer_id = 0;
    // 2 x 2 group of tiles: A pair starting at block i, B pair starting at block j (all-null when out of range)
    auto make_group = [&](int i, int j) {
        GroupDesc d;
        memset(&d, 0, sizeof(d));
        d.bias_off[0] = d.bias_off[1] = -1;
        if (i >= na || j >= nb) return d;
        for (int ii = 0; ii < 2; ++ii) {
            if (i + ii >= na) continue;
            const ABlock& a = ab[i + ii];
            d.A[ii] = S(a.row);
            d.ldo[ii] = p->tensor_cols[a.tensor];
            d.mvalid[ii] = a.mvalid;
            if (a.bias_tensor >= 0 && j == 0) d.bias_off[ii] = p->tensor_offset[a.bias_tensor] + a.m0;
            for (int jj = 0; jj < 2; ++jj) {
                if (j + jj >= nb) continue;
                d.out_off[ii][jj] = p->tensor_offset[a.tensor] + (int64_t)a.m0 * d.ldo[ii] + bb[j + jj].col;
                if (ii == 0 && a.tensor_hi >= 0) d.out_off_hi[jj] = p->tensor_offset[a.tensor_hi] + bb[j + jj].col;
            }
            if (ii == 0 && a.tensor_hi >= 0) {
                d.split16 = 1;
                d.ldo_hi = p->tensor_cols[a.tensor_hi];
                d.bias_off_hi = p->tensor_offset[a.bias_hi];
            }
        }
        for (int jj = 0; jj < 2; ++jj) {
            if (j + jj >= nb) continue;
            d.Bm[jj] = S(bb[j + jj].row);
            d.nvalid[jj] = bb[j + jj].nvalid;
        }
        return d;
    };
    auto emit = [&]() {
        for (int i = 0; i < na; i += 2)
            for (int j = 0; j < nb; j += 2) groups[n++] = make_group(i, j);
        for (int i0 = 0; i0 < na; i0 += 4)
            for (int j0 = 0; j0 < nb; j0 += 4) {
                BlockDesc b;
                memset(&b, 0, sizeof(b));
                for (int k = 0; k < 4; ++k) {
                    b.At[k] = i0 + k < na ? S(ab[i0 + k].row) : nullptr;
                    b.Bt[k] = j0 + k < nb ? S(bb[j0 + k].row) : nullptr;
                }
                for (int wr = 0; wr < 2; ++wr)
                    for (int wc = 0; wc < 2; ++wc) b.g[wr * 2 + wc] = make_group(i0 + 2 * wr, j0 + 2 * wc);
                blocks[nblk++] = b;
            }
        // blocks of the workgroup k-split kernel: per run of B tiles of one kind (stash / x / labels)
        for (int r0 = 0; r0 < nb;) {
            int r1 = r0;
            while (r1 < nb && bb[r1].kind == bb[r0].kind) ++r1;
            for (int i0 = 0; i0 < na; i0 += 4)
                for (int j0 = r0; j0 < r1; j0 += 4) {
                    Block4 q;
                    memset(&q, 0, sizeof(q));
                    q.layer = layer_id;
                    q.raw = bb[r0].kind; q.rncols = bb[r0].sncols;
                    for (int k = 0; k < 4; ++k) {
                        q.bias_off[k] = -1;
                        if (i0 + k < na) {
                            const ABlock& a = ab[i0 + k];
                            q.At[k] = S(a.row);
                            q.ldo[k] = p->tensor_cols[a.tensor];
                            q.a_off[k] = p->tensor_offset[a.tensor] + (int64_t)a.m0 * q.ldo[k];
                            q.mvalid[k] = a.mvalid;
                            q.wt[k] = a.tensor; q.bt[k] = a.bias_tensor >= 0 ? a.bias_tensor : 0;
                            if (a.bias_tensor >= 0 && j0 == 0) q.bias_off[k] = p->tensor_offset[a.bias_tensor] + a.m0;
                            if (k == 0 && a.tensor_hi >= 0) {
                                q.wt_hi = a.tensor_hi; q.bt_hi = a.bias_hi;
                                q.split16 = 1;
                                q.ldo_hi = p->tensor_cols[a.tensor_hi];
                                q.a_off_hi = p->tensor_offset[a.tensor_hi];
                                q.bias_off_hi = p->tensor_offset[a.bias_hi];
                            }
                        }
                        if (j0 + k < r1) {
                            q.Bt[k] = S(bb[j0 + k].row);
                            q.bcol[k] = bb[j0 + k].col;
                            q.nvalid[k] = bb[j0 + k].nvalid;
                            q.rcol[k] = bb[j0 + k].scol;
                        }
                    }
                    blocks4[nblk4++] = q;
                }
            r0 = r1;
        }
        na = 0; nb = 0;
        ++layer_id;
    };
    const int ye = p->model == DVAE_MODEL_M2 ? p->y_dim : 0, yd = p->y_dim;
    addA(L.dh1T, HD, 0, 1); addB(L.xT, XD, 0, 1); if (ye) addB(L.yT, ye, XD, 2); emit();
    addA(L.dh2T, HD, 2, 3); addB(L.h1T, HD, 0); emit();
    ab[na++] = ABlock{L.dmlvT, 32, 4, 0, 5, 6, 7}; addB(L.h2T, HD, 0); emit();   // one tile: rows 0-15 mu head, 16-31 log_var head
    addA(L.dd1T, HD, 8, 9); addB(L.zT, ZD, 0); if (yd) addB(L.yT, yd, ZD, 2); emit();
    addA(L.dd2T, HD, 10, 11); addB(L.d1T, HD, 0); emit();
    addA(L.daT, XD, 12, 13); addB(L.d2T, HD, 0); emit();
    if (L.info) {   // classifier (tensors 14-19) and auxiliary net (20-25): scripts/training_M2_info_vad.py:141-143
        addA(L.dc1T, HD, 14, 15); addB(L.xT, XD, 0, 1); emit();
        addA(L.dc2T, HD, 16, 17); addB(L.c1T, HD, 0); emit();
        addA(L.dc3T, 1, 18, 19); addB(L.c2T, HD, 0); emit();
        addA(L.da1T, HD, 20, 21); addB(L.zT, ZD, 0); emit();
        addA(L.da2T, HD, 22, 23); addB(L.a1T, HD, 0); emit();
        addA(L.da3T, 1, 24, 25); addB(L.a2T, HD, 0); emit();
    }
    if (n != L.ntiles || nblk != L.nblocks || nblk4 != L.nblocks4) { fprintf(stderr, "dvae: internal group / block count mismatch %d vs %d, %d vs %d, %d vs %d\n", n, L.ntiles, nblk, L.nblocks, nblk4, L.nblocks4); }
    // tensors -> kernel-layout copies
    for (int i = 0; i < p->n_tensors; ++i) {
        TensorDesc t;
        memset(&t, 0, sizeof(t));
        t.off = p->tensor_offset[i]; t.rows = p->tensor_rows[i]; t.cols = p->tensor_cols[i];
        t.sf_off = -1; t.st_off = -1; t.sf_split = 1 << 30;
        td[i] = t;
    }
    td[0].sf_ld = L.ld1; td[2].sf_ld = HD; td[4].sf_ld = HD; td[6].sf_ld = HD; td[8].sf_ld = L.ld3; td[10].sf_ld = HD; td[12].sf_ld = HD;
    td[2].st_ld = HD; td[4].st_ld = 32; td[6].st_ld = 32; td[8].st_ld = HD; td[10].st_ld = HD; td[12].st_ld = NO;
    td[0].sf_off = L.W1s; td[0].sf_nt = 4; td[0].sf_split = XD; td[0].sf_gap = XP - XD;
    // the 8-wave rows kernel under the split-bf16 policy multiplies the x block of layer 1 (and of the M2_info classifier) in split fp16
    const int f16c = (p->precision == DVAE_PREC_BF16X3 && p->rows_kernel >= 2 && PolX3v2::XF16) ? XD : 0;
    td[0].sf_f16_cols = f16c;
    td[2].sf_off = L.W2s; td[2].sf_nt = 4; td[2].st_off = L.W2t; td[2].st_nt = 4; td[2].st_cmax = HD;
    td[4].sf_off = L.Wmvs; td[4].sf_nt = 1; td[4].st_off = L.Wmvt; td[4].st_nt = 4; td[4].st_cmax = HD;
    td[6].sf_off = L.Wmvs; td[6].sf_nt = 1; td[6].sf_roff = 16; td[6].st_off = L.Wmvt; td[6].st_nt = 4; td[6].st_roff = 16; td[6].st_cmax = HD;
    td[8].sf_off = L.W3s; td[8].sf_nt = 4; td[8].st_off = L.W3zt; td[8].st_nt = 1; td[8].st_cmax = ZD;
    td[10].sf_off = L.W4s; td[10].sf_nt = 4; td[10].st_off = L.W4t; td[10].st_nt = 4; td[10].st_cmax = HD;
    td[12].sf_off = L.W5s; td[12].sf_nt = NT_OUT; td[12].st_off = L.W5t; td[12].st_nt = 4; td[12].st_cmax = HD;
    if (p->rows_kernel == 3) {
        // rows3 kernel: 16-row tiles / 32-deep k-steps (apply_types.hpp: kind16); the heads' rows interleaved (rowmap), the backward-z
        // matrix's rows likewise (trowmap)
        for (int i = 0; i < 14; ++i) td[i].kind16 = 1;
        td[0].sf_nt = 8; td[0].sf_gap = 544 - XD;
        td[2].sf_nt = 8; td[2].st_nt = 8;
        td[4].sf_nt = 2; td[4].rowmap = 1; td[4].st_nt = 8;
        td[6].sf_nt = 2; td[6].rowmap = 2; td[6].sf_roff = 0; td[6].st_nt = 8;
        td[8].sf_nt = 8; td[8].sf_split = ZD; td[8].sf_gap = 32 - ZD; td[8].st_nt = 2; td[8].trowmap = 1;
        td[10].sf_nt = 8; td[10].st_nt = 8;
        td[12].sf_nt = NO / 16; td[12].st_nt = 8;
    }
    if (L.info) {
        td[14].sf_off = L.Wc1s; td[14].sf_nt = 4; td[14].sf_ld = XP; td[14].sf_split = XD; td[14].sf_gap = XP - XD; td[14].sf_f16_cols = f16c;
        td[16].sf_off = L.Wc2s; td[16].sf_nt = 4; td[16].sf_ld = HD; td[16].st_off = L.Wc2t; td[16].st_nt = 4; td[16].st_ld = HD; td[16].st_cmax = HD;
        td[20].sf_off = L.Wa1s; td[20].sf_nt = 4; td[20].sf_ld = ZD; td[20].st_off = L.Wa1t; td[20].st_nt = 1; td[20].st_ld = HD; td[20].st_cmax = ZD;
        td[22].sf_off = L.Wa2s; td[22].sf_nt = 4; td[22].sf_ld = HD; td[22].st_off = L.Wa2t; td[22].st_nt = 4; td[22].st_ld = HD; td[22].st_cmax = HD;
    }
}

// folded launches per workspace (the block counters of fold_tail count arrivals across launches); dvae_train_init zeroes the
// workspace and forgets its count
static std::mutex g_fold_mu;
static std::unordered_map<const void*, unsigned> g_fold_seq;
// workgroups of the weight-gradient launch of a workspace (the size of its item table, fixed by dvae_train_init)
struct W4Grids { int g[3]; };      // table 0 (the one launch of an ungrouped plan), tables 1 / 2 (group 0 / 1 of a grouped plan)
static std::unordered_map<const void*, W4Grids> g_w4_grid;
static void w4_grid_put(const void* ws, const W4Grids& grids) { std::lock_guard<std::mutex> lk(g_fold_mu); g_w4_grid[ws] = grids; }
static int w4_grid_get(const void* ws, int table) { std::lock_guard<std::mutex> lk(g_fold_mu); auto it = g_w4_grid.find(ws); return it == g_w4_grid.end() ? -1 : it->second.g[table]; }
// dvae_train_grads_group: which part of the step the next dvae_train_grads call runs (-1: all of it)
static thread_local int g_w4_group = -1;
static unsigned fold_seq_next(const void* ws) { std::lock_guard<std::mutex> lk(g_fold_mu); return ++g_fold_seq[ws]; }
static void fold_seq_reset(const void* ws) { std::lock_guard<std::mutex> lk(g_fold_mu); g_fold_seq.erase(ws); }

// ---- deferred optimizer step: host-side state per workspace
struct PendingUpdate { float* params; float* m; float* v; int step; double lr, beta1, beta2, adam_eps; int n_slabs; };
struct DeferState { bool pending = false; PendingUpdate u{}; unsigned seq_arrive = 0, seq_done = 0; int ntasks = 0; };
static std::mutex g_defer_mu;
static std::unordered_map<const void*, DeferState> g_defer_state;
static DeferState defer_state_get(const void* ws) { std::lock_guard<std::mutex> lk(g_defer_mu); return g_defer_state[ws]; }
static void defer_state_put(const void* ws, const DeferState& st) { std::lock_guard<std::mutex> lk(g_defer_mu); g_defer_state[ws] = st; }
static void defer_state_reset(const void* ws) { std::lock_guard<std::mutex> lk(g_defer_mu); g_defer_state.erase(ws); }
// dvae_train_step_deferred -> dvae_train_grads: run the rows kernel in its deferred form
struct DeferRequest { bool on = false, have = false; PendingUpdate u{}; unsigned seq_arrive = 0, seq_done = 0; float* losses3 = nullptr; };
static thread_local DeferRequest g_defer_req;

// tasks of the deferred update (apply_common.hpp: DeferTask): 32 x 32 tiles of every tensor with kernel-layout copies, per column block
// (a tile never straddles the split of the forward copy), and 1024-element chunks of the tensors without copies
static int build_defer_tasks(const dvae_train_plan_t* p, const TensorDesc* td, DeferTask* out, int cap) {
    int n = 0;
    auto put = [&](int t, int r0, int cbeg, int cend, int kind) { if (n < cap) out[n] = DeferTask{t, r0, cbeg, cend, kind, 0, 0, 0}; ++n; };
    for (int t = 0; t < p->n_tensors; ++t) {
        const int rows = p->tensor_rows[t], cols = p->tensor_cols[t];
        if (td[t].sf_off < 0 && td[t].st_off < 0) {
            for (int e0 = 0; e0 < rows * cols; e0 += 1024) put(t, 0, e0, rows * cols, 1);
            continue;
        }
        const int split = td[t].sf_off >= 0 && td[t].sf_split < cols ? td[t].sf_split : cols;
        for (int r0 = 0; r0 < rows; r0 += 32) {
            for (int cb = 0; cb < split; cb += 32) put(t, r0, cb, split, 0);
            for (int cb = split; cb < cols; cb += 32) put(t, r0, cb, cols, 0);
        }
    }
    return n;
}

static ApplyArgs make_apply_args(const dvae_train_plan_t* plan, const Layout& L, float* params, float* m, float* v, char* ws, int n_slabs,
                                 bool adam, int step, double lr, double beta1, double beta2, double adam_eps, double grad_scale, float* losses3) {
    ApplyArgs a;
    memset(&a, 0, sizeof(a));
    a.p = params; a.m = m; a.v = v;
    a.slabs = (const float*)(ws + L.o_grads); a.slab_stride = plan->n_params; a.nslabs = n_slabs;
    a.tensors = (const TensorDesc*)(ws + L.o_tensors); a.ntensors = plan->n_tensors;
    a.chunk_tensor = (const unsigned char*)(ws + L.o_chunks); a.n_params = plan->n_params;
    a.wcopy = ws + L.o_wcopy; a.wpl = L.wcopy_elems;
    if (adam) {
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        a.one_minus_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_minus_b2 = (float)(1.0 - beta2);
        a.step_size = (float)(lr / bc1); a.bc2_sqrt = (float)sqrt(bc2); a.eps = (float)adam_eps; a.gscale = (float)grad_scale;
    }
    a.partials = (const double*)(ws + L.o_partials); a.npartials = (int)plan->rows_grid; a.B = plan->B; a.losses3 = losses3; a.accum = (double*)(uintptr_t)plan->loss_accum;
    a.info = plan->model == DVAE_MODEL_M2_INFO; a.alpha = (float)plan->info_alpha; a.beta = (float)plan->info_beta; a.gamma = (float)plan->info_gamma;
    return a;
}

static int launch_apply(const dvae_train_plan_t* plan, const Layout& L, float* params, float* m, float* v, char* ws, int n_slabs,
                        bool adam, int step, double lr, double beta1, double beta2, double adam_eps, double grad_scale,
                        float* losses3, hipStream_t s) {
    const ApplyArgs a = make_apply_args(plan, L, params, m, v, ws, n_slabs, adam, step, lr, beta1, beta2, adam_eps, grad_scale, losses3);
    // DVAE_APPLY=units (opt-in; bf16 copies, at most 12 slabs): the unit form of the deferred step as its own launch.  Measured (M2 y513,
    // 8192 frames, bf16x3, same box, alternating): 10.8 against 9.4 us gross for the one-thread-per-parameter kernel below -- a quarter of the
    // instructions and no lone 2-byte stores, but 324 workgroups instead of 1183 on a launch that is one load round trip, one store round
    // trip and its own start-up either way: the flat kernel stays the default.
    const int ntasks = defer_state_get(ws).ntasks;
    const char* ak = getenv("DVAE_APPLY");
#ifdef DVAE_DIAG
    if (adam && is_bf(plan->precision) && n_slabs <= 12 && ntasks > 0 && ntasks <= DEFER_MAX_TASKS && ak && strcmp(ak, "units") == 0) {
        const int nunits = 4 * ntasks;
        const dim3 gu((unsigned)((nunits + 3) / 4 + 1));
        const DeferTask* tasks = (const DeferTask*)(ws + L.o_defer);
        if (plan->precision == DVAE_PREC_BF16X3) hipLaunchKernelGGL((apply_units_kernel<__bf16, 2>), gu, dim3(256), 0, s, a, tasks, nunits);
        else hipLaunchKernelGGL((apply_units_kernel<__bf16, 1>), gu, dim3(256), 0, s, a, tasks, nunits);
        DVAE_LAUNCH_OK("apply_units_kernel");
        return 0;
    }
#else
    (void)ntasks; (void)ak;
#endif
    const dim3 grid((unsigned)((plan->n_params + 255) / 256 + 1));   // + 1: loss finalisation block
    if (plan->precision == DVAE_PREC_BF16X3) {
        if (adam) hipLaunchKernelGGL((apply_kernel<__bf16, true, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((apply_kernel<__bf16, false, 2>), grid, dim3(256), 0, s, a);
    } else if (plan->precision == DVAE_PREC_BF16) {
        if (adam) hipLaunchKernelGGL((apply_kernel<__bf16, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((apply_kernel<__bf16, false>), grid, dim3(256), 0, s, a);
    } else {
        if (adam) hipLaunchKernelGGL((apply_kernel<float, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((apply_kernel<float, false>), grid, dim3(256), 0, s, a);
    }
    DVAE_LAUNCH_OK("apply_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Host-side schedule of wgrad4_kernel: the item table (one workgroup = one item) the kernel reads.
//
// Uniform (plan->reserved0 == 0): `ksplit` equal frame slices for every block, workgroup -> (slice, block) by the XCD-aware index map the
// kernel had until round 4 (slice s on XCD s % 8 for 8 * (ksplit / 8) "pure" slices, the rest dealt out block-wise).
//
// Class-sliced (round 5): the blocks of one launch differ by 4x in work per k-step -- 4 x 4 tiles at three MFMAs per product (48 per wave
// and k-step), label-fed 4 x 4 blocks at two (binary labels: one operand plane), 4 x 1 / 1 x 4 edge blocks (12, bound by their operand
// loads) -- and with one slice count for all, the launch lasted as long as its heaviest blocks while a quarter of the CUs had finished
// (M2 y 513, 8192 frames, 10 slices: 13 k-steps x 1536 clocks on the heavy blocks, 42 % of that on the light ones).  Here every block b gets
// s_b slices by water-filling on cost_b x k-steps: the block that would finish last is split further until the CUs are used up.  A block
// writes slabs 0 .. s_b - 1; the slabs it never writes hold the zeros dvae_train_init put there, so the optimizer launch (and the slab
// reduction of the multi-GPU path) keep summing plan->ksplit = max s_b slabs for every parameter, in slab order: still one deterministic sum.
struct W4Sched { std::vector<W4Item> items; int grid = 0; };

static void w4_host_blocks(const dvae_train_plan_t& p, const Layout& L, std::vector<Block4>& out) {
    std::vector<GroupDesc> tiles((size_t)L.ntiles + 8);
    std::vector<BlockDesc> blocks((size_t)L.nblocks + 8);
    out.assign((size_t)L.nblocks4 + 8, Block4{});
    TensorDesc td[DVAE_TRAIN_MAX_TENSORS];
    memset(td, 0, sizeof(td));
    char* const base = reinterpret_cast<char*>((uintptr_t)1 << 20);      // never dereferenced: the descriptors' pointers only say "present"
    if (is_bf(p.precision)) fill_tables<__bf16>(&p, L, base, tiles.data(), blocks.data(), out.data(), td);
    else fill_tables<float>(&p, L, base, tiles.data(), blocks.data(), out.data(), td);
    out.resize((size_t)L.nblocks4);
}

// cost of one 16-frame k-step of block b for one wave, in clocks (a model: matrix-pipe time against operand-fragment pulls at one 1 KB
// fragment per ~70 clocks and wave; the in-lane bias sums where the block carries bias rows)
static double w4_block_cost(const dvae_train_plan_t& p, const Block4& b) {
    int na = 0, nb = 0;
    bool bias = b.split16 && b.bias_off_hi >= 0;
    for (int k = 0; k < 4; ++k) { if (b.At[k]) na = k + 1; if (b.Bt[k]) nb = k + 1; bias = bias || b.bias_off[k] >= 0; }
    const bool x3 = p.precision == DVAE_PREC_BF16X3, bf = is_bf(p.precision);
    const bool one_plane_b = x3 && b.raw == 2 && p.rows_kernel >= 2;      // label tiles: binary labels need no lo plane (two MFMAs per product)
    double mfma, load, fsum;
    // clocks per 1 KB operand fragment and wave: 70 for a lone wave (tools/r03/kstep_bench.hip); W4_KB_CLOCKS overrides (experiments)
    static const double kb_clocks = getenv("DVAE_W4_KB_CLOCKS") ? atof(getenv("DVAE_W4_KB_CLOCKS")) : 70.0;
    if (bf) {
        mfma = (double)na * nb * (x3 ? (one_plane_b ? 2 : 3) : 1) * 32.0;
        load = ((double)na * (x3 ? 2 : 1) + (double)nb * (x3 ? (one_plane_b ? 1 : 2) : 1)) * kb_clocks;
        fsum = bias ? (double)na * (x3 ? 2 : 1) * 16 * 4 * 0.5 : 0.0;
    } else {
        mfma = (double)na * nb * 2 * 4 * 64.0;                            // two 8-frame k-steps of four 32 x 32 x 2 MFMAs per product
        load = (double)(na + nb) * 2 * 70.0;
        fsum = bias ? (double)na * 2 * 8 * 4 * 0.5 : 0.0;
    }
    if (b.raw != 0 && p.B >= 131072 && bf) load *= 2.0;                   // raw fp32 input tiles (large batches): twice the bytes, converted in the loop
    return std::max(mfma, load) + 0.3 * std::min(mfma, load) + fsum;
}

static int64_t w4_kper(const dvae_train_plan_t& p, int slices) {
    const int64_t unit = 4 * (is_bf(p.precision) ? 16 : 8);
    return al((p.Bp + slices - 1) / slices, unit);
}

// s[b] = slices of block b (>= 1), at most `max_items` in all, at most 16 per block (the optimizer launch sums that many slabs), at
// least 128 frames per slice
static inline bool w4_classed(const dvae_train_plan_t& p) { return (p.reserved0 & ~W4_GROUPED) > 0; }
static inline bool w4_grouped(const dvae_train_plan_t& p) { return (p.reserved0 & W4_GROUPED) != 0; }
// group of a block in the two-launch schedule: 0 = launched first (decoder layers 3-5 and, M2_info, the side nets 6-11: tensors 8 and up, the
// upper part of the flat gradient), 1 = the encoder (layers 0-2: tensors 0-7, the lower part)
static inline int w4_group_of(const Block4& b) { return b.layer <= 2 ? 1 : 0; }

// group < 0: every block; otherwise only the blocks of that group take part (the others keep s = 0)
static void w4_class_slices(const dvae_train_plan_t& p, const std::vector<Block4>& blocks, int max_items, std::vector<int>& s, int group = -1) {
    const int n = (int)blocks.size();
    s.assign(n, 1);
    std::vector<double> cost(n);
    for (int b = 0; b < n; ++b) cost[b] = (group < 0 || w4_group_of(blocks[b]) == group) ? w4_block_cost(p, blocks[b]) : 0.0;
    for (int b = 0; b < n; ++b) if (cost[b] == 0.0) s[b] = 0;
    int cap = (int)std::min<int64_t>(16, p.Bp / 128);
    if (cap < 1) cap = 1;
    int total = 0;
    for (int b = 0; b < n; ++b) total += s[b];
    for (;;) {
        int worst = -1;
        double tw = 0.0;
        for (int b = 0; b < n; ++b) { if (s[b] == 0) continue; const double t = cost[b] * (double)w4_kper(p, s[b]); if (t > tw) { tw = t; worst = b; } }
        if (worst < 0) break;
        int s2 = s[worst] + 1;
        while (s2 <= cap && w4_kper(p, s2) >= w4_kper(p, s[worst])) ++s2;       // the next slice count that shortens the slices
        if (s2 > cap || total + (s2 - s[worst]) > max_items) break;            // the block that finishes last cannot be split further
        total += s2 - s[worst];
        s[worst] = s2;
    }
}

static void w4_debug_print(const dvae_train_plan_t& p, const std::vector<Block4>& blocks, const std::vector<int>& s) {
    if (getenv("DVAE_W4_DEBUG") == nullptr) return;                        // DVAE_W4_DEBUG=1: the schedule, block by block (stderr)
    {
        for (int b = 0; b < (int)blocks.size(); ++b) {
            if (s[b] == 0) continue;
            int na = 0, nb = 0;
            for (int k = 0; k < 4; ++k) { if (blocks[b].At[k]) na = k + 1; if (blocks[b].Bt[k]) nb = k + 1; }
            const int64_t kper = w4_kper(p, s[b]);
            fprintf(stderr, "dvae w4: block %2d layer %d %d x %d kind %d bias %d cost %6.0f slices %2d of %lld frames -> %.0f clocks per wave\n", b, blocks[b].layer, na, nb,
                    blocks[b].raw, (int)(blocks[b].bias_off[0] >= 0), w4_block_cost(p, blocks[b]), (int)((p.Bp + kper - 1) / kper), (long long)kper,
                    w4_block_cost(p, blocks[b]) * (double)kper / 64.0);
        }
    }
}

static void w4_plan_classes(dvae_train_plan_t* plan, bool grouped) {
    dvae_train_plan_t tmp = *plan;
    tmp.ksplit = 1;
    Layout L0;
    make_layout(tmp, L0);
    std::vector<Block4> blocks;
    w4_host_blocks(tmp, L0, blocks);
    int nslabs = 1, items_max = 0;
    for (int grp = grouped ? 0 : -1; grp <= (grouped ? 1 : -1); ++grp) {
        std::vector<int> s;
        w4_class_slices(tmp, blocks, 256, s, grp);
        w4_debug_print(tmp, blocks, s);
        int items = 0;
        for (size_t b = 0; b < blocks.size(); ++b) {
            if (s[b] == 0) continue;
            const int64_t kper = w4_kper(tmp, s[b]);
            const int eff = (int)((tmp.Bp + kper - 1) / kper);
            nslabs = std::max(nslabs, eff);
            items += eff;
        }
        items_max = std::max(items_max, items);
    }
    plan->ksplit = nslabs;
    plan->reserved0 = items_max | (grouped ? W4_GROUPED : 0);
}

// group: -1 = the one launch of an ungrouped plan; 0 / 1 = the launches of a grouped plan (w4_group_of)
static void w4_build_items(const dvae_train_plan_t& p, const Layout& L, W4Sched& out, int group = -1) {
    out.items.clear();
    out.grid = 0;
    if (!w4_classed(p)) {
        // uniform slices, the round-2 index map
        const int64_t kper = w4_kper(p, p.ksplit);
        const int ks = (int)((p.Bp + kper - 1) / kper), nblocks = L.nblocks4;
        const int a8 = ks >> 3, r8 = ks & 7, npure = a8 * nblocks, per = (r8 * nblocks + 7) >> 3;
        out.grid = 8 * (a8 * nblocks + per);
        for (int w = 0; w < out.grid; ++w) {
            const int xcd = w & 7, j = w >> 3;
            W4Item it{-1, 0, 0, 0};
            if (j < npure) { it.slice = xcd + 8 * (j / nblocks); it.block = j % nblocks; }
            else {
                const int e = xcd * per + (j - npure);
                if (j - npure < per && e < r8 * nblocks) { it.slice = 8 * a8 + e / nblocks; it.block = e % nblocks; }
            }
            if (it.block >= 0) { it.kbeg = (int64_t)it.slice * kper; it.kend = std::min<int64_t>(it.kbeg + kper, p.Bp); }
            out.items.push_back(it);
        }
        return;
    }
    std::vector<Block4> blocks;
    w4_host_blocks(p, L, blocks);
    dvae_train_plan_t tmp = p;
    std::vector<int> s;
    w4_class_slices(tmp, blocks, 256, s, group);
    struct It { W4Item it; double cost; };
    // items that read the same stash lines -- the blocks of one layer cut the same way, over the same frames -- form a group: one XCD
    std::map<std::tuple<int, int64_t, int>, std::vector<It>> groups;
    for (int b = 0; b < (int)blocks.size(); ++b) {
        if (s[b] == 0) continue;                                          // not in this group's launch
        const int64_t kper = w4_kper(p, s[b]);
        const int eff = (int)((p.Bp + kper - 1) / kper);
        const double c = w4_block_cost(p, blocks[b]);
        for (int i = 0; i < eff; ++i) {
            W4Item it{b, i, (int64_t)i * kper, std::min<int64_t>((int64_t)(i + 1) * kper, p.Bp)};
            groups[std::make_tuple(blocks[b].layer, kper, i)].push_back(It{it, c * (double)(it.kend - it.kbeg)});
        }
    }
    std::vector<std::vector<It>> gl;
    for (auto& kv : groups) gl.push_back(kv.second);
    auto gcost = [](const std::vector<It>& g) { double t = 0; for (const It& i : g) t += i.cost; return t; };
    std::stable_sort(gl.begin(), gl.end(), [&](const std::vector<It>& a, const std::vector<It>& b) { return gcost(a) > gcost(b); });
    std::vector<It> lists[8];
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int cap = 32;                                                   // CUs of an XCD: one round of workgroups
    auto lightest = [&](int need) { int best = -1; for (int x = 0; x < 8; ++x) if ((int)lists[x].size() + need <= cap && (best < 0 || load[x] < load[best])) best = x; return best; };
    for (auto& g : gl) {
        int x = lightest((int)g.size());
        if (x >= 0) { for (const It& i : g) { lists[x].push_back(i); load[x] += i.cost; } continue; }
        for (const It& i : g) {                                           // no XCD has room for the whole group: item by item
            x = lightest(1);
            if (x < 0) { x = 0; for (int q = 1; q < 8; ++q) if (lists[q].size() < lists[x].size()) x = q; }      // (more than 256 items: cannot happen by construction)
            lists[x].push_back(i); load[x] += i.cost;
        }
    }
    size_t maxlen = 0;
    for (int x = 0; x < 8; ++x) {
        std::stable_sort(lists[x].begin(), lists[x].end(), [](const It& a, const It& b) { return a.cost > b.cost; });      // longest first
        maxlen = std::max(maxlen, lists[x].size());
    }
    out.grid = (int)(8 * maxlen);
    out.items.assign((size_t)out.grid, W4Item{-1, 0, 0, 0});
    for (int x = 0; x < 8; ++x)
        for (size_t q = 0; q < lists[x].size(); ++q) out.items[q * 8 + x] = lists[x][q].it;
}

extern "C" int dvae_train_repack(const dvae_train_plan_t* plan, const float* params, void* ws, void* stream) {
    DVAE_CHECK_ARG(plan && params && ws, "train_repack: bad argument");
    DVAE_CHECK_ARG(!defer_state_get(ws).pending, "train_repack: an optimizer update is pending on this workspace (dvae_train_step_deferred): call dvae_train_flush BEFORE writing parameters");
    Layout L;
    make_layout(*plan, L);
    return launch_apply(plan, L, const_cast<float*>(params), nullptr, nullptr, (char*)ws, 0, false, 1, 0, 0, 0, 0, 0, nullptr, (hipStream_t)stream);
}

extern "C" int dvae_train_init(const dvae_train_plan_t* plan, const float* params, void* ws, void* stream) {
    DVAE_CHECK_ARG(plan && params && ws, "train_init: bad argument");
    Layout L;
    make_layout(*plan, L);
    DVAE_CHECK_ARG(L.total == plan->workspace_bytes, "train_init: plan does not match this library (workspace %lld vs %lld)",
                   (long long)plan->workspace_bytes, (long long)L.total);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    DVAE_HIP(hipMemsetAsync(w, 0, (size_t)L.total, s));
    fold_seq_reset(ws);
    defer_state_reset(ws);
    GroupDesc* tiles = new GroupDesc[L.ntiles + 8];
    BlockDesc* blocks = new BlockDesc[L.nblocks + 8];
    Block4* blocks4 = new Block4[L.nblocks4 + 8];
    TensorDesc td[DVAE_TRAIN_MAX_TENSORS];
    memset(td, 0, sizeof(td));
    if (is_bf(plan->precision)) fill_tables<__bf16>(plan, L, w, tiles, blocks, blocks4, td);
    else fill_tables<float>(plan, L, w, tiles, blocks, blocks4, td);
    hipError_t e1 = hipMemcpyAsync(w + L.o_tiles, tiles, (size_t)L.ntiles * sizeof(GroupDesc), hipMemcpyHostToDevice, s);
    hipError_t e2 = hipMemcpyAsync(w + L.o_tensors, td, sizeof(td), hipMemcpyHostToDevice, s);
    hipError_t e5 = hipMemcpyAsync(w + L.o_blocks, blocks, (size_t)L.nblocks * sizeof(BlockDesc), hipMemcpyHostToDevice, s);
    hipError_t e6 = hipMemcpyAsync(w + L.o_blocks4, blocks4, (size_t)L.nblocks4 * sizeof(Block4), hipMemcpyHostToDevice, s);
    W4Sched sched[3];                                                     // [0]: the one launch; [1], [2]: the two launches of a grouped plan
    W4Grids grids{{0, 0, 0}};
    hipError_t e8 = hipSuccess;
    for (int t = 0; t < 3; ++t) {
        const bool grouped = w4_grouped(*plan);
        if ((t == 0) == grouped) continue;                                // a plan uses either table 0 or tables 1 and 2
        w4_build_items(*plan, L, sched[t], t - 1);
        DVAE_CHECK_ARG(sched[t].grid > 0 && sched[t].grid <= W4_MAX_ITEMS, "train_init: weight-gradient schedule of %d workgroups does not fit the item table", sched[t].grid);
        grids.g[t] = sched[t].grid;
        const hipError_t e = hipMemcpyAsync(w + L.o_items4 + (int64_t)t * W4_MAX_ITEMS * sizeof(W4Item), sched[t].items.data(), sched[t].items.size() * sizeof(W4Item), hipMemcpyHostToDevice, s);
        if (e8 == hipSuccess) e8 = e;
    }
    const int64_t nchunks = plan->n_params / 64;
    unsigned char* ct = new unsigned char[nchunks + 64];
    memset(ct, 255, (size_t)nchunks + 64);
    for (int t = 0; t < plan->n_tensors; ++t) {
        const int64_t c0 = plan->tensor_offset[t] / 64, ne = (int64_t)plan->tensor_rows[t] * plan->tensor_cols[t];
        for (int64_t c = c0; c < c0 + (ne + 63) / 64; ++c) ct[c] = (unsigned char)t;
    }
    hipError_t e4 = hipMemcpyAsync(w + L.o_chunks, ct, (size_t)nchunks, hipMemcpyHostToDevice, s);
    DeferTask* dt = new DeferTask[DEFER_MAX_TASKS];
    memset(dt, 0, sizeof(DeferTask) * DEFER_MAX_TASKS);
    { DeferState st; st.ntasks = build_defer_tasks(plan, td, dt, DEFER_MAX_TASKS); defer_state_put(ws, st); }
    hipError_t e7 = hipMemcpyAsync(w + L.o_defer, dt, sizeof(DeferTask) * DEFER_MAX_TASKS, hipMemcpyHostToDevice, s);
    hipError_t e3 = hipStreamSynchronize(s);
    delete[] dt;
    delete[] tiles;
    delete[] blocks;
    delete[] blocks4;
    delete[] ct;
    DVAE_HIP(e1); DVAE_HIP(e2); DVAE_HIP(e4); DVAE_HIP(e5); DVAE_HIP(e6); DVAE_HIP(e7); DVAE_HIP(e8); DVAE_HIP(e3);
    w4_grid_put(ws, grids);
    return dvae_train_repack(plan, params, ws, stream);
}

template <typename P, int YP, bool YENC, bool INFO = false>
static int launch_rows(const RowsArgs& a, int grid, hipStream_t s) {
    const size_t lds = Pl<P>::bytes;
    static bool attr_done[64] = {};                      // per device: the attribute belongs to the device's copy of the code object
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)vae_rows_kernel<P, YP, YENC, INFO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(rows kernel, %zu B LDS): %s", lds, hipGetErrorString(e)); return (int)e; }
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((vae_rows_kernel<P, YP, YENC, INFO>), dim3(grid), dim3(256), lds, s, a);
    DVAE_LAUNCH_OK("vae_rows_kernel");
    return 0;
}

// dvae_train_step -> dvae_train_grads: "run the optimizer step in the weight-gradient launch if you can" (done: it did)
struct FoldRequest {
    bool want = false, done = false;
    float* params = nullptr; float* m = nullptr; float* v = nullptr; float* losses3 = nullptr;
    int step = 0; double lr = 0, beta1 = 0, beta2 = 0, adam_eps = 0;
};
static thread_local FoldRequest g_fold;

static int device_cu_count(int dev) {
    static int cus[64] = {};
    if (dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 1;
        cus[dev] = n;
    }
    return cus[dev];
}

// The pending update of a workspace (dvae_train_step_deferred), applied now with apply_kernel: the same slab sums, the same element
// arithmetic as the in-kernel form.  Every entry point that reads parameters, moments, weight copies or gradient slabs calls this first.
extern "C" int dvae_train_flush(const dvae_train_plan_t* plan, void* ws, void* stream) {
    DVAE_CHECK_ARG(plan && ws, "train_flush: bad argument");
    DeferState st = defer_state_get(ws);
    if (!st.pending) return 0;
    Layout L;
    make_layout(*plan, L);
    st.pending = false;
    defer_state_put(ws, st);
    ProfScope ps((hipStream_t)stream, 3);
    return launch_apply(plan, L, st.u.params, st.u.m, st.u.v, (char*)ws, st.u.n_slabs, true, st.u.step, st.u.lr, st.u.beta1, st.u.beta2, st.u.adam_eps,
                        1.0, nullptr /* the step's loss scalars were finalised by its own rows kernel */, (hipStream_t)stream);
}

extern "C" int dvae_train_pending(const void* ws) { return defer_state_get(ws).pending ? 1 : 0; }

// can this plan run the deferred form?  (8-wave rows kernel, M1 / M2 train step; the whole grid resident at once: one workgroup per CU;
// at most two update tasks per chain wave: otherwise the update would take longer in the opening than in its own launch)
static bool defer_possible(const dvae_train_plan_t* plan, const void* ws) {
    if (!(plan->rows_kernel == 2 && rows2_supported(plan->precision, plan->model))) return false;
    if (!(plan->model == DVAE_MODEL_M1 || plan->model == DVAE_MODEL_M2)) return false;
    if (plan->row_index != 0 && plan->row_count <= 0) return false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int nt = defer_state_get(ws).ntasks;
    if (nt <= 0 || nt > DEFER_MAX_TASKS) return false;
    if (plan->rows_grid > device_cu_count(dev) || 8 * plan->rows_grid < 4 * nt) return false;      // at most two 8-row units per chain wave
    if (wgrad_form(getenv("DVAE_WGRAD")) != 4 || plan->Bp <= 128) return false;      // the loss scalars come from an extra workgroup of wgrad4_kernel
    const int64_t kper = kper_of(plan);
    return (plan->Bp + kper - 1) / kper <= 12;
}

extern "C" int dvae_train_can_defer(const dvae_train_plan_t* plan, const void* ws) {
    if (!plan || !ws) return 0;
    // OPT-IN (DVAE_DEFER_APPLY=1): bit-identical and, on the MI355X, not faster -- see DESIGN.md (round 4, item 3) for the ablation
    const char* de = getenv("DVAE_DEFER_APPLY");
    if (!kDiagBuild || !(de && atoi(de) == 1) || getenv("DVAE_FOLD_APPLY") != nullptr) return 0;      // (the deferred rows kernel exists in -DDVAE_DIAG builds only)
    return defer_possible(plan, ws) ? 1 : 0;
}

extern "C" int dvae_train_grads(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                                const float* y, int ldy, const float* eps_noise, float elbo_eps, int reduce_slabs, void* stream) {
    DVAE_CHECK_ARG(plan && params && ws && x && ldx >= XD, "train_grads: bad argument");
    DVAE_CHECK_ARG(plan->y_dim == 0 || (y != nullptr && ldy >= plan->y_dim), "train_grads: y missing or ldy < y_dim");
    if (!g_defer_req.on) { const int frc = dvae_train_flush(plan, ws, stream); if (frc) return frc; }
    Layout L;
    make_layout(*plan, L);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    const bool bf = plan->precision == DVAE_PREC_BF16, x3 = plan->precision == DVAE_PREC_BF16X3;
    const int esz = (bf || x3) ? 2 : 4;
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = (const int64_t*)(uintptr_t)plan->row_index;
    a.n_rows = plan->row_count; a.bad_rows = (int*)(uintptr_t)plan->bad_row_counter;
    DVAE_CHECK_ARG(a.rows == nullptr || a.n_rows > 0, "train_grads: row_index set but row_count <= 0");
    a.rng_seed = plan->rng_seed; a.rng_step = g_rng_step_override >= 0 ? (unsigned long long)g_rng_step_override : plan->rng_step;
    a.x = x; a.y = y; a.eps = eps_noise; a.ldx = ldx; a.ldy = plan->y_dim ? ldy : 0; a.ydim = plan->y_dim;
    a.fastx = (ldx == XD) && (((uintptr_t)x & 15) == 0);
    a.fasty = (plan->y_dim == XD) && (ldy == XD) && (((uintptr_t)y & 15) == 0);
    a.B = plan->B; a.Bp = plan->Bp; a.ntiles = (int)((plan->B + TB - 1) / TB);
    a.invB = (float)(1.0 / (double)plan->B); a.elbo_eps = elbo_eps;
    char* wc = w + L.o_wcopy;
    auto WC = [&](int64_t off) { return (const void*)(wc + off * esz); };
    a.W1s = WC(L.W1s); a.W2s = WC(L.W2s); a.Wmvs = WC(L.Wmvs); a.W3s = WC(L.W3s); a.W4s = WC(L.W4s); a.W5s = WC(L.W5s);
    a.W5t = WC(L.W5t); a.W4t = WC(L.W4t); a.W3zt = WC(L.W3zt); a.Wmvt = WC(L.Wmvt); a.W2t = WC(L.W2t);
    a.b1 = params + plan->tensor_offset[1]; a.b2 = params + plan->tensor_offset[3];
    a.bmu = params + plan->tensor_offset[5]; a.blv = params + plan->tensor_offset[7];
    a.b3 = params + plan->tensor_offset[9]; a.b4 = params + plan->tensor_offset[11]; a.b5 = params + plan->tensor_offset[13];
    a.w5last = params + plan->tensor_offset[12] + (int64_t)(XD - 1) * HD;
    if (L.info) {
        a.Wc1s = WC(L.Wc1s); a.Wc2s = WC(L.Wc2s); a.Wc2t = WC(L.Wc2t); a.Wa1s = WC(L.Wa1s); a.Wa1t = WC(L.Wa1t); a.Wa2s = WC(L.Wa2s); a.Wa2t = WC(L.Wa2t);
        a.bc1 = params + plan->tensor_offset[15]; a.bc2 = params + plan->tensor_offset[17]; a.wc3 = params + plan->tensor_offset[18]; a.bc3 = params + plan->tensor_offset[19];
        a.ba1 = params + plan->tensor_offset[21]; a.ba2 = params + plan->tensor_offset[23]; a.wa3 = params + plan->tensor_offset[24]; a.ba3 = params + plan->tensor_offset[25];
        a.alpha = (float)plan->info_alpha; a.beta = (float)plan->info_beta; a.gamma = (float)plan->info_gamma;
    }
    a.partials = (double*)(w + L.o_partials);
    a.wcopy = wc; a.wcopy_bytes = L.wcopy_elems * esz * planes_of(plan->precision);
    a.spl = L.stash_rows * plan->Bp; a.wpl_bytes = (unsigned)(L.wcopy_elems * esz);
    a.dbg = g_dbg;
    { const char* ab = getenv("DVAE_ABLATE"); a.ablate = ab ? atoi(ab) : 0; }
    if (g_defer_req.on) {
        const PendingUpdate& u = g_defer_req.u;
        a.defer.on = 1; a.defer.have = g_defer_req.have ? 1 : 0;
        { const char* e = getenv("DVAE_DEFER_DIAG"); a.defer.diag = e ? atoi(e) : 0; }
        a.defer.a = g_defer_req.have ? make_apply_args(plan, L, u.params, u.m, u.v, w, u.n_slabs, true, u.step, u.lr, u.beta1, u.beta2, u.adam_eps, 1.0, g_defer_req.losses3)
                                     : make_apply_args(plan, L, const_cast<float*>(params), nullptr, nullptr, w, 1, false, 1, 0, 0, 0, 0, 0, g_defer_req.losses3);
        a.defer.tasks = (const DeferTask*)(w + L.o_defer); a.defer.ntasks = defer_state_get(ws).ntasks;
        a.defer.shard = (unsigned*)(w + L.o_defer + DEFER_O_SHARD);
        a.defer.done = (unsigned*)(w + L.o_defer + DEFER_O_DONE); a.defer.err = a.defer.done + 1;
        a.defer.seq_arrive = g_defer_req.seq_arrive; a.defer.seq_done = g_defer_req.seq_done;
        long long ms = 2000;                                   // bound of the arrival wait (the whole grid is resident: it is microseconds)
        { const char* e = getenv("DVAE_DEFER_TIMEOUT_MS"); if (e) ms = atoll(e); }
        a.defer.timeout_ticks = (unsigned long long)(ms < 0 ? 0 : ms) * 100000ull;
    }
    // DVAE_RAW_INPUTS=1 (opt-in, tested): the weight-gradient kernel takes x and the labels straight from the fp32 input matrices and the
    // rows kernel writes no stash for them (35 MB of writes less in its HBM-bound opening window).  Measured (M2 y513, 8192 frames, bf16x3,
    // same box, alternating): rows 46.0 -> 44.0 us, but the weight-gradient kernel 29.0 -> 34.6 us -- its input-fed blocks convert and
    // transpose 8 KB per k-step and wave through LDS behind a two-deep ring -- so the stash stays the default (step 78.6 vs 81.8 us).
    // Round 4, large batches: from ~1e5 frames on the rows kernel runs many tiles per workgroup and sets the step time (4.7 ns per frame against
    // 2.5 for the weight-gradient kernel, which is HBM-bound there), so the 6.3 KB per frame of input stash it no longer writes pay:
    // B = 262 144: 139.3 -> 143.2 M frames/s, B = 2^20: 137.0 -> 142.8 (x only: 143.3 / 139.1; profiles/r04_bigb_raw.txt).  Chosen
    // automatically from DVAE_RAW_AUTO_B frames on (131 072); DVAE_RAW_INPUTS=0 keeps the stash, =x / =1 force a variant at any size.
    const char* raw_env = getenv("DVAE_RAW_INPUTS");
    const bool raw_possible = plan->rows_kernel >= 2 && rows2_supported(plan->precision, plan->model) && a.rows == nullptr &&
                              wgrad_form(getenv("DVAE_WGRAD")) == 4 && g_mode.mode != 1;
    int64_t raw_auto_b = 131072;
    { const char* e = getenv("DVAE_RAW_AUTO_B"); if (e) raw_auto_b = atoll(e); }
    const bool raw_auto = raw_env == nullptr && raw_possible && g_mode.mode == 0 && plan->B >= raw_auto_b;
    const bool raw_inputs = raw_possible && ((raw_env != nullptr && strcmp(raw_env, "0") != 0) || raw_auto);
    // DVAE_RAW_INPUTS=x: only the x tile is read raw (the label stash stays: one plane for binary labels); any other value: x and labels
    const int raw_mask = !raw_inputs ? 0 : ((raw_env != nullptr && strcmp(raw_env, "x") == 0) ? 1 : 3);
    a.stash_inputs = 3 & ~raw_mask;
    a.mode = g_mode.mode;
    {   // label lo plane on demand: 8-wave kernel + the workgroup k-split weight-gradient kernel, split-bf16 operands, labels from the stash
        static std::atomic<unsigned> launch_counter{1};
        const char* wk0 = getenv("DVAE_WGRAD");
        const bool wg4 = wgrad_form(wk0) == 4 && (plan->Bp > 128 || raw_inputs || wk0 != nullptr);
        a.ylo_epoch = (unsigned*)(w + L.o_flags);
        a.ylo_dirty = (int*)(w + L.o_flags + 1024);
        // (the second launch of a grouped step belongs to the rows kernel of the first call: same id, or its label blocks would misread the epoch word)
        static thread_local unsigned last_id = 0;
        static thread_local const void* last_ws = nullptr;
        if (g_w4_group == 1) {
            DVAE_CHECK_ARG(last_ws == ws && last_id != 0, "train_grads_group: group 1 must follow a group 0 call on the same workspace");
            a.launch_id = last_id;
        } else {
            a.launch_id = launch_counter.fetch_add(1, std::memory_order_relaxed);
            if (a.launch_id == 0) a.launch_id = launch_counter.fetch_add(1, std::memory_order_relaxed);      // 0 = the memset value of a fresh workspace
            last_id = a.launch_id; last_ws = ws;
        }
        a.ylo_skip = (x3 && plan->y_dim > 0 && plan->rows_kernel >= 2 && rows2_supported(plan->precision, plan->model) && wg4 && !(raw_mask & 2) &&
                      getenv("DVAE_YLO_ALWAYS") == nullptr) ? 1 : 0;
    }
    a.out_r = g_mode.out_r; a.out_mu = g_mode.out_mu; a.out_lv = g_mode.out_lv; a.out_z = g_mode.out_z; a.ld_r = g_mode.ld_r;
    a.g_r = g_mode.g_r; a.g_mu = g_mode.g_mu; a.g_lv = g_mode.g_lv; a.g_z = g_mode.g_z; a.ld_gr = g_mode.ld_gr;
    DVAE_CHECK_ARG(a.mode == 0 || plan->rows_kernel == 2, "rows-kernel modes 1 / 2 exist in the 8-wave kernel only (plan->rows_kernel == 2)");
    char* st = w + L.o_stash;
    auto ST = [&](int64_t row) { return (void*)(st + row * plan->Bp * esz); };
    a.xT = ST(L.xT); a.yT = ST(L.yT); a.h1T = ST(L.h1T); a.h2T = ST(L.h2T); a.dh1T = ST(L.dh1T); a.dh2T = ST(L.dh2T);
    a.dmlvT = ST(L.dmlvT); a.zT = ST(L.zT); a.d1T = ST(L.d1T); a.d2T = ST(L.d2T); a.dd1T = ST(L.dd1T); a.dd2T = ST(L.dd2T); a.daT = ST(L.daT);
    if (L.info) {
        a.c1T = ST(L.c1T); a.c2T = ST(L.c2T); a.dc1T = ST(L.dc1T); a.dc2T = ST(L.dc2T); a.dc3T = ST(L.dc3T);
        a.a1T = ST(L.a1T); a.a2T = ST(L.a2T); a.da1T = ST(L.da1T); a.da2T = ST(L.da2T); a.da3T = ST(L.da3T);
    }
    const int grid = (int)plan->rows_grid;
    const bool m2 = plan->model == DVAE_MODEL_M2;
    const int gsel = g_w4_group;                                          // dvae_train_grads_group: -1 the whole step, 0 rows + first wgrad launch, 1 second wgrad launch
    DVAE_CHECK_ARG(gsel < 0 || (w4_grouped(*plan) && g_mode.mode == 0 && !g_eval_only), "train_grads_group: the plan was not made for two weight-gradient launches (DVAE_EXCHANGE_GROUPS=2 when the plan is made)");
    int rc = 0;
    if (gsel != 1) {
        ProfScope ps(s, 0);
        if (plan->rows_kernel == 3 && rows3_supported(plan->precision, plan->model)) {
            rc = launch_rows3(plan->model, plan->y_dim, a, grid, s);
        } else if (plan->rows_kernel == 2 && rows2_supported(plan->precision, plan->model)) {
            rc = launch_rows2(plan->precision, plan->model, plan->y_dim, a, grid, s);
#ifdef DVAE_DIAG
        } else if (x3) {
            if (L.info) rc = launch_rows<PolX3, 16, false, true>(a, grid, s);
            else if (!m2) rc = launch_rows<PolX3, 0, false>(a, grid, s);
            else if (plan->y_dim == 1) rc = launch_rows<PolX3, 16, true>(a, grid, s);
            else rc = launch_rows<PolX3, 528, true>(a, grid, s);
        } else if (bf) {
            if (L.info) rc = launch_rows<PolBF16, 16, false, true>(a, grid, s);
            else if (!m2) rc = launch_rows<PolBF16, 0, false>(a, grid, s);
            else if (plan->y_dim == 1) rc = launch_rows<PolBF16, 16, true>(a, grid, s);
            else rc = launch_rows<PolBF16, 528, true>(a, grid, s);
#else
        } else if (x3 || bf) {
            set_error("train_grads: the 4-wave rows kernel under the bf16 policies exists in the diagnostic build only (build.py --diag)");
            rc = DVAE_E_UNSUPPORTED;
#endif
        } else {
            if (L.info) rc = launch_rows<PolF32, 16, false, true>(a, grid, s);
            else if (!m2) rc = launch_rows<PolF32, 0, false>(a, grid, s);
            else if (plan->y_dim == 1) rc = launch_rows<PolF32, 16, true>(a, grid, s);
            else rc = launch_rows<PolF32, 528, true>(a, grid, s);
        }
    }
    if (rc) return rc;
    if (g_eval_only || a.mode == 1) return 0;
    const int64_t kper = kper_of(plan);
    const int ks = w4_classed(*plan) ? plan->ksplit : (int)((plan->Bp + kper - 1) / kper);      // slabs the launch fills (class-sliced: the largest slice count)
    DVAE_CHECK_ARG(ks <= plan->ksplit, "train_grads: internal k-split mismatch");
    DVAE_CHECK_ARG(!w4_classed(*plan) || wgrad_form(getenv("DVAE_WGRAD")) == 4,
                   "train_grads: the plan was made for the workgroup k-split weight-gradient kernel (class-sliced schedule); DVAE_WGRAD changed since");
    float* slabs = (float*)(w + L.o_grads);
    // DVAE_WGRAD=lds selects the workgroup-blocked kernel for the bf16 policies (operands staged once per 4 x 4 block in LDS: half the
    // L2 -> CU operand traffic).  Measured (M2 y513, 8192 frames): 39.3 vs 32.9 us under bf16x3, 25.0 vs 21.4 us under bf16 -- SLOWER than the
    // register-ring kernel at every k-split tried, and both kernels take the same time on a stash that is already cache-warm: the
    // weight-gradient pass is bound neither by operand traffic nor by cold reads.  The register-ring kernel stays default.
    const char* wk = getenv("DVAE_WGRAD");
    int wrep = 1;
    { const char* e = getenv("DVAE_WGRAD_REPEAT"); if (e) { wrep = atoi(e); if (wrep < 1) wrep = 1; } }   // diagnostic: re-run on the warm stash
    for (int rep = 0; rep < wrep; ++rep)
    if (wgrad_form(wk) == 4 && (plan->Bp > 128 || raw_inputs || wk != nullptr)) {      // one 128-frame slice: the 2 x 2 kernel's short epilogue wins (11.2 vs 12.8 us)
      // one launch (table 0), or the launches of a grouped plan: tables 1 and 2, both (gsel < 0) or the one asked for
      const int tb0 = !w4_grouped(*plan) ? 0 : (gsel == 1 ? 2 : 1), tb1 = !w4_grouped(*plan) ? 0 : (gsel == 0 ? 1 : 2);
      for (int tb = tb0; tb <= tb1; ++tb) {
        ProfScope ps(s, rep == 0 ? 1 : 2);
        const int w4grid = w4_grid_get(w, tb);
        DVAE_CHECK_ARG(w4grid > 0, "train_grads: workspace was not set up by dvae_train_init");
        const dim3 g3((unsigned)w4grid);                                  // one workgroup per item of the host's table (w4_build_items)
        RawIn ri;
        ri.x = x; ri.y = y; ri.ldx = ldx; ri.ldy = plan->y_dim ? ldy : 0; ri.B = plan->B;
        const int use_raw = raw_mask;
        static bool attr_done[64][3] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) dev = 0;
        const int pi = x3 ? 2 : (bf ? 1 : 0);
        if (!attr_done[dev][pi]) {
            if (x3) DVAE_HIP(hipFuncSetAttribute((const void*)wgrad4_kernel<PolX3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg4<PolX3>::BYTES));
            else if (bf) DVAE_HIP(hipFuncSetAttribute((const void*)wgrad4_kernel<PolBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg4<PolBF16>::BYTES));
            else DVAE_HIP(hipFuncSetAttribute((const void*)wgrad4_kernel<PolF32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wg4<PolF32>::BYTES));
            attr_done[dev][pi] = true;
        }
        const Block4* bl = (const Block4*)(w + L.o_blocks4);
        const W4Item* w4items = (const W4Item*)(w + L.o_items4) + (size_t)tb * W4_MAX_ITEMS;
        // the optimizer step in this launch's tail (dvae_train_step asked for it): only when every workgroup of the grid is resident at
        // once -- one per CU, the tail's wait depends on it -- and the block counters fit the flag header
        ApplyArgs fa_apply;
        memset(&fa_apply, 0, sizeof(fa_apply));
        FoldArgs fold{nullptr, 0u, 0u};
        if (g_fold.want && !w4_classed(*plan) && wrep == 1 && a.mode == 0 && ks > 1 && ks <= 16 && L.nblocks4 <= FOLD_MAXB && (int)g3.x <= device_cu_count(dev)) {
            fa_apply = make_apply_args(plan, L, g_fold.params, g_fold.m, g_fold.v, w, ks, true, g_fold.step, g_fold.lr, g_fold.beta1, g_fold.beta2,
                                       g_fold.adam_eps, 1.0, g_fold.losses3);
            fold.cnt = (unsigned*)(w + L.o_flags);
            fold.target = (unsigned)ks * fold_seq_next(w);
            fold.max_polls = 1u << 20;
            { const char* e = getenv("DVAE_FOLD_MAX_POLLS"); if (e) fold.max_polls = (unsigned)strtoul(e, nullptr, 10); }
            g_fold.done = true;
        }
        int fin_block = -1;
        const unsigned* fin_err = nullptr;
        dim3 g3l = g3;
        if (a.defer.on) {                                             // + one workgroup that turns the rows kernel's partial sums into the loss scalars
            fa_apply = a.defer.a;
            fin_block = (int)g3.x; fin_err = a.defer.err;
            g3l = dim3(g3.x + 1);
        }
        if (x3) hipLaunchKernelGGL((wgrad4_kernel<PolX3>), g3l, dim3(256), Wg4<PolX3>::BYTES, s, bl, w4items, ks, plan->Bp, a.spl, slabs, plan->n_params, ri, use_raw, a.ylo_skip ? a.ylo_epoch : nullptr, a.launch_id, fa_apply, fold, fin_block, fin_err);
        else if (bf) hipLaunchKernelGGL((wgrad4_kernel<PolBF16>), g3l, dim3(256), Wg4<PolBF16>::BYTES, s, bl, w4items, ks, plan->Bp, a.spl, slabs, plan->n_params, ri, use_raw, (const unsigned*)nullptr, 0u, fa_apply, fold, fin_block, fin_err);
        else hipLaunchKernelGGL((wgrad4_kernel<PolF32>), g3l, dim3(256), Wg4<PolF32>::BYTES, s, bl, w4items, ks, plan->Bp, a.spl, slabs, plan->n_params, ri, use_raw, (const unsigned*)nullptr, 0u, fa_apply, fold, fin_block, fin_err);
        DVAE_LAUNCH_OK("wgrad4_kernel");
      }
#ifdef DVAE_DIAG
    } else if ((bf || x3) && wk && strcmp(wk, "lds") == 0) {
        ProfScope ps(s, rep == 0 ? 1 : 2);
        const dim3 g3((unsigned)(L.nblocks * ks));
        static bool attr_done[64][2] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64) dev = 0;
        if (x3) {
            if (!attr_done[dev][1]) { DVAE_HIP(hipFuncSetAttribute((const void*)wgrad_lds_kernel<PolX3>, hipFuncAttributeMaxDynamicSharedMemorySize, WgLds<PolX3>::BYTES)); attr_done[dev][1] = true; }
            hipLaunchKernelGGL((wgrad_lds_kernel<PolX3>), g3, dim3(256), WgLds<PolX3>::BYTES, s, (const BlockDesc*)(w + L.o_blocks), L.nblocks, ks, plan->Bp, a.spl, kper, slabs, plan->n_params);
        } else {
            if (!attr_done[dev][0]) { DVAE_HIP(hipFuncSetAttribute((const void*)wgrad_lds_kernel<PolBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, WgLds<PolBF16>::BYTES)); attr_done[dev][0] = true; }
            hipLaunchKernelGGL((wgrad_lds_kernel<PolBF16>), g3, dim3(256), WgLds<PolBF16>::BYTES, s, (const BlockDesc*)(w + L.o_blocks), L.nblocks, ks, plan->Bp, a.spl, kper, slabs, plan->n_params);
        }
        DVAE_LAUNCH_OK("wgrad_lds_kernel");
#else
    } else if ((bf || x3) && wk && strcmp(wk, "lds") == 0) {
        set_error("train_grads: DVAE_WGRAD=lds exists in the diagnostic build only (build.py --diag)");
        return DVAE_E_UNSUPPORTED;
#endif
    } else {
    int GPW = 2;                                            // groups (waves) per workgroup
    { const char* e = getenv("DVAE_GPW"); if (e) { GPW = atoi(e); if (GPW < 1 || GPW > 4) GPW = 2; } }   // diagnostic override
    const dim3 g2((unsigned)(((L.ntiles + GPW - 1) / GPW) * ks));
    {
        ProfScope ps(s, rep == 0 ? 1 : 2);
        if (x3) hipLaunchKernelGGL((wgrad_kernel<PolX3>), g2, dim3(64 * GPW), 0, s, (const GroupDesc*)(w + L.o_tiles), L.ntiles, ks, plan->Bp, a.spl, kper, slabs, plan->n_params);
        else if (bf) hipLaunchKernelGGL((wgrad_kernel<PolBF16>), g2, dim3(64 * GPW), 0, s, (const GroupDesc*)(w + L.o_tiles), L.ntiles, ks, plan->Bp, a.spl, kper, slabs, plan->n_params);
        else hipLaunchKernelGGL((wgrad_kernel<PolF32>), g2, dim3(64 * GPW), 0, s, (const GroupDesc*)(w + L.o_tiles), L.ntiles, ks, plan->Bp, a.spl, kper, slabs, plan->n_params);
    }
    DVAE_LAUNCH_OK("wgrad_kernel");
    }
    if (reduce_slabs && ks > 1) {
        ProfScope ps(s, 2);
        int64_t lo = 0, hi = plan->n_params;                              // a group's launch reduces its own part of the flat gradient
        if (gsel == 0) lo = plan->tensor_offset[8];
        if (gsel == 1) hi = plan->tensor_offset[8];
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(512), dim3(256), 0, s, slabs + lo, hi - lo, ks, plan->n_params);
        DVAE_LAUNCH_OK("slab_reduce_kernel");
    }
    return 0;
}

// The step's gradient pass in two calls, for a plan made under DVAE_EXCHANGE_GROUPS=2: group 0 = the rows kernel + the weight-gradient launch of
// the decoder-side tensors (flat gradient [tensor_offset[8], n_params)), group 1 = the launch of the encoder's ([0, tensor_offset[8])); the
// caller may start the exchange of group 0's part between the two calls.  dvae_train_grads on such a plan runs both.
extern "C" int dvae_train_grads_group(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx, const float* y, int ldy,
                                      const float* eps_noise, float elbo_eps, int group, int reduce_slabs, void* stream) {
    DVAE_CHECK_ARG(plan && (group == 0 || group == 1), "train_grads_group: group must be 0 or 1");
    DVAE_CHECK_ARG(w4_grouped(*plan), "train_grads_group: the plan was not made for two weight-gradient launches (DVAE_EXCHANGE_GROUPS=2 when the plan is made)");
    g_w4_group = group;
    const int rc = dvae_train_grads(plan, params, ws, x, ldx, y, ldy, eps_noise, elbo_eps, reduce_slabs, stream);
    g_w4_group = -1;
    return rc;
}

// float range [lo, hi) of a group's part of the flat gradient (group < 0: everything); *ngroups = 2 for a plan made for two launches, else 1
extern "C" int dvae_train_group_range(const dvae_train_plan_t* plan, int group, int64_t* lo, int64_t* hi, int* ngroups) {
    DVAE_CHECK_ARG(plan && lo && hi && group >= -1 && group <= 1, "train_group_range: bad argument");
    *lo = 0; *hi = plan->n_params;
    if (group == 0) *lo = plan->tensor_offset[8];
    if (group == 1) *hi = plan->tensor_offset[8];
    if (ngroups) *ngroups = w4_grouped(*plan) ? 2 : 1;
    return 0;
}

static int used_slabs(const dvae_train_plan_t* plan) {
    if (w4_classed(*plan)) return plan->ksplit;                          // class-sliced schedule: the largest slice count of any block
    const int64_t kper = kper_of(plan);
    return (int)((plan->Bp + kper - 1) / kper);
}

extern "C" int dvae_train_apply(const dvae_train_plan_t* plan, float* params, float* m, float* v, void* ws, int n_slabs,
                                int step, double lr, double beta1, double beta2, double adam_eps, double grad_scale,
                                float* losses3, void* stream) {
    DVAE_CHECK_ARG(plan && params && m && v && ws && step >= 1, "train_apply: bad argument");
    { const int frc = dvae_train_flush(plan, ws, stream); if (frc) return frc; }
    Layout L;
    make_layout(*plan, L);
    if (n_slabs <= 0) n_slabs = used_slabs(plan);
    DVAE_CHECK_ARG(n_slabs <= plan->ksplit, "train_apply: n_slabs %d > plan ksplit %d", n_slabs, plan->ksplit);
    ProfScope ps((hipStream_t)stream, 3);
    return launch_apply(plan, L, params, m, v, (char*)ws, n_slabs, true, step, lr, beta1, beta2, adam_eps, grad_scale, losses3,
                        (hipStream_t)stream);
}

extern "C" int dvae_train_step(const dvae_train_plan_t* plan, float* params, float* m, float* v, void* ws,
                               const float* x, int ldx, const float* y, int ldy, const float* eps_noise, float elbo_eps,
                               int step, double lr, double beta1, double beta2, double adam_eps, float* losses3, void* stream) {
    DVAE_CHECK_ARG(plan && params && m && v && ws && step >= 1, "train_step: bad argument");
    g_rng_step_override = step;
    // DVAE_FOLD_APPLY=1: the optimizer step in the tail of the weight-gradient launch (two launches per step; results bit-identical,
    // tested).  Opt-in, because it is SLOWER on the MI355X (M2 y513, 8192 frames, bf16x3, same box, alternating): the weight-gradient
    // kernel goes 28.7 -> 47.1 us while the separate optimizer launch it replaces costs 9.5 us gross.  Ablation of the 18.4 us tail
    // (FOLD_DIAG builds): arrive + wait for the block's slices 2.6 us; the 13 MB of slab partials + p, m, v as device-coherent loads
    // 8 us; Adam + weight-copy element math and stores 8 us -- the kernel runs ONE wave per SIMD (512 registers of accumulators), so
    // both phases are latency-bound, whereas apply_kernel does the same work at full occupancy in ~6 us behind a ~3 us launch gap.
    // (First attempt with release / acquire fences instead of write-through stores + coherent loads: 75 us -- every fence writes back
    // or invalidates the XCD's whole L2 under the workgroups that are still multiplying.)
    const char* fe = getenv("DVAE_FOLD_APPLY");                       // read per call (tests flip it)
    const bool fold_on = kDiagBuild && fe && atoi(fe) != 0;      // (the folded tail exists in -DDVAE_DIAG builds only)
    g_fold = FoldRequest();
    g_fold.want = fold_on; g_fold.params = params; g_fold.m = m; g_fold.v = v; g_fold.step = step; g_fold.lr = lr; g_fold.beta1 = beta1;
    g_fold.beta2 = beta2; g_fold.adam_eps = adam_eps; g_fold.losses3 = losses3;
    int rc = dvae_train_grads(plan, params, ws, x, ldx, y, ldy, eps_noise, elbo_eps, 0, stream);
    const bool folded = g_fold.done;
    g_fold = FoldRequest();
    g_rng_step_override = -1;
    if (rc) return rc;
    if (folded) return 0;
    return dvae_train_apply(plan, params, m, v, ws, 0, step, lr, beta1, beta2, adam_eps, 1.0, losses3, stream);
}

// Two launches per step: rows(n) applies the update of step n - 1 in its opening and finalises the losses of step n at its end, wgrad(n)
// leaves the gradient slabs of step n; the update of step n stays PENDING until the next deferred step or dvae_train_flush.
extern "C" int dvae_train_step_deferred(const dvae_train_plan_t* plan, float* params, float* m, float* v, void* ws,
                                        const float* x, int ldx, const float* y, int ldy, const float* eps_noise, float elbo_eps,
                                        int step, double lr, double beta1, double beta2, double adam_eps, float* losses3, void* stream) {
    DVAE_CHECK_ARG(plan && params && m && v && ws && step >= 1 && losses3, "train_step_deferred: bad argument");
    if (!dvae_train_can_defer(plan, ws))
        return dvae_train_step(plan, params, m, v, ws, x, ldx, y, ldy, eps_noise, elbo_eps, step, lr, beta1, beta2, adam_eps, losses3, stream);   // flushes first
    DeferState st = defer_state_get(ws);
    g_defer_req = DeferRequest();
    g_defer_req.on = true; g_defer_req.have = st.pending; g_defer_req.u = st.u; g_defer_req.losses3 = losses3;
    g_defer_req.seq_arrive = st.seq_arrive + (st.pending ? 1u : 0u); g_defer_req.seq_done = st.seq_done + 1u;
    g_rng_step_override = step;
    const int rc = dvae_train_grads(plan, params, ws, x, ldx, y, ldy, eps_noise, elbo_eps, 0, stream);
    g_defer_req = DeferRequest();
    g_rng_step_override = -1;
    if (rc) return rc;
    st.seq_arrive += st.pending ? 1u : 0u; st.seq_done += 1u;
    st.pending = true;
    st.u = PendingUpdate{params, m, v, step, lr, beta1, beta2, adam_eps, used_slabs(plan)};
    defer_state_put(ws, st);
    return 0;
}

/* ---- whole-model autograd path of the drop-in modules (packages/models/models.py) ---- */
extern "C" int dvae_module_forward(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                                   const float* y, int ldy, const float* eps_noise, float* out_r, int ld_r, float* out_mu,
                                   float* out_lv, float* out_z, int repack, void* stream) {
    DVAE_CHECK_ARG(plan && params && ws && x && eps_noise && out_r && out_mu && out_lv && ld_r >= XD, "module_forward: bad argument");
    DVAE_CHECK_ARG(plan->rows_kernel == 2 && plan->row_index == 0, "module_forward: needs the 8-wave rows kernel and no gather table");
    { const int frc = dvae_train_flush(plan, ws, stream); if (frc) return frc; }
    if (repack) { int rc = dvae_train_repack(plan, params, ws, stream); if (rc) return rc; }
    g_mode = ModeArgs();
    g_mode.mode = 1; g_mode.out_r = out_r; g_mode.ld_r = ld_r; g_mode.out_mu = out_mu; g_mode.out_lv = out_lv; g_mode.out_z = out_z;
    const int rc = dvae_train_grads(plan, params, ws, x, ldx, y, ldy, eps_noise, 0.f, 0, stream);
    g_mode = ModeArgs();
    return rc;
}

extern "C" int dvae_module_backward(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                                    const float* y, int ldy, const float* eps_noise, const float* g_r, int ld_gr,
                                    const float* g_mu, const float* g_lv, const float* g_z, float* grad_flat, int accumulate,
                                    void* stream) {
    DVAE_CHECK_ARG(plan && params && ws && x && eps_noise && grad_flat, "module_backward: bad argument");
    DVAE_CHECK_ARG(plan->rows_kernel == 2 && plan->row_index == 0, "module_backward: needs the 8-wave rows kernel and no gather table");
    DVAE_CHECK_ARG(g_r == nullptr || ld_gr >= XD, "module_backward: ld_gr < 513");
    g_mode = ModeArgs();
    g_mode.mode = 2; g_mode.g_r = g_r; g_mode.ld_gr = ld_gr; g_mode.g_mu = g_mu; g_mode.g_lv = g_lv; g_mode.g_z = g_z;
    const int rc = dvae_train_grads(plan, params, ws, x, ldx, y, ldy, eps_noise, 0.f, 0, stream);
    g_mode = ModeArgs();
    if (rc) return rc;
    Layout L;
    make_layout(*plan, L);
    const float* slabs = (const float*)((const char*)ws + L.o_grads);
    hipLaunchKernelGGL(slab_sum_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, slabs, plan->n_params, used_slabs(plan), plan->n_params,
                       grad_flat, accumulate);
    DVAE_LAUNCH_OK("slab_sum_kernel");
    return 0;
}

extern "C" int dvae_train_debug_stamps(void* buf) {
    g_dbg = (unsigned long long*)buf;
    return 0;
}

extern "C" int dvae_train_eval(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                               const float* y, int ldy, const float* eps_noise, float elbo_eps, float* losses3, void* stream) {
    DVAE_CHECK_ARG(plan && params && ws && x && losses3, "train_eval: bad argument");
    // forward + loss sums only: the rows kernel also writes the stash, which is simply not consumed
    Layout L;
    make_layout(*plan, L);
    const int saved = 0;
    (void)saved;
    // run the rows kernel through dvae_train_grads' argument setup, but stop before the wgrad launch
    g_eval_only = true;
    int rc = dvae_train_grads(plan, params, ws, x, ldx, y, ldy, eps_noise, elbo_eps, 0, stream);
    g_eval_only = false;
    if (rc) return rc;
    ApplyArgs a;
    memset(&a, 0, sizeof(a));
    char* w = (char*)ws;
    a.partials = (const double*)(w + L.o_partials); a.npartials = (int)plan->rows_grid; a.B = plan->B; a.losses3 = losses3; a.accum = (double*)(uintptr_t)plan->loss_accum;
    a.info = plan->model == DVAE_MODEL_M2_INFO; a.alpha = (float)plan->info_alpha; a.beta = (float)plan->info_beta; a.gamma = (float)plan->info_gamma;
    hipLaunchKernelGGL((apply_kernel<float, true>), dim3(1), dim3(256), 0, (hipStream_t)stream, a);   // grid of 1 = the loss block only
    DVAE_LAUNCH_OK("apply_kernel(loss only)");
    return 0;
}

namespace dvae { namespace fused {
__global__ __launch_bounds__(256) void noise_kernel(unsigned long long seed, unsigned long long step, int64_t B, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one (frame, half) pair per thread
    if (i >= 2 * B) return;
    const int64_t frame = i >> 1; const int h = (int)(i & 1);
    float e[8];
    frame_noise8(seed, (unsigned long long)frame, step, h, e);
#pragma unroll
    for (int j = 0; j < 4; ++j) { out[frame * ZD + 4 * h + j] = e[j]; out[frame * ZD + 8 + 4 * h + j] = e[4 + j]; }
}
} }

extern "C" int dvae_train_noise(const dvae_train_plan_t* plan, uint64_t step, float* eps_out, void* stream) {
    DVAE_CHECK_ARG(plan && eps_out, "train_noise: null argument");
    hipLaunchKernelGGL(dvae::fused::noise_kernel, dim3((unsigned)((2 * plan->B + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)plan->rng_seed, (unsigned long long)step, plan->B, eps_out);
    DVAE_LAUNCH_OK("noise_kernel");
    return 0;
}

extern "C" int dvae_train_profile(int enable) {
    g_prof = enable != 0;
    return 0;
}

extern "C" int dvae_train_profile_read(double ms[4], int64_t calls[4]) {
    hipError_t e = hipDeviceSynchronize();
    for (int i = 0; i < g_npending; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_pending[i].a, g_pending[i].b) == hipSuccess) {
            g_ms[g_pending[i].which] += (double)t;
            g_calls[g_pending[i].which] += 1;
        }
        (void)hipEventDestroy(g_pending[i].a);
        (void)hipEventDestroy(g_pending[i].b);
    }
    g_npending = 0;
    for (int i = 0; i < 4; ++i) { ms[i] = g_ms[i]; calls[i] = g_calls[i]; g_ms[i] = 0; g_calls[i] = 0; }
    DVAE_HIP(e);
    return 0;
}
