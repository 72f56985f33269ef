// Optimizer step of the fused train step (scripts/training_M2.py:146-147: torch.optim.Adam.step on every parameter, torch's op order,
// SURVEY.md 8a-13) and the refresh of the kernel-layout weight copies: the ONE definition behind
//   * apply_kernel (train_fused.hip): one thread per parameter, its own launch -- the three-launch step, the multi-GPU step, flushes;
//   * the folded tail of wgrad4_kernel (diagnostic builds);
//   * the DEFERRED step (round 4): the update of step n runs in the opening of step n + 1's rows kernel, on the chain waves that would
//     otherwise wait for the x tile (train_rows2.hip) -- two launches per step.
// All three perform the same element arithmetic on the same fixed-order slab sums: results are bit-identical (tested).
#pragma once
#include <type_traits>
#include "fused_tiles.hpp"
#include "rows_common.hpp"

namespace dvae {
namespace fused {

// torch.optim.Adam on one element (torch's op order: eps is added after the division by sqrt(bias_correction2))
__device__ __forceinline__ void adam_element(const ApplyArgs& g, float& pi, float m_old, float v_old, float gi, float& mi, float& vi) {
    gi *= g.gscale;
    mi = m_old + g.one_minus_b1 * (gi - m_old);
    vi = v_old * g.b2 + g.one_minus_b2 * (gi * gi);
    const float denom = sqrtf(vi) / g.bc2_sqrt + g.eps;
    pi = pi - g.step_size * (mi / denom);
}

// Adam update of parameter idx (gradient gi = the fixed-order slab sum) and the refresh of its kernel-layout weight copies
template <typename T, bool ADAM, int NP>
__device__ __forceinline__ void apply_element(const ApplyArgs& g, int64_t idx, const TensorDesc& d, float pi, float m_old, float v_old, float gi) {
    const int64_t i = idx - d.off;
    if (i >= (int64_t)d.rows * d.cols) return;
    T* wc = (T*)g.wcopy;
    if (ADAM) {
        float mi, vi;
        adam_element(g, pi, m_old, v_old, gi, mi, vi);
        g.p[idx] = pi; g.m[idx] = mi; g.v[idx] = vi;
    }
    const int r = (int)((unsigned)i / (unsigned)d.cols), c = (int)i - r * d.cols;      // a tensor holds far fewer than 2^31 elements
    constexpr int E = 16 / (int)sizeof(T), KS = 2 * E;
    if (d.sf_off >= 0) {
        const int rr = r + d.sf_roff, cc = c < d.sf_split ? c : c + d.sf_gap;
        const int64_t o = WFRAG ? d.sf_off + ((int64_t)((cc / KS) * d.sf_nt + (rr >> 5)) * 64 + ((cc % KS) / E) * 32 + (rr & 31)) * E + cc % E
                                : d.sf_off + (int64_t)rr * d.sf_ld + cc;
        const T ph = (T)pi;
        wc[o] = ph;
        if constexpr (NP == 2) wc[o + g.wpl] = (T)(pi - (float)ph);
    }
    if (d.st_off >= 0 && c < d.st_cmax) {
        const int rr = c, cc = r + d.st_roff;
        const int64_t o = WFRAG ? d.st_off + ((int64_t)((cc / KS) * d.st_nt + (rr >> 5)) * 64 + ((cc % KS) / E) * 32 + (rr & 31)) * E + cc % E
                                : d.st_off + (int64_t)rr * d.st_ld + cc;
        const T ph = (T)pi;
        wc[o] = ph;
        if constexpr (NP == 2) wc[o + g.wpl] = (T)(pi - (float)ph);
    }
}

// loss scalars from the rows kernel's per-workgroup partial sums: the arithmetic of the last step (one thread)
__device__ __forceinline__ void write_losses(const ApplyArgs& g, const double (*red)[4]) {
    const float recon = (float)((red[0][0] + red[1][0] + red[2][0] + red[3][0]) / (double)g.B);
    const float kl = (float)((red[0][1] + red[1][1] + red[2][1] + red[3][1]) / (double)g.B);
    g.losses3[0] = recon + kl; g.losses3[1] = recon; g.losses3[2] = kl;
    if (g.accum) { g.accum[0] += (double)(recon + kl); g.accum[1] += (double)recon; g.accum[2] += (double)kl; }
    if (g.info) {   // scripts/training_M2_info_vad.py:162-183
        const float bc = (float)((red[0][2] + red[1][2] + red[2][2] + red[3][2]) / (double)g.B);
        const float ba = (float)((red[0][3] + red[1][3] + red[2][3] + red[3][3]) / (double)g.B);
        const float classif = g.alpha * bc, aux_enc = g.beta * ba;
        g.losses3[3] = (recon + kl) + classif - aux_enc;      // enc_loss
        g.losses3[4] = classif;
        g.losses3[5] = g.gamma * ba;                          // aux_loss
        g.losses3[6] = aux_enc;
        g.losses3[7] = 0.f;
        if (g.accum) for (int q = 3; q < 8; ++q) g.accum[q] += (double)g.losses3[q];
    }
}
// (one 256-thread workgroup; every thread calls)
__device__ __forceinline__ void finalize_losses(const ApplyArgs& g, double (*red)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0, k = 0.0, c = 0.0, x = 0.0;
    for (int i = threadIdx.x; i < g.npartials; i += 256) {
        a += g.partials[4 * i]; k += g.partials[4 * i + 1]; c += g.partials[4 * i + 2]; x += g.partials[4 * i + 3];
    }
    a = wave_sum(a); k = wave_sum(k); c = wave_sum(c); x = wave_sum(x);
    if (lane == 0) { red[wave][0] = a; red[wave][1] = k; red[wave][2] = c; red[wave][3] = x; }
    __syncthreads();
    if (threadIdx.x == 0) write_losses(g, red);
}

// sum of the gradient slabs at flat index idx: every load issued before the first addition, additions in slab order (deterministic)
// COH: the slabs were written by other workgroups of THIS launch (write-through stores): device-coherent loads (sc1), which do not
// look at this XCD's L2 lines
template <bool COH = false>
__device__ __forceinline__ float slab_sum_at(const ApplyArgs& g, int64_t idx) {
    auto ld = [&](int64_t o) __attribute__((always_inline)) {
        if constexpr (COH) return __hip_atomic_load(g.slabs + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return g.slabs[o];
    };
    auto sum_slabs = [&](auto nsc) __attribute__((always_inline)) {
        constexpr int NS = decltype(nsc)::value;
        float part[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) part[k] = ld((k < g.nslabs ? k : 0) * g.slab_stride + idx);   // independent loads
        float t = part[0];
#pragma unroll
        for (int k = 1; k < NS; ++k) if (k < g.nslabs) t += part[k];                                    // fixed order: deterministic
        return t;
    };
    if (g.nslabs <= 8) return sum_slabs(std::integral_constant<int, 8>{});
    if (g.nslabs <= 12) return sum_slabs(std::integral_constant<int, 12>{});
    if (g.nslabs <= 16) return sum_slabs(std::integral_constant<int, 16>{});
    float gi = ld(idx);
    for (int k = 1; k < g.nslabs; ++k) gi += ld(k * g.slab_stride + idx);
    return gi;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The deferred optimizer step (dvae_train_step_deferred, include/dvae_train.h).
//
// Step n's launches are rows(n) + wgrad(n); its Adam update is PENDING when they end and runs at the top of rows(n + 1): the chain waves of
// that kernel have nothing to do until the helper waves have brought the x tile into LDS (2.8 - 5.5 us, HBM-bound), so they share the
// update out between them as TASKS -- a 32 x 32 tile of one weight matrix (or 1024 consecutive elements of a tensor without copies) per
// wave: slab sums in slab order, Adam, parameters and moments back, and the tile's kernel-layout copies as whole 16-byte fragments (the
// forward copy straight from registers: a lane's 16 consecutive parameters are one k-step of its row; the transposed copy through a 5 KB
// LDS tile per wave and the transposing LDS read the stash uses).  Every weight of step n + 1 depends on it, on every CU:
//   * parameters and copies leave as write-through (sc1) stores; a wave waits for its stores (vmcnt(0)) and adds 1 to one of 32 arrival
//     counters (separate 128-byte lines; fire and forget);
//   * chain wave 0 of every workgroup polls the 32 counters (one sc1 load per lane and poll) until each holds launch-number x its share
//     of the 4 x grid waves, bounded by wall time (a bound that runs out sets the error word: the step's loss becomes NaN);
//   * a workgroup barrier (BARR) lets the other waves through; only then does any wave request a weight fragment or a bias value, and
//     those loads are sc1 loads (MI355X_MICROARCH.md, hand-offs measured with sc1 loads in place of the acquire: stores all sc1, waited
//     for before the counter add, consumer polls with sc1 loads and loads after a barrier the poller joins).
// The loss scalars of a step no longer wait for the optimizer launch: the workgroup whose partial sums arrive LAST (a returning atomic at
// the very end of the rows kernel) reduces them with apply_kernel's own reduction shape.
// The grid must be resident at once (one workgroup per CU): the host defers only then; otherwise and for the last step of a run
// dvae_train_flush / any other entry point applies the pending update with apply_kernel.
// 16-byte write-through (sc1) store at a wave-uniform base + per-lane byte offset: compiler-visible (hazards, wait counts)
template <typename Frag>
__device__ __forceinline__ void store16_sc1(void* base_uniform, int voff_bytes, const Frag& f) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), rs, voff_bytes, 0, 16);
}

constexpr int DEFER_LDT = 40;     // LDS tile row stride (elements): 32 columns + 8 (80 bytes: 16-byte aligned rows)
template <typename T, int NP> struct DeferLds { static constexpr int wave_elems = NP * 32 * DEFER_LDT; static constexpr size_t bytes = (size_t)4 * wave_elems * sizeof(T); };

// one task on one wave (EXEC all ones); `tile` = this wave's LDS scratch
template <typename T, int NP>
__device__ __forceinline__ void defer_task(const ApplyArgs& g, const DeferTask tk, T* tile, int lane) {
    static_assert(sizeof(T) == 2, "deferred step: bf16 copies");
    typedef T Frag8 __attribute__((ext_vector_type(8)));
    const TensorDesc d = g.tensors[tk.tensor];                   // wave-uniform
    const int h = lane >> 5, row = lane & 31;
    const bool flat = tk.kind == 1;
    const int r = flat ? 0 : tk.r0 + row;
    const int c = flat ? tk.cbeg + 16 * lane : tk.cbeg + 16 * h;          // first of this lane's 16 consecutive elements
    int nvalid = (flat || r < d.rows) ? tk.cend - c : 0;
    nvalid = nvalid < 0 ? 0 : (nvalid > 16 ? 16 : nvalid);
    const int64_t base = nvalid > 0 ? d.off + (flat ? (int64_t)c : (int64_t)r * d.cols + c) : d.off;      // idle lanes read the tensor's first elements
    float pn[16];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int64_t b8 = nvalid > 8 * f ? base + 8 * f : d.off;
        // every load first (26 x 16 bytes at <= 12 slabs); rows of odd length are only 4-byte aligned: gfx950 takes dwordx4 at dword alignment
        f32x4 pv[2], mv[2], vv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            pv[q] = reinterpret_cast<const F4U*>(g.p + b8 + 4 * q)->v;
            mv[q] = reinterpret_cast<const F4U*>(g.m + b8 + 4 * q)->v;
            vv[q] = reinterpret_cast<const F4U*>(g.v + b8 + 4 * q)->v;
        }
        constexpr int NSM = 12;
        f32x4 sv[NSM][2];
#pragma unroll
        for (int k = 0; k < NSM; ++k)
#pragma unroll
            for (int q = 0; q < 2; ++q) sv[k][q] = reinterpret_cast<const F4U*>(g.slabs + (int64_t)(k < g.nslabs ? k : 0) * g.slab_stride + b8 + 4 * q)->v;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gi = sv[0][j >> 2][j & 3];
#pragma unroll
            for (int k = 1; k < NSM; ++k) if (k < g.nslabs) gi += sv[k][j >> 2][j & 3];       // slab order: apply_kernel's additions
            float pi = pv[j >> 2][j & 3], mi, vi;
            adam_element(g, pi, mv[j >> 2][j & 3], vv[j >> 2][j & 3], gi, mi, vi);
            const bool ok = 8 * f + j < nvalid;
            if (ok) {
                __hip_atomic_store(g.p + b8 + j, pi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: biases are read by this launch
                g.m[b8 + j] = mi; g.v[b8 + j] = vi;
            }
            pn[8 * f + j] = ok ? pi : 0.f;
        }
    }
    if (flat) return;                                                // wave-uniform
    T* const wc = (T*)g.wcopy;
    Frag8 fh[2], fl[2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const T ph = (T)pn[8 * f + j];
            fh[f][j] = ph;
            fl[f][j] = (T)(pn[8 * f + j] - (float)ph);
        }
    if (d.sf_off >= 0) {
        // forward copy: the lane's 16 parameters are k-step cc / 16 of row rr, its two 8-element fragments the k-step's two halves
        const int cc = c < d.sf_split ? c : c + d.sf_gap;            // (a tile never straddles the split: cbeg is the start of a column block + a multiple of 32)
        const int rr = r + d.sf_roff;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (nvalid > 8 * f) {
                const int64_t o = d.sf_off + ((int64_t)((cc >> 4) * d.sf_nt + (rr >> 5)) * 64 + f * 32 + (rr & 31)) * 8;
                store16_sc1(wc, (int)(o * (int64_t)sizeof(T)), fh[f]);
                if constexpr (NP == 2) store16_sc1(wc, (int)((o + g.wpl) * (int64_t)sizeof(T)), fl[f]);
            }
        }
    }
    if (d.st_off >= 0 && tk.cbeg < d.st_cmax) {                      // wave-uniform
        // transposed copy: tile[r local][c local] in LDS, read back as 8 consecutive r of one c (ds_read_b64_tr_b16)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            *reinterpret_cast<Frag8*>(tile + row * DEFER_LDT + 16 * h + 8 * f) = fh[f];
            if constexpr (NP == 2) *reinterpret_cast<Frag8*>(tile + 32 * DEFER_LDT + row * DEFER_LDT + 16 * h + 8 * f) = fl[f];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave's own writes (LDS operations of one wave complete in order)
        const int l31 = lane & 31, i16 = l31 & 15, q = i16 >> 2, pp = i16 & 3, cg = l31 >> 4;
        const T* blk = tile + q * DEFER_LDT + 16 * cg + 4 * pp;
        const int ct = tk.cbeg + l31;                                 // this lane's row of the transposed matrix
        const bool okc = ct < tk.cend && ct < d.st_cmax;
        typedef short s16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gq = h + 2 * i;                                 // rows 8 gq .. 8 gq + 7 of the tile
            Frag8 tf[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const T* bp = blk + pl * 32 * DEFER_LDT;
                const s16x4 r0v = lds_tr16(bp + (8 * gq) * DEFER_LDT), r1v = lds_tr16(bp + (8 * gq + 4) * DEFER_LDT);
                const s16x8 raw = {r0v[0], r0v[1], r0v[2], r0v[3], r1v[0], r1v[1], r1v[2], r1v[3]};
                tf[pl] = __builtin_bit_cast(Frag8, raw);
            }
            const int colT = tk.r0 + 8 * gq + d.st_roff;
            if (okc && tk.r0 + 8 * gq < d.rows) {
                const int64_t o = d.st_off + ((int64_t)((colT >> 4) * d.st_nt + (ct >> 5)) * 64 + ((colT >> 3) & 1) * 32 + (ct & 31)) * 8;
                store16_sc1(wc, (int)(o * (int64_t)sizeof(T)), tf[0]);
                if constexpr (NP == 2) store16_sc1(wc, (int)((o + g.wpl) * (int64_t)sizeof(T)), tf[1]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the tile is free for the wave's next task
    }
}

}  // namespace fused
}  // namespace dvae
