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
    // Every product and sum is its own correctly rounded operation -- contraction into fused multiply-adds is switched OFF for this body:
    // the three callers inline it into different surroundings, and a contraction the compiler chose in one and not in another differs in the
    // last bit (round 4: the in-kernel update and apply_kernel agreed on the first step, where m = v = 0, and differed by one ulp in a
    // quarter of the elements of m on the second; __fmul_rn / __fadd_rn are plain operators in HIP's headers and do not prevent it).
#pragma clang fp contract(off)
    gi = gi * g.gscale;
    const float dm = gi - m_old;
    const float wm = g.one_minus_b1 * dm;
    mi = m_old + wm;
    const float g2 = gi * gi;
    const float va = v_old * g.b2;
    const float vb = g.one_minus_b2 * g2;
    vi = va + vb;
    const float denom = sqrtf(vi) / g.bc2_sqrt + g.eps;
    const float q = mi / denom;
    const float st = g.step_size * q;
    pi = pi - st;
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
    // position of (rr, cc) in a fragment-major copy with `nt` row tiles: 32-row tiles / 16-deep k-steps, or (kind16) 16-row tiles / 32-deep
    auto pos = [&](int64_t off, int nt, int rr, int cc, int ld) __attribute__((always_inline)) -> int64_t {
        if (!WFRAG) return off + (int64_t)rr * ld + cc;
        if (d.kind16) return off + ((int64_t)((cc >> 5) * nt + (rr >> 4)) * 64 + ((cc & 31) >> 3) * 16 + (rr & 15)) * 8 + (cc & 7);
        return off + ((int64_t)((cc / KS) * nt + (rr >> 5)) * 64 + ((cc % KS) / E) * 32 + (rr & 31)) * E + cc % E;
    };
    auto heads_row = [](int k, int map) { return 16 * (k >> 3) + 4 * ((k & 7) >> 1) + (k & 1) + (map == 2 ? 2 : 0); };
    if (d.sf_off >= 0) {
        const int rr = d.rowmap ? heads_row(r, d.rowmap) : r + d.sf_roff, cc = c < d.sf_split ? c : c + d.sf_gap;
        const int64_t o = pos(d.sf_off, d.sf_nt, rr, cc, d.sf_ld);
        if constexpr (NP == 2 && sizeof(T) == 2) {
            if (c < d.sf_f16_cols) {                               // the x block of layer 1: split fp16 of W * 2^6 (struct X16)
                const float ws = pi * X16::WS;
                const T ph = X16::hi<T>(ws);
                wc[o] = ph;
                wc[o + g.wpl] = X16::hi<T>(ws - X16::val(ph));
            } else { const T ph = (T)pi; wc[o] = ph; wc[o + g.wpl] = (T)(pi - (float)ph); }
        } else {
            const T ph = (T)pi;
            wc[o] = ph;
            if constexpr (NP == 2) wc[o + g.wpl] = (T)(pi - (float)ph);
        }
    }
    if (d.st_off >= 0 && c < d.st_cmax) {
        const int rr = d.trowmap ? heads_row(c, d.trowmap) : c, cc = r + d.st_roff;
        const int64_t o = pos(d.st_off, d.st_nt, rr, cc, d.st_ld);
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
template <typename T, int NP> struct DeferLds { static constexpr int wave_elems = NP * 8 * DEFER_LDT; static constexpr size_t bytes = (size_t)4 * wave_elems * sizeof(T); };

// up to four consecutive floats at element `idx` of a wave-uniform array: one 16-byte store (rows of odd length are only 4-byte aligned:
// gfx950 takes 16-byte accesses at dword alignment), single floats at the ragged end of a row; SC1: write-through
template <bool SC1>
__device__ __forceinline__ void store_f4(float* base_uniform, int64_t idx, const f32x4& v, int nv) {
    if (nv >= 4) {
        if constexpr (SC1) {
            typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rs, (int)(idx * 4), 0, 16);
        } else reinterpret_cast<F4U*>(base_uniform + idx)->v = v;
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nv) {
                if constexpr (SC1) __hip_atomic_store(base_uniform + idx + j, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else base_uniform[idx + j] = v[j];
            }
    }
}

// One UNIT of a task on one wave (EXEC all ones): rows r0 + 8 u .. + 7 of the task's 32 x 32 tile (flat tasks: elements 256 u ..), u = 0 .. 3.
// Element mapping: lane = (rr = lane >> 3, p = lane & 7) = row rr of the unit, columns 4 p .. 4 p + 3 -- a wave instruction covers 8 rows x
// 128 contiguous bytes, so every load and store of parameters, moments and slabs moves whole lines (a first version gave a lane 16
// consecutive elements of one row: 64 lines per instruction, each fetched four times, and lone 4-byte stores: 11 us of stores and 7 us of
// loads).  Units, not tasks, are dealt out to the waves of the grid: 4 x 323 units over 1024 waves is five or six per CU on every CU,
// where whole tiles gave a quarter of the CUs two tiles and the rest one (the arrival wait ends with the slowest CU).
// `tile` = this wave's LDS scratch ([planes][8][DEFER_LDT]).
// COH: the consumers are other workgroups of THIS launch (the deferred form): parameters and copies leave as write-through (sc1) stores;
// false: plain stores (apply_units_kernel: its own launch, the kernel boundary publishes them)
template <typename T, int NP, bool COH = true>
__device__ __forceinline__ void defer_unit(const ApplyArgs& g, const DeferTask tk, int u, T* tile, int lane, int diag = 0) {
    static_assert(sizeof(T) == 2, "deferred step: bf16 copies");
    typedef T Frag8 __attribute__((ext_vector_type(8)));
    typedef T Frag4 __attribute__((ext_vector_type(4)));
    const TensorDesc d = g.tensors[tk.tensor];                   // wave-uniform
    const bool flat = tk.kind == 1;
    const int p = lane & 7, rr = lane >> 3;
    const int r = flat ? 0 : tk.r0 + 8 * u + rr;
    const int c = flat ? tk.cbeg + 4 * (lane + 64 * u) : tk.cbeg + 4 * p;
    int nv = (flat || r < d.rows) ? tk.cend - c : 0;
    nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
    const int64_t b4 = nv > 0 ? d.off + (flat ? (int64_t)c : (int64_t)r * d.cols + c) : d.off;       // idle lanes read the tensor's first elements
    // every load first (rows of odd length are only 4-byte aligned: gfx950 takes 16-byte accesses at dword alignment)
    const f32x4 pv = reinterpret_cast<const F4U*>(g.p + b4)->v;
    const f32x4 mv = reinterpret_cast<const F4U*>(g.m + b4)->v;
    const f32x4 vv = reinterpret_cast<const F4U*>(g.v + b4)->v;
    constexpr int NSM = 12;
    f32x4 sv[NSM];
#pragma unroll
    for (int k = 0; k < NSM; ++k) sv[k] = (diag & 16) ? pv : reinterpret_cast<const F4U*>(g.slabs + (int64_t)(k < g.nslabs ? k : 0) * g.slab_stride + b4)->v;
    __builtin_amdgcn_sched_barrier(0);
    float pn[4];
    f32x4 po, mo, vo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float gi = sv[0][j];
#pragma unroll
        for (int k = 1; k < NSM; ++k) if (k < g.nslabs) gi += sv[k][j];                   // slab order: apply_kernel's additions
        float pi = pv[j], mi, vi;
        adam_element(g, pi, mv[j], vv[j], gi, mi, vi);
        po[j] = pi; mo[j] = mi; vo[j] = vi;
        pn[j] = j < nv ? pi : 0.f;
    }
    if (nv > 0 && !(diag & 8)) {
        store_f4<COH>(g.p, b4, po, nv);                                                   // sc1: biases are read by this launch
        store_f4<false>(g.m, b4, mo, nv);
        store_f4<false>(g.v, b4, vo, nv);
    }
    if (flat || (diag & 32)) return;                                 // wave-uniform
    T* const wc = (T*)g.wcopy;
    Frag4 fh, fl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const T ph = (T)pn[j];
        fh[j] = ph;
        fl[j] = (T)(pn[j] - (float)ph);
    }
    Frag4 fwh = fh, fwl = fl;                                        // forward-copy planes: split fp16 of W * 2^6 in the x block of layer 1
    if (NP == 2 && tk.cbeg < d.sf_f16_cols) {                        // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ws = pn[j] * X16::WS;
            fwh[j] = X16::hi<T>(ws);
            fwl[j] = X16::hi<T>(ws - X16::val(fwh[j]));
        }
    }
    if (d.sf_off >= 0 && nv > 0) {
        // forward copy: columns 4 p .. 4 p + 3 of the tile are elements 4 (p & 1) .. of fragment half (p >> 1) & 1 of k-step cc / 16:
        // 8-byte stores, two lanes per 16-byte fragment, 8 rows per 128 contiguous bytes
        const int cc = (tk.cbeg < d.sf_split ? tk.cbeg : tk.cbeg + d.sf_gap) + 4 * p;   // (a tile never straddles the split: cbeg = start of a column block + a multiple of 32)
        const int rr2 = r + d.sf_roff;
        const int64_t o = d.sf_off + ((int64_t)((cc >> 4) * d.sf_nt + (rr2 >> 5)) * 64 + ((cc >> 3) & 1) * 32 + (rr2 & 31)) * 8 + (cc & 4);
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wc, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, fwh), rs, (int)(o * (int64_t)sizeof(T)), 0, COH ? 16 : 0);
        if constexpr (NP == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, fwl), rs, (int)((o + g.wpl) * (int64_t)sizeof(T)), 0, COH ? 16 : 0);
    }
    if (d.st_off >= 0 && tk.cbeg < d.st_cmax) {                      // wave-uniform
        // transposed copy: the unit's [8 r][32 c] in LDS, read back as the 8 consecutive r of one c (ds_read_b64_tr_b16): lanes 0 .. 31 the
        // hi plane, lanes 32 .. 63 the lo plane
        *reinterpret_cast<Frag4*>(tile + rr * DEFER_LDT + 4 * p) = fh;
        if constexpr (NP == 2) *reinterpret_cast<Frag4*>(tile + 8 * DEFER_LDT + rr * DEFER_LDT + 4 * p) = fl;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave's own writes (LDS operations of one wave complete in order)
        const int h = lane >> 5, l31 = lane & 31, i16 = l31 & 15, q = i16 >> 2, pp = i16 & 3, cg = l31 >> 4;
        const T* bp = tile + (NP == 2 ? h : 0) * 8 * DEFER_LDT + q * DEFER_LDT + 16 * cg + 4 * pp;
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x4 r0v = lds_tr16(bp), r1v = lds_tr16(bp + 4 * DEFER_LDT);
        const s16x8 raw = {r0v[0], r0v[1], r0v[2], r0v[3], r1v[0], r1v[1], r1v[2], r1v[3]};
        const Frag8 tf = __builtin_bit_cast(Frag8, raw);
        const int ct = tk.cbeg + l31;                                 // this lane's row of the transposed matrix
        const int colT = tk.r0 + 8 * u + d.st_roff;
        if (ct < tk.cend && ct < d.st_cmax && tk.r0 + 8 * u < d.rows && (NP == 2 || h == 0)) {
            const int64_t o = d.st_off + ((int64_t)((colT >> 4) * d.st_nt + (ct >> 5)) * 64 + ((colT >> 3) & 1) * 32 + (ct & 31)) * 8 + (NP == 2 && h ? g.wpl : 0);
            if constexpr (COH) store16_sc1(wc, (int)(o * (int64_t)sizeof(T)), tf);
            else *reinterpret_cast<Frag8*>(wc + o) = tf;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the tile is free for the wave's next unit
    }
}

}  // namespace fused
}  // namespace dvae
