// M-step of the MCEM loop (EM.M_step, packages/models/mcem.py:91-153), H / g / cost part with the sample variances held in registers.
// Own translation unit: built with -fno-slp-vectorize (disentangled-vae_amd/build.py).  Under plain -O3 hipcc pairs the 170 register-resident
// values of a thread into 64-bit tuples for packed fp32 arithmetic (v_pk_fma_f32), defines the halves of a tuple at different times and
// spills 300 - 500 registers around the loads; the packed forms are no faster on gfx950 either (MI355X_MICROARCH.md, cycle constants).
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "fused_tiles.hpp"
#include "mcem_types.hpp"

namespace dvae {
namespace fused {

constexpr int KMAX = 16;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for_j(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_j<I + 1, N>(f); }
}

// 1 / x for the positive, normal variances of the M-step: the hardware reciprocal (1 ulp).  The IEEE division sequence (v_div_scale x 2,
// v_rcp, four v_fma, v_div_fmas, v_div_fixup) and even one Newton step on top of v_rcp_f32 tip hipcc's allocation of this kernel from 20
// to 150 spilled registers (170 of a thread's 256 hold the sample variances); sums of ten such terms per bin are compared with the
// reference at 1e-5 (tests/test_gpu_mcem.py), three orders above the reciprocal's error.
__device__ __forceinline__ float rcp_pos(float x) { return __builtin_amdgcn_rcpf(x); }

// Geometry of the frames kernel: FT16 threads = 16 frames x FG bin groups, a thread owns bins grp + FG j (j < FJ16) of one frame.
// 512 threads / 32 groups / 17 bins (256 registers per thread).  1024 threads / 9 bins (MSTEP_FT=1024, round 5) leaves 128 registers for
// 90 variances + the H column + 20 accumulators and spills 41 of them: slower.
#ifndef MSTEP_FT
#define MSTEP_FT 512
#endif
#ifndef MSTEP_FLY
#define MSTEP_FLY 2      // reciprocals in flight per thread inside a bin (more: more temporaries live, see the remark at the loops)
#endif
// Round 5: FR = 8 (and 4: 128 bin groups, 5 bins per thread) frames per workgroup for short frame axes (one utterance of 300 frames = 20 workgroups of 16 frames on 256 CUs: the kernel is
// per-workgroup latency there): 64 bin groups, 9 bins per thread -- half the loads and half the arithmetic per thread, twice the workgroups.
constexpr int FT16 = MSTEP_FT, FW = FT16 / 64;
template <int FR> struct MsGeo { static constexpr int FG = FT16 / FR, FJ = (513 + FG - 1) / FG; };
template <int RR, int K, int FR = 16>
__global__ __launch_bounds__(FT16) void mstep_frames_reg_kernel(const float* __restrict__ X2, const float* __restrict__ Vs, int R, int64_t N, int,
                                                                const float* __restrict__ Wun_all, float* __restrict__ H, float* __restrict__ g,
                                                                float* __restrict__ Vb, float* __restrict__ norms_out, double* __restrict__ partial,
                                                                const int* __restrict__ seg_start, const int* __restrict__ seg_count,
                                                                const int* __restrict__ tile_seg, float* __restrict__ Wout) {
    constexpr int FG = MsGeo<FR>::FG, FJ16 = MsGeo<FR>::FJ;         // (FJ16: bins per thread, whatever FR)
    extern __shared__ __attribute__((aligned(16))) float lw[];     // Wun [513][K], then wave partials [FW][2K][FR], then sums [2K][FR]
    float* wpart = lw + (XD * K + 3) / 4 * 4;
    float* sums = wpart + FW * 2 * K * FR;
    float* x2s = sums + 2 * K * FR;                                  // [j][thread]: X2 of this thread's 17 bins (read in each of the three passes)
    float* vlast = x2s + FJ16 * FT16;                                // [r][thread]: the sample variances of the LAST bin row (bin 512: one real bin, group 0 only)
    __shared__ float nrm[KMAX];
    __shared__ double redc[FW];
    const int tid = threadIdx.x, fr = tid & (FR - 1), grp = tid / FR, lane = tid & 63, wave = tid >> 6;
    // A workgroup's 16 frames are HALF of every 128-byte line of the (sample, bin) rows of Vs / X2 / Vb; the other half belongs to the
    // workgroup of the neighbouring 16 frames.  Workgroup i runs on XCD i % 8 (speed only, never correctness), so consecutive frame
    // blocks would pull every line into two L2s (twice the HBM traffic of the 164 MB the sample variances of 25 utterances take).
    // Remap: physical workgroups w and w + 8 of a group of 16 -- same XCD, dispatched together -- take the two halves.
#ifndef MSTEP_XCD_PAIRS
#define MSTEP_XCD_PAIRS 1
#endif
    int bid = blockIdx.x;
    if (FR == 16 && MSTEP_XCD_PAIRS && bid < ((int)gridDim.x & ~15)) { const int w = bid & 15; bid = (bid & ~15) + ((w & 7) << 1) + (w >> 3); }
    const int u = tile_seg ? tile_seg[bid / (32 / FR)] : 0;        // the segment table is per 32 frames
    const int64_t n0 = (int64_t)bid * FR;
    const int64_t nend = seg_start ? (int64_t)seg_start[u] + seg_count[u] : N;
    const bool live = n0 + fr < nend;
    const bool any_live = n0 < nend;                               // (the second half of a segment's last 32 frames may be all padding)
    const int64_t n = live ? n0 + fr : (any_live ? nend - 1 : n0); // padding frames read a valid column
    const int64_t FN = (int64_t)XD * N;
    const float* Wun = Wun_all + (int64_t)u * XD * K;
    // Vs of this thread's (bin, frame) pairs: requested first
    // Buffer accesses throughout: ONE per-lane byte offset (bin grp, frame n) for Vs, X2 and Vb, everything else -- the sample r, the bin
    // step 32 j -- in the scalar offset.  With plain 64-bit addresses hipcc computes the 170 + 17 + 17 of them ahead of the loads, keeps
    // them live through the three passes and spills 500 registers.  The host takes this kernel only while R F N floats stay below 2 GB.
    // The scalar offset is NOT part of the descriptor's range check on gfx9 (only the per-lane offset is compared with num_records), so
    // every access must be in range by construction: the one bin row past 512 that exists only for grp 0 (j = 16) is read by the other
    // groups at grp 0's own offset (voff16: a valid element whose value they never use).
    // registers: the first FJ16 - 1 bin rows (160 values); the last row -- bin 512, which only group 0 owns -- waits in thread-private LDS
    // slots like X2: ten registers less at the peak of pass 1, which is what hipcc was short of (it spilled 12 - 13 of the variances)
    float vs[FJ16 - 1][RR];
    auto vsv = [&](auto jc, int r) __attribute__((always_inline)) -> float {
        constexpr int j = decltype(jc)::value;
        if constexpr (j == FJ16 - 1) return vlast[r * FT16 + tid]; else return vs[j][r];
    };
    const int voff = (int)(((int64_t)grp * N + n) * 4);
    const int voff16 = (int)(n * 4);                               // bin 512 + 0: the only row of j = 16
    const unsigned jstep = (unsigned)(FG * N * 4);
    const __amdgpu_buffer_rsrc_t rs_vs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vs), 0, (int)((int64_t)R * FN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X2), 0, (int)(FN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_vb = __builtin_amdgcn_make_buffer_rsrc(Vb, 0, (int)(FN * 4), 0x00020000);
    // Request order = arrival order (vector-memory results return in issue order): first the small operands every pass needs -- the un-normalised
    // W for the workgroup's LDS copy, X2 of the thread's bins (parked in thread-private LDS slots, read back in each pass), the frame's H column
    // and gain -- then the 170 sample variances, bin by bin; the small operands go to LDS once a third of the variances is requested (they have
    // landed by then, and their registers are free again for the rest of the requests).  The workgroup barrier behind the W copy orders LDS
    // only (s_waitcnt lgkmcnt): __syncthreads() would drain vmcnt, i.e. every pass would start only after the LAST variance had landed.
    // Round 5 measured that serialisation (tools/r05/mstep_ablate.sh: loads alone 40 us + passes alone 61 us = the kernel's 95 us at 25
    // utterances) and, in the ISA, what made it worse: with the variances requested FIRST and W / X2 / H behind them, hipcc ran out of the
    // 256 registers at the end of the request stream and spilled the last 13 variances one by one as  buffer_load -> s_waitcnt vmcnt(0) ->
    // scratch_store : thirteen exposed memory round trips per workgroup.
    auto x2_at = [&](int j) __attribute__((always_inline)) { return x2s[j * FT16 + tid]; };
    constexpr int WCH = (XD * K + FT16 - 1) / FT16;
    constexpr int JA = (FJ16 + 2) / 3;                              // bins requested ahead of the LDS writes
    float wtmp[WCH], x2t[FJ16], hk[K], vlt[RR];
#pragma unroll
    for (int c = 0; c < WCH; ++c) { const int i = tid + c * FT16; wtmp[c] = Wun[i < XD * K ? i : XD * K - 1]; }
#pragma unroll
    for (int j = 0; j < FJ16; ++j)
        x2t[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x2, j == FJ16 - 1 ? voff16 : voff, (int)(jstep * (unsigned)j), 0));
#pragma unroll
    for (int r = 0; r < RR; ++r)
        vlt[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_vs, voff16, (int)(jstep * (unsigned)(FJ16 - 1) + (unsigned)((int64_t)(r < R ? r : 0) * FN * 4)), 0));
#pragma unroll
    for (int k = 0; k < K; ++k) hk[k] = H[(int64_t)k * N + n];
    const float gn = g[n];
    __builtin_amdgcn_sched_barrier(0);
    auto request = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const unsigned soff = jstep * (unsigned)j + (unsigned)((int64_t)(r < R ? r : 0) * FN * 4);
#if defined(MSTEP_ABL) && MSTEP_ABL == 2      // timing ablation: no reads of the sample variances
            vs[j][r] = 1.0f + 0.001f * (float)(j + r + tid); (void)soff;
#else
            vs[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_vs, voff, (int)soff, 0));
#endif
        }
    };
    static_for_j<0, JA>(request);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < WCH; ++c) { const int i = tid + c * FT16; if (i < XD * K) lw[i] = wtmp[c]; }
#pragma unroll
    for (int j = 0; j < FJ16; ++j) x2s[j * FT16 + tid] = x2t[j];
#pragma unroll
    for (int r = 0; r < RR; ++r) vlast[r * FT16 + tid] = vlt[r];
    __builtin_amdgcn_sched_barrier(0);
    static_for_j<JA, FJ16 - 1>(request);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < 32 * K) {                                            // column norms of the un-normalised W (mcem.py:130): 32 threads per column
        // (these waves also run pass 1 and everyone meets at its barrier: sixteen threads per column walked 33 dependent LDS reads)
        const int k = tid >> 5, q = tid & 31;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < (XD + 31) / 32; ++i) { const int f = q + 32 * i; s += f < XD ? fabsf(lw[f * K + k]) : 0.f; }
        s += __shfl_xor(s, 16, 32); s += __shfl_xor(s, 8, 32); s += __shfl_xor(s, 4, 32); s += __shfl_xor(s, 2, 32); s += __shfl_xor(s, 1, 32);
        if (q == 0) { nrm[k] = s; if (n0 == (seg_start ? seg_start[u] : 0)) norms_out[u * KMAX + k] = s; }
    }

    // sum over the FG bin groups of `cnt` per-thread values: wave shuffles (4 groups per wave), LDS (FW waves)
    auto group_sums = [&](auto& v, int cnt) {
        constexpr int M = (int)(sizeof(v) / sizeof(float));
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i < cnt) {
                float a = v[i];
                if constexpr (FR == 4) a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x124, 0xf, 0xf, false));   // lanes 4 apart (row_ror:4)
                if constexpr (FR <= 8) a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x128, 0xf, 0xf, false));   // lanes 8 apart (row_ror:8)
                a = xsum16(a);                                      // (v_permlane16/32_swap: mcem_types.hpp; the operands of ds_bpermute's additions)
                a = xsum32(a);
                if (lane < FR) wpart[(wave * cnt + i) * FR + fr] = a;
            }
        }
        __syncthreads();
        if (grp < cnt) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < FW; ++w) a += wpart[(w * cnt + grp) * FR + fr];
            sums[grp * FR + fr] = a;
        }
        __syncthreads();
    };

#if defined(MSTEP_ABL) && MSTEP_ABL == 1          // timing ablation: loads only (no passes)
    {
        float t = gn;
        static_for_j<0, FJ16>([&](auto jc) { t += x2_at(decltype(jc)::value);
#pragma unroll
            for (int r = 0; r < RR; ++r) t += vsv(jc, r); });
#pragma unroll
        for (int k = 0; k < K; ++k) t += hk[k];
        if (t == 123.456f) g[n] = t;
        return;
    }
#endif
    // ---- H update (mcem.py:118-123) with Vb = Wun H ----
    float acc[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; ++k) acc[k] = 0.f;
    static_for_j<0, FJ16>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int f = grp + FG * j;
        if (f < XD) {
            float vb = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) vb = fmaf(lw[f * K + k], hk[k], vb);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int r = 0; r < RR; ++r) { if (r < R) { const float inv = rcp_pos(fmaf(gn, vsv(jc, r), vb)); a1 += inv; a2 += inv * inv; } if ((r % MSTEP_FLY) == MSTEP_FLY - 1) __builtin_amdgcn_sched_barrier(0); }   // (two divisions in flight: ten interleaved IEEE sequences hold 80 temporaries)
            const float p2 = x2_at(j) * a2;
#pragma unroll
            for (int k = 0; k < K; ++k) { const float w = lw[f * K + k]; acc[2 * k] = fmaf(w, p2, acc[2 * k]); acc[2 * k + 1] = fmaf(w, a1, acc[2 * k + 1]); }
        }
        __builtin_amdgcn_sched_barrier(0);        // one bin at a time: hoisting every bin's W row above the loop costs 110 more registers
    });
    group_sums(acc, 2 * K);
#pragma unroll
    for (int k = 0; k < K; ++k) hk[k] = hk[k] * sqrtf(sums[(2 * k) * FR + fr] / sums[(2 * k + 1) * FR + fr]);
    __syncthreads();

    // ---- new Vb = Wun Hnew (mcem.py:126); g update (mcem.py:137-143) ----
    float gv[2] = {0.f, 0.f};
    static_for_j<0, FJ16>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int f = grp + FG * j;
        if (f < XD) {
            float vb = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) vb = fmaf(lw[f * K + k], hk[k], vb);
            if (live) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vb), rs_vb, voff, (int)(jstep * (unsigned)j), 0);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < RR; ++r) { if (r < R) { const float v = vsv(jc, r); const float inv = rcp_pos(fmaf(gn, v, vb)); s1 = fmaf(v, inv, s1); s2 = fmaf(v, inv * inv, s2); } if ((r % MSTEP_FLY) == MSTEP_FLY - 1) __builtin_amdgcn_sched_barrier(0); }
            gv[0] = fmaf(x2_at(j), s2, gv[0]); gv[1] += s1;
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    group_sums(gv, 2);
    const float gnew = gn * sqrtf(sums[fr] / sums[FR + fr]);

    // ---- cost (mcem.py:69-71) with the updated g; H is stored normalised (mcem.py:134).  log through the hardware log2 (~1 ulp of log2) ----
    double c = 0.0;
    static_for_j<0, FJ16>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int f = grp + FG * j;
        if (f < XD) {
            float vb = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) vb = fmaf(lw[f * K + k], hk[k], vb);
            const float x2 = x2_at(j);
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < RR; ++r) { if (r < R) { const float vx = fmaf(gnew, vsv(jc, r), vb); s += __builtin_amdgcn_logf(vx) * 0.693147180559945309f + x2 * rcp_pos(vx); } if ((r % MSTEP_FLY) == MSTEP_FLY - 1) __builtin_amdgcn_sched_barrier(0); }
            c += (double)s;
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if (!live) c = 0.0;
    c = wave_sum(c);
    if (lane == 0) redc[wave] = c;
    if (grp == 0 && live) {
        g[n] = gnew;
#pragma unroll
        for (int k = 0; k < K; ++k) H[(int64_t)k * N + n] = hk[k] * nrm[k];
    }
    if (Wout != nullptr && any_live) {
        // W = Wun / norm (mcem.py:132) here instead of in mstep_finish_kernel (dvae_mcem_em_iteration_lazy): every workgroup holds Wun and
        // its column norms; the utterance's workgroups share the 513 rows out (the operands of the finish kernel's division: the same bits)
        const int64_t seg0 = seg_start ? (int64_t)seg_start[u] : 0;
        const int lt = (int)((n0 - seg0) / FR), ntl = (int)((nend - seg0 + FR - 1) / FR);
        float* const Wu = Wout + (int64_t)u * XD * K;
        for (int i = tid; i < XD * K; i += FT16) {
            const int f = i / K;
            if (f % ntl == lt) Wu[i] = lw[i] / nrm[i - f * K];
        }
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < FW; ++w) t += redc[w];
        partial[bid] = t;
    }
}


// W update (mcem.py:107-111): num[k] = sum_n X2 sum_r Vx^-2 H[k,n], den[k] = sum_n sum_r Vx^-1 H[k,n], Wun = W sqrt(num / den), Vx = g Vs + Vb.
// One WAVE per (bin f, utterance u), four bins per workgroup, 64 frames per step -- as mstep_w_kernel (mcem.hip), with what that kernel
// lacks at batch scale (116 us for 164 MB, round 4): every load of a step (R sample variances, Vb, X2, g, K rows of H) is requested before
// the first is used and the NEXT step's loads are in flight under this step's arithmetic (two register sets); buffer addressing with one
// per-lane offset per matrix shape; the hardware reciprocal (rcp_pos above) instead of ten IEEE divisions per frame.
template <int RR, int K>
__global__ __launch_bounds__(256) void mstep_w_reg_kernel(const float* __restrict__ X2, const float* __restrict__ Vs, int R, int64_t N,
                                                          const float* __restrict__ W, const float* __restrict__ H, const float* __restrict__ g,
                                                          const float* __restrict__ Vb, float* __restrict__ Wun,
                                                          const int* __restrict__ seg_start, const int* __restrict__ seg_count,
                                                          const double* __restrict__ partial_prev, float* __restrict__ cost_prev, int tile_frames) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 4 + wave, u = blockIdx.y;
    const int64_t nbeg = seg_start ? seg_start[u] : 0, cnt = seg_count ? seg_count[u] : N, nend = nbeg + cnt;
    if (cost_prev != nullptr && blockIdx.x == 0) {
        // the cost of the PREVIOUS iteration from the partial sums its frames kernel left (the frames kernel of this iteration overwrites
        // them behind this launch): what mstep_finish_kernel does, without its launch (dvae_mcem_em_iteration_lazy)
        __shared__ double redc[4];
        const int t0 = (int)(nbeg / tile_frames), t1 = (int)((nbeg + cnt + tile_frames - 1) / tile_frames);
        mstep_cost_of_partials(partial_prev, t0, t1, (double)R * XD * (double)cnt, cost_prev + u, redc);
    }
    if (f >= XD) return;
    const int64_t FN = (int64_t)XD * N;
    const __amdgpu_buffer_rsrc_t rs_vs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vs), 0, (int)((int64_t)R * FN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X2), 0, (int)(FN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_vb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vb), 0, (int)(FN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(H), 0, (int)((int64_t)K * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, (int)(N * 4), 0x00020000);
    const unsigned fnb = (unsigned)(FN * 4), nb = (unsigned)(N * 4);
    struct Set { float vs[RR], vb, x2, gn, hk[K], m; };
    auto ldf = [](__amdgpu_buffer_rsrc_t rs, int voff, unsigned soff) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)soff, 0));
    };
    auto load = [&](Set& q, int64_t c) __attribute__((always_inline)) {
        const int64_t nn = nbeg + 64 * c + lane;
        const bool lv = nn < nend;
        const int64_t n = lv ? nn : nend - 1;                            // padding lanes read a valid column with weight 0
        const int vfn = (int)(((int64_t)f * N + n) * 4), vn = (int)(n * 4);
#pragma unroll
        for (int r = 0; r < RR; ++r) q.vs[r] = ldf(rs_vs, vfn, (unsigned)(r < R ? r : 0) * fnb);
        q.vb = ldf(rs_vb, vfn, 0); q.x2 = ldf(rs_x2, vfn, 0); q.gn = ldf(rs_g, vn, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) q.hk[k] = ldf(rs_h, vn, (unsigned)k * nb);
        q.m = lv ? 1.f : 0.f;
    };
    float num[K], den[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { num[k] = 0.f; den[k] = 0.f; }
    auto compute = [&](const Set& q) __attribute__((always_inline)) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            if (r < R) {
                const float inv = rcp_pos(fmaf(q.gn, q.vs[r], q.vb));
                a1 += inv; a2 = fmaf(inv, inv, a2);
            }
        }
        const float p2 = q.x2 * a2 * q.m;
        a1 *= q.m;
#pragma unroll
        for (int k = 0; k < K; ++k) { num[k] = fmaf(p2, q.hk[k], num[k]); den[k] = fmaf(a1, q.hk[k], den[k]); }
    };
    const int64_t nch = (cnt + 63) / 64;
    constexpr int NALL = 5;
    if (nch > 0 && nch <= NALL) {
        // one utterance of up to 320 frames: every step's operands requested at once, straight-line (steps past the end load the last
        // valid column with weight 0 -- load() clamps -- and add zeros: the sums of the loop below, bit for bit)
        Set S[NALL];
#pragma unroll
        for (int c = 0; c < NALL; ++c) load(S[c], c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NALL; ++c) compute(S[c]);
    } else {
        Set A, B;
        if (nch > 0) load(A, 0);
        for (int64_t c = 0; c < nch; c += 2) {
            if (c + 1 < nch) load(B, c + 1);
            compute(A);
            if (c + 1 < nch) {
                if (c + 2 < nch) load(A, c + 2);
                compute(B);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float a = wave_sum(num[k]), b = wave_sum(den[k]);
        if (lane == k) {
            const int64_t o = ((int64_t)u * XD + f) * K + k;
            Wun[o] = W[o] * sqrtf(a / b);
        }
    }
}


namespace mstep {
// frames per workgroup of the frames kernel = frames per cost partial: 4 / 8 while the 4- / 8-frame workgroups fit the chip in one round
// (DVAE_MSTEP_FRAMES=4 / 8 / 16 forces)
int frames_per_workgroup(int64_t N) {
    const char* e = getenv("DVAE_MSTEP_FRAMES");
    const int forced = e ? atoi(e) : 0;
    if (forced == 4 || forced == 8 || forced == 16) return forced;
    return (N + 3) / 4 <= 256 ? 4 : ((N + 7) / 8 <= 256 ? 8 : 16);
}
int launch_frames_reg(const float* X2, const float* Vs, int R, int64_t N, int K, const float* Wun, float* H, float* g, float* Vb,
                      float* norms, double* partial, const int* seg_start, const int* seg_count, const int* tile_seg, float* Wout, hipStream_t s) {
    const int fr = frames_per_workgroup(N);
    const int nt = (int)((N + fr - 1) / fr);
    const int FJ = fr == 4 ? MsGeo<4>::FJ : (fr == 8 ? MsGeo<8>::FJ : MsGeo<16>::FJ);
    const size_t lds = ((size_t)(XD * K + 3) / 4 * 4 + FW * 2 * K * fr + 2 * K * fr + FJ * FT16 + 10 * FT16) * sizeof(float);
    static bool attr_done16 = false;
    if (!attr_done16) {
        const size_t lds_max = ((size_t)(XD * KMAX + 3) / 4 * 4 + FW * 2 * KMAX * 16 + 2 * KMAX * 16 + MsGeo<16>::FJ * FT16 + 10 * FT16) * sizeof(float);
        hipError_t e = hipFuncSetAttribute((const void*)mstep_frames_reg_kernel<10, 10, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mstep_frames_reg_kernel<10, 10, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mstep_frames_reg_kernel<10, 10, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(mstep_frames_reg_kernel, %zu B LDS): %s", lds_max, hipGetErrorString(e)); return (int)e; }
        attr_done16 = true;
    }
    if (fr == 4) hipLaunchKernelGGL((mstep_frames_reg_kernel<10, 10, 4>), dim3(nt), dim3(FT16), lds, s, X2, Vs, R, N, K, Wun, H, g, Vb, norms, partial, seg_start, seg_count, tile_seg, Wout);
    else if (fr == 8) hipLaunchKernelGGL((mstep_frames_reg_kernel<10, 10, 8>), dim3(nt), dim3(FT16), lds, s, X2, Vs, R, N, K, Wun, H, g, Vb, norms, partial, seg_start, seg_count, tile_seg, Wout);
    else hipLaunchKernelGGL((mstep_frames_reg_kernel<10, 10, 16>), dim3(nt), dim3(FT16), lds, s, X2, Vs, R, N, K, Wun, H, g, Vb, norms, partial, seg_start, seg_count, tile_seg, Wout);
    DVAE_LAUNCH_OK("mstep_frames_reg_kernel");
    return 0;
}
int launch_w_reg(const float* X2, const float* Vs, int R, int64_t N, int U, const float* W, const float* H, const float* g, const float* Vb,
                 float* Wun, const int* seg_start, const int* seg_count, const double* partial_prev, float* cost_prev, hipStream_t s) {
    hipLaunchKernelGGL((mstep_w_reg_kernel<10, 10>), dim3((XD + 3) / 4, U), dim3(256), 0, s, X2, Vs, R, N, W, H, g, Vb, Wun, seg_start, seg_count,
                       partial_prev, cost_prev, frames_per_workgroup(N));
    DVAE_LAUNCH_OK("mstep_w_reg_kernel");
    return 0;
}
}  // namespace mstep

}  // namespace fused
}  // namespace dvae
