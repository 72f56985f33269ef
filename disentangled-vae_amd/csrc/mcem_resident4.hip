// The weight-stationary Metropolis-Hastings chain (mcem_resident.hip) under the EXACT-fp32 policy on FOUR-frame tiles: ONE utterance of 300
// frames -- what scripts/evaluate_ntcd_M2.py runs per process (packages/models/mcem.py:207-290), under the drop-in classes' default policy --
// becomes 75 workgroups instead of 19.
//
// Why this shape exists for fp32 only: every fp32 MFMA shape runs at the same 32 MAC per clock and SIMD (tools/r05/mfma16_chain_bench.hip:
// 64.0 / 32.0 / 8.2 clocks for 32x32x2 / 16x16x4 / 4x4x1), and the fp32 matrix instructions occupy the SIMD's own FMA lanes, so a chain step
// costs its matrix work + its epilogue + its serial section, all of it proportional to the frames of the tile except the last.  The narrowest
// shape, v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4 x 4 outer products per instruction -- makes the tile four frames: the output layer
// of a step is 256 MFMAs of 8 clocks per wave (2 100 clocks against 8 200 on sixteen frames).  (The bf16 shapes gain nothing this way: their
// 4 x 4 form has a quarter of the 16 x 16 form's rate.)
//
//   * lane = (block b = lane >> 2, j = lane & 3).  Per instruction and block: D_b[i][j] += A_b[i] * B_b[j], A_b[i] in lane 4 b + i, B_b[j] in
//     lane 4 b + j, D_b[i][j] in register i of lane 4 b + j (tools/r05/mfma4x4_probe.hip).  B is the activation of FRAME j -- the same in all
//     sixteen blocks -- and block b holds four ROWS of the weight matrix: one instruction = 64 rows x 4 frames x one k.
//   * output layer: wave w owns bins 128 w .. 128 w + 127 as two 64-row groups; the lane's weights W[128 w + 64 R + 4 b + (lane & 3)][k], all 128 k,
//     are 256 registers (240 named as AGPRs by the MFMA statements, as in the other chain kernels: hazard table in mcem_resident.hip), gathered
//     once per launch from the 32-row fragment-major copy (16-byte chunks = four consecutive k of a row); bin 512 is the fp32 dot product
//     finished by wave 3;
//   * layers 1 and 2: wave w owns features 32 w .. 32 w + 31 = eight blocks; the other eight blocks take the second half of k (their B operand
//     is the activation at k + 64, their result is added across the lane halves with one v_permlane32_swap per register);
//   * chain state on wave 0: lane (b, j) holds latent b of frame j; sums over the sixteen blocks of a frame (prior, likelihood) are xor
//     butterflies -- row rotations (DPP) for lane distances 4 and 8, permlane swaps for 16 and 32 -- so every lane of a frame holds the same
//     bits and takes the same accept decision;
//   * a workgroup's tile: eight consecutive tiles (one 128-byte line of every (bin, N) row) go to workgroups of ONE XCD, so the 16-byte runs
//     they read and write meet in that XCD's L2.
// Same arithmetic per element as the other fp32 chains (hardware exp2 / log2 / rcp, per-lane float sums, double across lanes and waves); a
// frame's 513 terms are partitioned over lanes differently, so the kernels agree to rounding, not bit for bit.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "fused_tiles.hpp"
#include "mcem_types.hpp"
#include "../../include/dvae_mcem.h"

namespace dvae {
namespace fused {

namespace q4 {
constexpr int T4 = 4;                                                    // frames per tile
constexpr int LDH = HD + 4, LDZ = ZD + 4, LDC = HD + 4, LDY = NO + 4;
constexpr size_t O_HA = 0;
constexpr size_t O_HB = O_HA + (size_t)T4 * LDH * sizeof(float);
constexpr size_t O_ZB = O_HB + (size_t)T4 * LDH * sizeof(float);
constexpr size_t O_X2 = (O_ZB + (size_t)T4 * LDZ * sizeof(float) + 15) / 16 * 16;   // X2: [wave][row group][lane] x 4 bins
constexpr size_t O_VB = O_X2 + (size_t)4 * 2 * 64 * 4 * sizeof(float);
constexpr size_t O_C1 = O_VB + (size_t)4 * 2 * 64 * 4 * sizeof(float);              // [frame][LDC]: b3 + W3[:, 16:] y
constexpr size_t O_BIAS = O_C1 + (size_t)T4 * LDC * sizeof(float);
constexpr size_t O_W512 = O_BIAS + (size_t)(2 * HD + NO) * sizeof(float);
constexpr size_t O_P512 = O_W512 + (size_t)HD * sizeof(float);
constexpr size_t O_RED = O_P512 + (size_t)4 * T4 * sizeof(float);
constexpr size_t O_ACC = O_RED + (size_t)4 * T4 * sizeof(double);         // [frame]: the last step's accept decision, for the other waves
constexpr size_t LDS = O_ACC + (size_t)T4 * sizeof(int);
static_assert(O_X2 % 16 == 0 && O_C1 % 16 == 0 && O_BIAS % 16 == 0 && O_W512 % 16 == 0 && O_RED % 8 == 0, "4-frame chain: LDS layout");
static_assert((size_t)T4 * LDY * sizeof(float) <= O_C1 - O_X2, "the label image fits the X2 / Vb area");

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float exp_(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }
__device__ __forceinline__ float log_(float v) { return __builtin_amdgcn_logf(v) * 0.693147180559945309f; }
__device__ __forceinline__ float tanh_(float v) {
    const float e = __builtin_amdgcn_exp2f(v * 2.88539008177792681f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// value of lane ^ 4 / lane ^ 8 within the row of 16 (row rotations: lane l reads l - n mod 16)
__device__ __forceinline__ unsigned dpp_x4(unsigned u, bool bit2) {
    const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x124, 0xf, 0xf, false);      // row_ror:4  -> from l - 4
    const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x12c, 0xf, 0xf, false);      // row_ror:12 -> from l + 4
    return bit2 ? dn : up;
}
__device__ __forceinline__ unsigned dpp_x8(unsigned u) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, false); }
// x summed over the lanes that share (lane & 3) -- the sixteen blocks of a frame: a butterfly (lane ^ 4, ^ 8, ^ 16, ^ 32), so every lane of
// the class ends with the same bits; NB = 8: over the eight blocks of a 32-lane half only (lane ^ 4, ^ 8, ^ 16)
template <int NB>
__device__ __forceinline__ float bsum(float x, bool bit2) {
    x += __builtin_bit_cast(float, dpp_x4(__builtin_bit_cast(unsigned, x), bit2));
    x += __builtin_bit_cast(float, dpp_x8(__builtin_bit_cast(unsigned, x)));
    x = xsum16(x);
    if constexpr (NB == 16) x = xsum32(x);
    return x;
}
__device__ __forceinline__ double bsum16d(double x, bool bit2) {
    auto part = [&](double v, auto f) __attribute__((always_inline)) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = f((unsigned)u), hi = f((unsigned)(u >> 32));
        return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
    };
    x += part(x, [&](unsigned u) { return dpp_x4(u, bit2); });
    x += part(x, [&](unsigned u) { return dpp_x8(u); });
    x = xsum16(x);
    x = xsum32(x);
    return x;
}

// one MFMA of the output layer's chains, written out (the weight operand lives in an AGPR: see mcem_resident.hip); LAST carries the wait
// states in front of the first VALU read of the accumulator (12 = the 8-pass figure; this is a 2-pass instruction)
template <bool AG, bool LAST, bool FIRST = false>
__device__ __forceinline__ void mf4(f32x4_t& acc, float a, float b) {
    if constexpr (FIRST) {               // (two wait states in front of the chain's first read of the accumulator, should the bias ever reach it through a VALU copy)
        if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        else asm volatile("s_nop 1\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    } else if constexpr (LAST) {
        if constexpr (AG) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "a"(a), "v"(b));
        else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (AG) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}
template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
}  // namespace q4

template <int YP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mcem_resident4_kernel(const MhArgs g) {
    using namespace q4;
    constexpr int OB4 = HD, OB5 = 2 * HD;
    constexpr int KAG = 120;                                              // k < KAG of each row group: fragments in AGPRs (240 in all)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const Ha = reinterpret_cast<float*>(smem + O_HA);
    float* const Hb = reinterpret_cast<float*>(smem + O_HB);
    float* const Zb = reinterpret_cast<float*>(smem + O_ZB);
    f32x4_t* const X2s = reinterpret_cast<f32x4_t*>(smem + O_X2);
    f32x4_t* const Vbs = reinterpret_cast<f32x4_t*>(smem + O_VB);
    float* const c1s = reinterpret_cast<float*>(smem + O_C1);
    float* const Bias = reinterpret_cast<float*>(smem + O_BIAS);
    float* const w512s = reinterpret_cast<float*>(smem + O_W512);
    float* const p512 = reinterpret_cast<float*>(smem + O_P512);          // [wave][frame]
    double* const red = reinterpret_cast<double*>(smem + O_RED);          // [wave][frame]
    int* const accf = reinterpret_cast<int*>(smem + O_ACC);

    // eight consecutive tiles = one 128-byte line of every (bin, N) row: on ONE XCD (workgroup i runs on XCD i % 8)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = (slot >> 3) * 64 + xcd * 8 + (slot & 7);
    if (tile >= g.ntiles) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 3, b = lane >> 2;
    const bool bit2 = (lane & 4) != 0;
    const int b7 = b & 7, kh = b >> 3;                                    // layers 1 / 2: the lane's block of rows, its half of k
    const int fb = 32 * wave_u;
    const unsigned long long t_entry = g.dbg ? __builtin_amdgcn_s_memtime() : 0ull;   // (diagnostic stamps: slots 10 / 11 = prologue / chain in shader clocks)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.wcopy), 0, (int)g.wcopy_bytes, 0x00020000);
    // four consecutive k (quad o) of row `row` of an fp32 copy with nt 32-row tiles per k-step of 8: [k-step][row tile][half * 32 + row % 32][4]
    auto ldq = [&](int64_t base_elems, int nt, int row, int o) __attribute__((always_inline)) {
        const unsigned off = (unsigned)(base_elems * 4) + (unsigned)((((o >> 1) * nt + (row >> 5)) * 64 + (o & 1) * 32 + (row & 31)) * 16);
        return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)off, 0, 0));
    };

    // ---- resident weights (once per launch) ----
    float w5A[2][HD], w4A[HD / 2], w3A[ZD / 2];
#pragma unroll
    for (int R = 0; R < 2; ++R) {
        const int row = 128 * wave_u + 64 * R + 4 * b + j;                // (as the A operand the lane's low bits are the row within the block)
#pragma unroll
        for (int o = 0; o < HD / 4; ++o) {
            const f32x4_t v = ldq(g.oW5, NT_OUT, row, o);
            w5A[R][4 * o] = v[0]; w5A[R][4 * o + 1] = v[1]; w5A[R][4 * o + 2] = v[2]; w5A[R][4 * o + 3] = v[3];
        }
    }
    {
        const int row = fb + 4 * b7 + j;
#pragma unroll
        for (int o = 0; o < HD / 8; ++o) {
            const f32x4_t v = ldq(g.oW4, 4, row, (HD / 8) * kh + o);
            w4A[4 * o] = v[0]; w4A[4 * o + 1] = v[1]; w4A[4 * o + 2] = v[2]; w4A[4 * o + 3] = v[3];
        }
#pragma unroll
        for (int o = 0; o < ZD / 8; ++o) {
            const f32x4_t v = ldq(g.oW3, 4, row, (ZD / 8) * kh + o);
            w3A[4 * o] = v[0]; w3A[4 * o + 1] = v[1]; w3A[4 * o + 2] = v[2]; w3A[4 * o + 3] = v[3];
        }
    }
    // the tile's own operands -- X2 / Vb rows, gain, start latents, the labels of the 16-row label forms -- are requested HERE, behind the
    // resident fragments and in front of the first barrier (which waits for every load): one memory round trip for the whole prologue
    // instead of four in a row (fragments -> label weights -> labels -> X2 / Vb: 12.9k clocks per launch before)
    const int64_t n0 = (int64_t)tile * T4;
    const bool live = n0 + j < g.N;
    const int64_t nf = live ? n0 + j : g.N - 1;                            // clamped frame index of this lane
    const float g_n = g.g ? g.g[nf] : 1.f;
    // (F, N) matrices through buffer descriptors: bin 128 w + 64 R + 4 b + i of frame nf = ONE per-lane byte offset (bin 4 b) + a wave-uniform offset
    const int fn_bytes = (int)((int64_t)XD * g.N * 4);                     // < 2^31: checked by the launcher
    const int voff = (int)(((int64_t)(4 * b) * g.N + nf) * 4);
    const unsigned rowb = (unsigned)g.N * 4u;
    auto soff = [&](int R, int i) __attribute__((always_inline)) { return (int)((unsigned)(128 * wave_u + 64 * R + i) * rowb); };
    float x2_512 = 0.f, vb_512 = 0.f;
    f32x4_t xvR[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, vvR[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (g.X2) {
        const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X2), 0, fn_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_vb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Vb), 0, fn_bytes, 0x00020000);
#pragma unroll
        for (int R = 0; R < 2; ++R) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xvR[R][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x2, voff, soff(R, i), 0));
                vvR[R][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_vb, voff, soff(R, i), 0));
            }
        }
        if (wave_u == 3) { x2_512 = g.X2[(int64_t)512 * g.N + nf]; vb_512 = g.Vb[(int64_t)512 * g.N + nf]; }
    }
    float z = 0.f, zp = 0.f;                                               // chain state (wave 0): lane (b, j) holds latent b of frame j
    if (wave_u == 0 && g.nit > 0) z = g.Z0[(int64_t)b * g.N + nf];
    float yv2[2][16];                                                      // (label forms of up to 16 rows) the labels of the thread's two frames
    if constexpr (YP > 0 && YP != NO) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int fr = 2 * (tid >> 7) + e;
            const bool in = n0 + fr < g.N;
#pragma unroll
            for (int q = 0; q < 16; ++q) yv2[e][q] = (q < g.ydim && in) ? g.y[(int64_t)q * g.N + n0 + fr] : 0.f;
        }
    }
    for (int i = tid; i < 2 * HD + NO; i += 256) Bias[i] = g.bias[i];
    // element (row, column k) of a copy: k-step k / 8, lane' = (k % 8) / 4 * 32 + row % 32, element k % 4
    const float* const wc = reinterpret_cast<const float*>(g.wcopy);
    auto welem = [&](int64_t base, int nt, int row, int k) __attribute__((always_inline)) {
        return wc[base + ((int64_t)((k / 8) * nt + (row >> 5)) * 64 + ((k % 8) / 4) * 32 + (row & 31)) * 4 + (k % 4)];
    };
    if (tid < HD) w512s[tid] = welem(g.oW5, NT_OUT, 512, tid);            // row 512 of the output layer
    float wy[16];                                                          // (label forms of up to 16 rows) W3[f][16 + q] of the thread's feature
    if constexpr (YP > 0 && YP != NO) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wy[q] = welem(g.oW3, 4, tid & (HD - 1), ZD + q);
    }
    __syncthreads();
    const float b512 = Bias[OB5 + 512];

    // ---- per tile: label part of decoder layer 1 (fp32, constant along the chain), X2 / Vb -> LDS ----
    if constexpr (YP == NO) {
        // 513 label rows: c1[frame][feature] = b3 + W3[:, 16:] y, the label block gathered quad by quad, the lane halves take 66 quads each
        float* const Yb = reinterpret_cast<float*>(smem + O_X2);           // [frame][LDY]
        for (int idx = tid; idx < T4 * NO; idx += 256) {
            const int f = idx >> 2, fr = idx & 3;
            Yb[fr * LDY + f] = (f < g.ydim && n0 + fr < g.N) ? g.y[(int64_t)f * g.N + n0 + fr] : 0.f;
        }
        __syncthreads();
        f32x4_t ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
        const int row = fb + 4 * b7 + j;
        constexpr int QH = XP / 8;                                         // quads per lane half (528 label columns = 132 quads)
#pragma unroll 2
        for (int o = 0; o < QH; ++o) {
            const int oq = QH * kh + o;
            const f32x4_t a = ldq(g.oW3, 4, row, ZD / 4 + oq);
            const f32x4_t yv = *reinterpret_cast<const f32x4_t*>(Yb + j * LDY + 4 * oq);
            ca = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], yv[0], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], yv[1], cb, 0, 0, 0);
            ca = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], yv[2], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], yv[3], cb, 0, 0, 0);
        }
        f32x4_t c = ca + cb;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = xsum32(c[i]);
        const f32x4_t b3v = *reinterpret_cast<const f32x4_t*>(Bias + fb + 4 * b7);
        if (b < 8) *reinterpret_cast<f32x4_t*>(c1s + j * LDC + fb + 4 * b7) = c + b3v;
        __syncthreads();                                                   // the label image is consumed: its area is X2 / Vb from here on
    } else {
        const int f = tid & (HD - 1), fg = tid >> 7;                       // feature, pair of frames
        const float b3 = Bias[f];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float c = b3;
            if constexpr (YP > 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) c = fmaf(wy[q], yv2[e][q], c);
            }
            c1s[(2 * fg + e) * LDC + f] = c;
        }
    }
    if (g.X2) {                                                            // (requested in front of the first barrier)
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            X2s[(wave_u * 2 + R) * 64 + lane] = xvR[R];
            Vbs[(wave_u * 2 + R) * 64 + lane] = vvR[R];
        }
    }

    float prior_cur = 0.f;
    double ll_cur = 0.0;
    const int zoff = (int)(((int64_t)b * g.N + nf) * 4);                   // element (latent b, frame nf) of a (16, N) matrix

    unsigned long long tlast = 0ull, tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (g.dbg) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tsum[k] += t - tlast; tlast = t; }
    };
    // one decoder pass over the latents in Zb: EPI(row group R, bin offset i, pre-activation incl. bias, x2, vb) for the lane's eight bins
    // (bin 128 w + 64 R + 4 b + i of frame j), EPI512(pre-activation) on wave 3; ends BEHIND the output layer (no trailing barrier)
    // (PRE runs behind the requests of the latents and in front of the first MFMA: work that needs no MFMA result fills the LDS round trip)
    auto pass = [&](auto&& pre, auto&& epi, auto&& epi512) __attribute__((always_inline)) {
        // layer 1: z -> h1 (the lane halves take eight latents each)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        {
            const f32x4_t z0 = *reinterpret_cast<const f32x4_t*>(Zb + j * LDZ + 8 * kh), z1 = *reinterpret_cast<const f32x4_t*>(Zb + j * LDZ + 8 * kh + 4);
            pre();
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w3A[e], z0[e], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w3A[4 + e], z1[e], acc, 0, 0, 0);
        }
        f32x4_t v;
        {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(c1s + j * LDC + fb + 4 * b7);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = tanh_(xsum32(acc[i]) + c[i]);
            if (b < 8) *reinterpret_cast<f32x4_t*>(Ha + j * LDH + fb + 4 * b7) = v;
        }
        stamp(2);
        __syncthreads();                                                   // B1
        stamp(3);
        // layer 2: h1 -> h2 (the lane halves take 64 of the 128 k each), and this wave's 32 terms of bin 512's pre-activation
        {
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            // (the whole activation row of the lane's k half is requested first: four MFMAs -- 33 clocks -- do not cover an LDS round trip, and
            // with one quad in flight the phase took 1 400 clocks for 525 of matrix work)
            const float* const hr = Ha + j * LDH + (HD / 2) * kh;
            f32x4_t hv[HD / 8];
#pragma unroll
            for (int o = 0; o < HD / 8; ++o) hv[o] = *reinterpret_cast<const f32x4_t*>(hr + 4 * o);
            __builtin_amdgcn_sched_barrier(0);                             // (hipcc otherwise sinks every read to just above its MFMAs, two quads in flight)
#pragma unroll
            for (int o = 0; o < HD / 8; ++o) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w4A[4 * o], hv[o][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w4A[4 * o + 1], hv[o][1], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w4A[4 * o + 2], hv[o][2], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w4A[4 * o + 3], hv[o][3], a1, 0, 0, 0);
            }
            acc = a0 + a1;
        }
        {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(Bias + OB4 + fb + 4 * b7);
            const f32x4_t w = *reinterpret_cast<const f32x4_t*>(w512s + fb + 4 * b7);
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = tanh_(xsum32(acc[i]) + c[i]); p = fmaf(v[i], w[i], p); }
            if (b < 8) *reinterpret_cast<f32x4_t*>(Hb + j * LDH + fb + 4 * b7) = v;
            p = bsum<8>(p, bit2);                                          // over the wave's eight row blocks (both lane halves hold them)
            if (b == 0) p512[wave_u * T4 + j] = p;
        }
        stamp(4);
        __syncthreads();                                                   // B2
        stamp(5);
        // output layer: two groups of 64 bins per wave, their chains alternating (a dependent MFMA follows two instructions behind)
        f32x4_t oa[2];
        oa[0] = *reinterpret_cast<const f32x4_t*>(Bias + OB5 + 128 * wave_u + 4 * b);
        oa[1] = *reinterpret_cast<const f32x4_t*>(Bias + OB5 + 128 * wave_u + 64 + 4 * b);
        {
            const float* const hr = Hb + j * LDH;
            constexpr int RING = 8;                                        // quads of h2 in flight: eight MFMAs (66 clocks) per quad
            f32x4_t hv[RING];
#pragma unroll
            for (int o = 0; o < RING; ++o) hv[o] = *reinterpret_cast<const f32x4_t*>(hr + 4 * o);
            __builtin_amdgcn_sched_barrier(0);
            sfor<0, HD / 4>([&](auto oc) {
                constexpr int o = decltype(oc)::value;
                sfor<0, 4>([&](auto ec) {
                    constexpr int e = decltype(ec)::value, k = 4 * o + e;
                    mf4<(k < KAG), false, (k == 0)>(oa[0], w5A[0][k], hv[o % RING][e]);
                    mf4<(k < KAG), (k == HD - 1), (k == 0)>(oa[1], w5A[1][k], hv[o % RING][e]);
                });
                if constexpr (o + RING < HD / 4) { hv[o % RING] = *reinterpret_cast<const f32x4_t*>(hr + 4 * (o + RING)); __builtin_amdgcn_sched_barrier(0); }
            });
        }
#pragma unroll
        for (int R = 0; R < 2; ++R) {
            const f32x4_t xq = X2s[(wave_u * 2 + R) * 64 + lane], vq = Vbs[(wave_u * 2 + R) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) epi(R, i, oa[R][i], xq[i], vq[i]);
        }
        if (wave_u == 3) {
            const float a = b512 + ((p512[j] + p512[T4 + j]) + (p512[2 * T4 + j] + p512[3 * T4 + j]));
            epi512(a);
        }
        stamp(6);
    };

    // The decoder variances of the kept samples (compute_Vs, mcem.py:280-290) are those of the chain's STATE at the kept steps, and a state's
    // variances were computed by the pass that proposed it: every lane keeps exp(pre-activation) of its eight bins for the current state
    // (vcur) and for the last proposal (vprop), takes the proposal's when wave 0 has accepted it (flag through LDS, read behind the next
    // barrier) and writes vcur out at the kept steps -- the same bits a decoder pass over the stored sample returns (tested), without the
    // ten extra passes per chain that were a fifth of its time.
    const bool want_vs = g.Vs != nullptr && g.nit > 0;
    float vprop[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, vcur[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float vprop512 = 0.f, vcur512 = 0.f;
    auto settle = [&](int md) __attribute__((always_inline)) {             // md: the step whose decision `accf` holds (-1: the evaluation of the initial state)
        if (accf[j] != 0) {
#pragma unroll
            for (int R = 0; R < 2; ++R)
#pragma unroll
                for (int i = 0; i < 4; ++i) vcur[R][i] = vprop[R][i];
            vcur512 = vprop512;
        }
        if (md >= g.burnin && live) {
            float* const vs_r = g.Vs + (int64_t)(md - g.burnin) * XD * g.N;
            const __amdgpu_buffer_rsrc_t rs_vs = __builtin_amdgcn_make_buffer_rsrc(vs_r, 0, fn_bytes, 0x00020000);
#pragma unroll
            for (int R = 0; R < 2; ++R)
#pragma unroll
                for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vcur[R][i]), rs_vs, voff, soff(R, i), 0);
            if (wave_u == 3 && b == 0) vs_r[(int64_t)512 * g.N + nf] = vcur512;
        }
    };
    const int mstart = g.nit > 0 ? -1 : 0;
    const int mend = g.nit > 0 ? g.nit : 0;
    // the draws of chain step m + 1 are requested while step m runs (wave 0)
    float nz = 0.f, lu = 0.f;
    const __amdgpu_buffer_rsrc_t rs_nz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.noise), 0, (int)((int64_t)g.nit * ZD * g.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.logu), 0, (int)((int64_t)g.nit * g.N * 4), 0x00020000);
    auto load_draws = [&](int m) __attribute__((always_inline)) {
        if (m < g.nit) {
            const int mb = __builtin_amdgcn_readfirstlane((int)((unsigned)m * (unsigned)ZD * rowb));
            const int mlu = __builtin_amdgcn_readfirstlane((int)((unsigned)m * rowb));
            nz = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_nz, zoff, mb, 0));
            lu = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_lu, (int)(nf * 4), mlu, 0));
        }
    };
    if (wave_u == 0 && g.nit > 0) load_draws(0);
    // the kept sample and the trace of step m leave the CU behind step m + 1's proposal (see the 32-frame kernel)
    int pend_m = -1; float pend_prob = 0.f; bool pend_acc = false;
    auto flush_step = [&]() __attribute__((always_inline)) {
        if (pend_m >= 0) {
            if (live && b == 0) {
                if (g.accp) g.accp[(int64_t)pend_m * g.N + nf] = pend_prob;
                if (g.accd) g.accd[(int64_t)pend_m * g.N + nf] = pend_acc ? 1 : 0;
            }
            if (pend_m >= g.burnin && live) g.Zs[((int64_t)nf * g.R + (pend_m - g.burnin)) * ZD + b] = z;     // mcem.py:271-273
            pend_m = -1;
        }
    };
    if (g.dbg) tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long t_loop = tlast;
    for (int m = mstart; m < mend; ++m) {
        float prior_p = 0.f, lu_cur = 0.f;
        if (wave_u == 0) {
            if (m >= 0) {
                zp = z + g.sd * nz;                                                               // mcem.py:244
                asm volatile("v_mov_b32 %0, %1" : "=v"(lu_cur) : "v"(lu));                        // (a copy hipcc cannot sink below the next request: 32-frame kernel)
            } else {
                zp = z;
            }
            Zb[j * LDZ + b] = zp;
            prior_p = bsum<16>(zp * zp, bit2);
            __builtin_amdgcn_sched_barrier(0);
            if (m >= 0) load_draws(m + 1);
        }
        stamp(0);
        __syncthreads();                                                   // B0
        stamp(1);
        double ll = 0.0;
        float slog = 0.f, sdiv = 0.f;                                      // sums of log2(vx) and x2 / vx over the lane's eight bins
        pass(
            [&]() { if (want_vs && m > mstart) settle(m - 1); },          // (its flag travels beside the latents: at the top of the step the read was waited for on the spot)
            [&](int R, int i, float a, float x2, float vb) {
                const float ea = exp_(a);
                vprop[R][i] = ea;
                const float vx = fmaf(g_n, ea, vb);                                               // mcem.py:248-249
                slog += __builtin_amdgcn_logf(vx);                                                // mcem.py:252-253: log(vx) + x2 / vx
                sdiv = fmaf(x2, __builtin_amdgcn_rcpf(vx), sdiv);
            },
            [&](float a) {
                const float ea = exp_(a);
                vprop512 = ea;
                const float vx = fmaf(g_n, ea, vb_512);
                const float term = log_(vx) + div_(x2_512, vx);
                if (b == 0) ll += (double)term;
            });
        ll += (double)fmaf(slog, 0.693147180559945309f, sdiv);
        ll = bsum16d(ll, bit2);
        if (b == 0) red[wave_u * T4 + j] = ll;
        if (wave_u == 0) flush_step();                                     // (the last step's kept sample and trace: in wave 0's wait for wave 3's bin 512, not in the serial section)
        stamp(7);
        __syncthreads();                                                   // B3
        stamp(8);
        if (wave_u == 0) {
            const double ll_p = red[j] + red[T4 + j] + red[2 * T4 + j] + red[3 * T4 + j];
            if (m < 0) {
                ll_cur = ll_p; prior_cur = prior_p;
                if (b == 0) accf[j] = 1;
            } else {
                const float acc_prob = (float)(ll_cur - ll_p) + 0.5f * (prior_cur - prior_p);       // mcem.py:252-254
                const bool is_acc = lu_cur < acc_prob;                                               // mcem.py:257 (the same bits in the frame's sixteen lanes)
                if (is_acc) { ll_cur = ll_p; prior_cur = prior_p; z = zp; }
                if (b == 0) accf[j] = is_acc ? 1 : 0;
                pend_m = m; pend_prob = acc_prob; pend_acc = is_acc;                                 // stored behind the next proposal (flush_step)
            }
        }
        // red / p512 / Zb are next written behind the barriers of the following pass
    }
    if (wave_u == 0) flush_step();
    if (wave_u == 0 && g.nit > 0 && g.Zlast != nullptr && live) g.Zlast[(int64_t)b * g.N + nf] = z;   // the chain's final state (may be Z0 itself)
    if (g.dbg && lane == 0 && g.nit > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) g.dbg[((size_t)tile * 4 + wave_u) * 16 + k] = tsum[k];
        g.dbg[((size_t)tile * 4 + wave_u) * 16 + 9] = (unsigned long long)(mend - mstart);
        g.dbg[((size_t)tile * 4 + wave_u) * 16 + 10] = t_loop - t_entry;
        g.dbg[((size_t)tile * 4 + wave_u) * 16 + 11] = __builtin_amdgcn_s_memtime() - t_loop;
    }

    if (want_vs) {                                                         // the last step's decision
        __syncthreads();
        settle(mend - 1);
    }
    // ---- decode mode (dvae_mcem_decode): Vs[r] = decoder([Zs[:, r, :] | y])  (mcem.py:280-290) ----
    if (g.Vs != nullptr && g.nit == 0) {
        for (int r_s = 0; r_s < g.R; ++r_s) {
            __syncthreads();
            if (wave_u == 0) Zb[j * LDZ + b] = g.Zs[((int64_t)nf * g.R + r_s) * ZD + b];
            __syncthreads();
            float* const vs_r = g.Vs + (int64_t)r_s * XD * g.N;
            const __amdgpu_buffer_rsrc_t rs_vs = __builtin_amdgcn_make_buffer_rsrc(vs_r, 0, fn_bytes, 0x00020000);
            pass(
                []() {},
                [&](int R, int i, float a, float, float) {
                    if (live) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, exp_(a)), rs_vs, voff, soff(R, i), 0);
                },
                [&](float a) { if (live && b == 0) vs_r[(int64_t)512 * g.N + nf] = exp_(a); });
        }
    }
}

template <int YP>
static int launch_resident4_t(const MhArgs& a, hipStream_t s) {
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)mcem_resident4_kernel<YP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q4::LDS);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(mcem_resident4_kernel, %zu B LDS): %s", q4::LDS, hipGetErrorString(e)); return (int)e; }
        attr_done[dev] = true;
    }
    const int grid = (a.ntiles + 63) / 64 * 64;                           // whole groups of 8 tiles x 8 XCDs; workgroups past the last tile return at once
    hipLaunchKernelGGL((mcem_resident4_kernel<YP>), dim3(grid), dim3(256), q4::LDS, s, a);
    DVAE_LAUNCH_OK("mcem_resident4_kernel");
    return 0;
}

bool resident4_chain_supported(int precision, int yp) { return precision == DVAE_PREC_F32 && (yp == 0 || yp == 16 || yp == XP); }

// a.ntiles: 4-frame tiles
int launch_resident4_chain(int yp, const MhArgs& a, hipStream_t s) {
    if (yp == 0) return launch_resident4_t<0>(a, s);
    if (yp == 16) return launch_resident4_t<16>(a, s);
    if (yp == XP) return launch_resident4_t<NO>(a, s);
    set_error("mcem resident chain (4 frames): label rows 0, 1..16 or 513 only");
    return DVAE_E_UNSUPPORTED;
}

}  // namespace fused
}  // namespace dvae
