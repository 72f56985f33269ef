// Device helpers shared by the rows kernels of the fused train step (train_fused.hip: 4-wave kernel; train_rows2.hip:
// 8-wave chain + helper kernel): reparametrisation noise, kernel arguments, tile loaders, LDS -> stash transposition.
#pragma once
#include "fused_tiles.hpp"
#include "apply_types.hpp"

namespace dvae {
namespace fused {

// Philox4x32-10 (Salmon et al., SC'11) -> four standard normals by Box-Muller.  Counter = (frame lo, frame hi, step lo,
// step hi << 8 | draw), key = seed: every frame of every step has its own stream, independent of tiling and grid.
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned long long frame, unsigned long long step, unsigned draw,
                                               float (&out)[4]) {
    unsigned c0 = (unsigned)frame, c1 = (unsigned)(frame >> 32), c2 = (unsigned)step, c3 = ((unsigned)(step >> 32) << 8) | draw;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.f / 16777216.f), u1 = ((float)(c1 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.f / 16777216.f), u3 = ((float)(c3 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float ra = sqrtf(-2.f * __logf(u0)), rb = sqrtf(-2.f * __logf(u2));
    float sa, ca, sb, cb;
    __sincosf(6.283185307179586f * u1, &sa, &ca);
    __sincosf(6.283185307179586f * u3, &sb, &cb);
    out[0] = ra * ca; out[1] = ra * sa; out[2] = rb * cb; out[3] = rb * sb;
}
// noise of latent features 4h .. 4h+3 (draw 2h) and 8+4h .. 8+4h+3 (draw 2h+1) of one frame: the C-tile ownership of wave 0
__device__ __forceinline__ void frame_noise8(unsigned long long seed, unsigned long long frame, unsigned long long step, int h, float (&e)[8]) {
    float a[4], b[4];
    philox_normal4(seed, frame, step, 2u * h, a);
    philox_normal4(seed, frame, step, 2u * h + 1u, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) { e[j] = a[j]; e[4 + j] = b[j]; }
}

struct RowsArgs {
    const float* x; const float* y; const float* eps;
    int ldx, ldy, ydim;
    int fastx, fasty;      // rows are dense (ld == 513) and 16-byte aligned: whole-tile vector loads
    int64_t B, Bp;
    unsigned long long rng_seed, rng_step;   // in-kernel reparametrisation noise (eps == nullptr)
    const int64_t* rows;                      // optional gather: frame b of the step is row rows[b] of x / y (epoch shuffle without a copy)
    int64_t n_rows;                           // rows of the frame store behind x / y when `rows` is set: indices outside [0, n_rows) are
    int* bad_rows;                            // clamped to row 0 and counted here (never dereferenced out of range)
    int ntiles;
    float invB, elbo_eps;
    const void *W1s, *W2s, *Wmvs, *W3s, *W4s, *W5s, *W5t, *W4t, *W3zt, *Wmvt, *W2t;
    const float *b1, *b2, *bmu, *blv, *b3, *b4, *b5;
    const float* w5last;   // fp32 row 512 of the output layer (the one real feature of the 17th 32-row tile)
    void *xT, *yT, *h1T, *h2T, *dh1T, *dh2T, *dmlvT, *zT, *d1T, *d2T, *dd1T, *dd2T, *daT;
    const void* wcopy; int64_t wcopy_bytes;   // whole weight-copy buffer (one buffer descriptor)
    int64_t spl;                              // stash: elements between the hi and lo operand planes (PolX3)
    unsigned wpl_bytes;                       // weight copies: bytes between the planes
    // M2_info (DeepGenerativeModel_v5): classifier on x, auxiliary classifier on z (both 128-128-1, relu/relu/sigmoid)
    const void *Wc1s, *Wc2s, *Wc2t, *Wa1s, *Wa1t, *Wa2s, *Wa2t;
    const float *bc1, *bc2, *wc3, *bc3, *ba1, *ba2, *wa3, *ba3;
    void *c1T, *c2T, *dc1T, *dc2T, *dc3T, *a1T, *a2T, *da1T, *da2T, *da3T;
    float alpha, beta, gamma;
    // Whole-model autograd path of the drop-in modules (packages/models/models.py -> disentangled-vae_amd/module_path.py):
    //   mode 0: fused train step (loss + backward on chip);
    //   mode 1: forward only -- r = exp(a) [B, 513], mu / log_var / z [B, 16] are written out, nothing is stashed;
    //   mode 2: backward from upstream gradients -- the forward is recomputed, d a = g_r * r, d mu += g_mu, d log_var += g_lv,
    //           d z += g_z (any of them may be null = zero); the stash feeds the weight-gradient kernel as in mode 0.
    int mode;
    float *out_r, *out_mu, *out_lv, *out_z;
    const float *g_r, *g_mu, *g_lv, *g_z;
    int ld_r, ld_gr;
    double* partials;
    unsigned long long* dbg;    // diagnostic stamps (100 MHz wall clock), null in production
    int ablate;                 // diagnostic ablation mask (env DVAE_ABLATE), 0 in production
    int stash_inputs;           // bit 0: stash the x tile, bit 1: stash the label tile; a cleared bit = the weight-gradient kernel reads that input from its fp32 matrix (8-wave kernel only)
    // Label tiles that one bf16 plane holds exactly (binary VAD / IBM labels) have an all-zero lo plane: the 8-wave kernel then stores only
    // the hi plane of the label stash (ylo_skip != 0) and raises *ylo_epoch to `launch_id` as soon as ANY tile of the launch does need its
    // lo plane; the weight-gradient kernel reads the label lo plane (and issues the hi * lo products) only in that case.
    // ylo_dirty[tile]: this tile slot's lo plane in the stash holds non-zero values from an earlier launch -- a tile that needs no lo plane
    // now still rewrites it (with its zeros) once, so that a launch in which SOME tile is flagged never reads stale lo values of the others.
    unsigned* ylo_epoch; unsigned launch_id; int ylo_skip; int* ylo_dirty;
    // deferred optimizer step (apply_common.hpp): the previous step's Adam update at the top of this launch, the loss scalars at its end
    DeferArgs defer;
};

#ifdef DVAE_FINE_STAMPS
#define DVAE_FSTAMP(i) do { if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define DVAE_FSTAMP(i) do { } while (0)
#endif
#define DVAE_STAMP(i) do { if (g.dbg && tid == 0) g.dbg[(size_t)blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)


// LDS tile [32 frames][features fbase .. fbase+31] -> fragment-major stash tile (E consecutive frames of
// one feature = one 16-byte fragment).  Called by the wave that wrote those LDS columns, with EXEC all ones.
// bf16: the transposition is done by the LDS itself: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of
// a 4-row x 16-column block (gfx950), so a fragment costs 2 reads instead of 8 two-byte ones.  The lane with index
// 4q + p in its group supplies the address of row q, columns 4p .. 4p+3 of the block.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr16(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// Stash stores go out as buffer stores with selectable cache-policy bits (R2_STASH_AUX: gfx950 buffer aux, 0 = plain, 1 = sc0, 2 = nt,
// 16 = sc1 = write-through, the line is not kept in the writing XCD's L2 -- the stash is read by ANOTHER kernel, on other XCDs).  They
// are compiler-visible stores (hazards, wait counts); a first attempt with inline-asm `global_store_dwordx4 ... sc1` corrupted 0.7 % of
// the stash bytes (no wait state behind a 128-bit store whose data registers the next VALU instruction rewrites: the compiler's hazard
// recogniser does not see into inline asm) and still 0.06 % with an s_nop behind it.
// Measured (round 3, M2 y513, 8192 frames, bf16x3, same box, alternating; us per step): plain stash and slab stores 74.7, sc1 stash
// stores 73.1, sc1 stash and slab stores 72.9.  With plain stores the 80 MB of stash sit dirty in the L2s until evicted or until the
// kernel ends.
#ifndef R2_STASH_AUX
#define R2_STASH_AUX 16
#endif
template <typename Frag>
__device__ __forceinline__ void stash_store16(void* base_uniform, int voff_bytes, const Frag& f) {
#if R2_STASH_AUX == 0
    *reinterpret_cast<Frag*>((char*)base_uniform + voff_bytes) = f;
#else
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, f), rs, voff_bytes, 0, R2_STASH_AUX);
#endif
}

// SRC16: the LDS planes hold split-fp16 values scaled by X16::XS (the x image of encoder layer 1): the stash gets the split-bf16 planes of
// the unscaled values, which is what the weight-gradient kernel multiplies
template <typename P, bool SRC16 = false>
__device__ __forceinline__ void stash_tile(const typename P::T* lds, int ldl, int fbase, typename P::T* stash_tile_ptr, int64_t spl,
                                           int64_t b0, int l31, int h, float scale = 1.f, int col_limit = 1 << 30, int nplanes = P::NP) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    constexpr int E = P::E;
    if (stash_tile_ptr == nullptr) return;
    T* const dbase = stash_tile_ptr + (b0 / P::KSTEP) * (64 * E);              // wave-uniform: the tile's first fragment block
    if constexpr (sizeof(T) == 2) {
        const int i16 = l31 & 15, q = i16 >> 2, pp = i16 & 3, cg = l31 >> 4;
        const T* blk = lds + q * ldl + fbase + 16 * cg + 4 * pp;
        const bool keep = fbase + l31 < col_limit;
        typedef short s16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int i = 0; i < TB / (2 * E); ++i) {
            const int gq = h + 2 * i;                      // frame group: frames gq*8 .. gq*8+7
            Frag f[P::NP];
#pragma unroll
            for (int pl = 0; pl < P::NP; ++pl) {
                const T* bp = blk + pl * Pl<P>::lds;
                const s16x4 r0 = lds_tr16(bp + (8 * gq) * ldl), r1 = lds_tr16(bp + (8 * gq + 4) * ldl);
                s16x8 raw = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                f[pl] = __builtin_bit_cast(Frag, raw);
            }
            if constexpr (SRC16 && P::NP == 2) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const float v = (X16::val(f[0][j]) + X16::val(f[1][j])) * X16::XINV;
                    f[0][j] = P::cvt(v); f[1][j] = P::cvt(v - (float)f[0][j]);
                }
            } else
            if (scale != 1.f) {
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if constexpr (P::NP == 2) {
                        const float v = ((float)f[0][j] + (float)f[1][j]) * scale;
                        f[0][j] = P::cvt(v); f[1][j] = P::cvt(v - (float)f[0][j]);
                    } else f[0][j] = P::cvt((float)f[0][j] * scale);
                }
            }
            if (!keep) {
#pragma unroll
                for (int pl = 0; pl < P::NP; ++pl)
#pragma unroll
                    for (int j = 0; j < E; ++j) f[pl][j] = P::cvt(0.f);
            }
            stash_store16(dbase, (l31 * E + gq * 32 * E) * (int)sizeof(T), f[0]);
            if constexpr (P::NP == 2) { if (nplanes == 2) stash_store16(dbase + spl, (l31 * E + gq * 32 * E) * (int)sizeof(T), f[1]); }      // wave-uniform
        }
    } else {
#pragma unroll
        for (int i = 0; i < TB / (2 * E); ++i) {
            const int gq = h + 2 * i;                      // frame group: frames gq*E .. gq*E+E-1
            Frag f;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const T v = lds[(gq * E + j) * ldl + fbase + l31];
                f[j] = scale == 1.f ? v : P::cvt((float)v * scale);
                if (fbase + l31 >= col_limit) f[j] = P::cvt(0.f);
            }
            stash_store16(dbase, (l31 * E + gq * 32 * E) * (int)sizeof(T), f);
        }
    }
}

// generic (edge tile / strided / unaligned input): global [32 frames][ncols] fp32 -> LDS as T, zero padded
template <typename P, bool F16 = false, typename RowOf>
__device__ __forceinline__ void load_rows_to_lds(const float* __restrict__ src, int ld, int ncols, int pcols, int64_t b0, int64_t B,
                                                 typename P::T* U, int ldu, int tid, RowOf rowof, float* xf = nullptr,
                                                 float* log2sum = nullptr, float eps = 0.f) {
    const int total = TB * pcols;
    for (int idx = tid; idx < total; idx += 256) {
        const int row = idx / pcols, col = idx - row * pcols;
        float v = 0.f;
        if (col < ncols && b0 + row < B) {
            v = src[rowof(row) * ld + col];
            if (log2sum && col < XD - 1) *log2sum += __builtin_amdgcn_logf(v + eps);      // live frames, bins 0 .. 511 (see tile513_log2sum)
        }
        if constexpr (F16 && P::NP == 2) {                         // split fp16 of XS * v (struct X16)
            const float vs = v * X16::XS;
            const typename P::T vh = X16::hi<typename P::T>(vs);
            U[row * ldu + col] = vh;
            U[Pl<P>::lds + row * ldu + col] = X16::hi<typename P::T>(vs - X16::val(vh));
        } else {
            const typename P::T vh = P::cvt(v);
            U[row * ldu + col] = vh;
            if constexpr (P::NP == 2) U[Pl<P>::lds + row * ldu + col] = P::cvt(v - (float)vh);
        }
        if (xf && col < ncols) xf[row * ncols + col] = v;
    }
}

// dense 513-column tile (32 rows back to back in memory).  Thread t takes the 4-column chunks c = t + 256 i
// (i < 16): row c >> 7, columns 4 (c & 127) .. +3 -- shifts only, and the LDS image row U[row][col .. col+3] is
// one aligned 8-byte store.  The global address (row * 513 + col floats) is only 4-byte aligned: gfx950 takes
// dwordx4 loads at dword alignment.  Column 512 of row (t & 31) rides in slot 16.
constexpr int NQ513 = 17;
struct __attribute__((packed, aligned(4))) F4U { f32x4 v; };
// R2_IN_NT (round 5): 1 = the step's x / label tiles are requested non-temporally (each byte is used once, while the weight copies every CU of
// the XCD streams should stay in the 4 MB L2 beside them); 2 = the loss epilogue's re-read of x as well.  Measured SLOWER, same box, three rounds
// (profiles/r05_rows_nontemporal_inputs_ab.txt): rows kernel 43.5 -> 48.7 (1) -> 51.3 us (2) -- the nt loads themselves take longer than the L2
// lines they spare are worth; the default policy stays
#ifndef R2_IN_NT
#define R2_IN_NT 0
#endif
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
__device__ __forceinline__ f32x4 ld_in4(const float* p) {
#if R2_IN_NT >= 1
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4_a4*>(p));
#else
    return reinterpret_cast<const F4U*>(p)->v;
#endif
}
__device__ __forceinline__ f32x4 ld_in4_again(const float* p) {
#if R2_IN_NT >= 2
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4_a4*>(p));
#else
    return reinterpret_cast<const F4U*>(p)->v;
#endif
}
// chunks [I0, I1) of the tile (16 chunks of 4 columns per thread; slot 16 = column 512: tile513_issue_last)
template <int I0, int I1, typename RowOf>
__device__ __forceinline__ void tile513_issue_part(const float* __restrict__ base, RowOf rowof, f32x4 (&v)[NQ513], int tid) {
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const int c = tid + 256 * i;
        v[i] = ld_in4(base + rowof(c >> 7) * XD + 4 * (c & 127));
    }
}
template <typename RowOf>
__device__ __forceinline__ void tile513_issue_last(const float* __restrict__ base, RowOf rowof, f32x4 (&v)[NQ513], int tid) {
    v[16][0] = base[rowof(tid & 31) * XD + XD - 1];
}
template <typename RowOf>
__device__ __forceinline__ void tile513_issue(const float* __restrict__ base, RowOf rowof, f32x4 (&v)[NQ513], int tid) {
    tile513_issue_part<0, 16>(base, rowof, v, tid);
    tile513_issue_last(base, rowof, v, tid);
    __builtin_amdgcn_sched_barrier(0);     // all loads in flight before the first LDS commit
}
// chunks [I0, I1) -> (hi, lo) planes of the LDS image; lo_bits collects the OR of every lo-plane word written
template <typename P, int I0, int I1, bool F16 = false>
__device__ __forceinline__ void tile513_commit_part(const f32x4 (&v)[NQ513], typename P::T* U, int ldu, int tid, float* xf, unsigned long long& lo_bits) {
    typedef typename P::Pack4 Pack4;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 7, col = 4 * (c & 127);
        if (xf) {                                              // dense [frame][513] fp32 copy (rows 4-byte aligned)
            float* d = xf + row * XD + col;
            d[0] = v[i][0]; d[1] = v[i][1]; d[2] = v[i][2]; d[3] = v[i][3];
        }
        if constexpr (F16 && P::NP == 2) {                         // split fp16 of XS * v (struct X16)
            Pack4 pk, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float vs = v[i][j] * X16::XS;
                pk[j] = X16::hi<typename P::T>(vs);
                pl[j] = X16::hi<typename P::T>(vs - X16::val(pk[j]));
            }
            *reinterpret_cast<Pack4*>(U + row * ldu + col) = pk;
            *reinterpret_cast<Pack4*>(U + Pl<P>::lds + row * ldu + col) = pl;
            continue;
        }
        Pack4 pk;
        pk[0] = P::cvt(v[i][0]); pk[1] = P::cvt(v[i][1]); pk[2] = P::cvt(v[i][2]); pk[3] = P::cvt(v[i][3]);
        *reinterpret_cast<Pack4*>(U + row * ldu + col) = pk;
        if constexpr (P::NP == 2) {
            Pack4 pl;
            pl[0] = P::cvt(v[i][0] - (float)pk[0]); pl[1] = P::cvt(v[i][1] - (float)pk[1]);
            pl[2] = P::cvt(v[i][2] - (float)pk[2]); pl[3] = P::cvt(v[i][3] - (float)pk[3]);
            *reinterpret_cast<Pack4*>(U + Pl<P>::lds + row * ldu + col) = pl;
            lo_bits |= __builtin_bit_cast(unsigned long long, pl) & 0x7fff7fff7fff7fffull;   // -0 is not "non-zero"
        }
    }
}
// column 512 (slot 16), then the PCOLS - 513 zero columns
template <typename P, int PCOLS, bool F16 = false>
__device__ __forceinline__ void tile513_commit_last(const f32x4 (&v)[NQ513], typename P::T* U, int ldu, int tid, float* xf, unsigned long long& lo_bits) {
    constexpr int PADC = PCOLS - XD;
    if (tid < TB) {
        if (xf) xf[tid * XD + XD - 1] = v[16][0];
        if constexpr (F16 && P::NP == 2) {
            const float vs = v[16][0] * X16::XS;
            const typename P::T vh = X16::hi<typename P::T>(vs);
            U[tid * ldu + XD - 1] = vh;
            U[Pl<P>::lds + tid * ldu + XD - 1] = X16::hi<typename P::T>(vs - X16::val(vh));
        } else {
        const typename P::T vh = P::cvt(v[16][0]);
        U[tid * ldu + XD - 1] = vh;
        if constexpr (P::NP == 2) {
            const typename P::T vl = P::cvt(v[16][0] - (float)vh);
            U[Pl<P>::lds + tid * ldu + XD - 1] = vl;
            if ((float)vl != 0.f) lo_bits |= 1ull;
        }
        }
    }
    for (int idx = tid; idx < TB * PADC; idx += 256) {
        const int r = idx / PADC, c = XD + idx - r * PADC;
        U[r * ldu + c] = P::cvt(0.f);
        if constexpr (P::NP == 2) U[Pl<P>::lds + r * ldu + c] = P::cvt(0.f);
    }
}
template <typename P, int PCOLS, bool F16 = false>
__device__ __forceinline__ void tile513_commit(const f32x4 (&v)[NQ513], typename P::T* U, int ldu, int tid, float* xf = nullptr, bool* any_lo = nullptr) {
    unsigned long long lo_bits = 0ull;                          // OR of every lo-plane word this thread writes (any_lo: is the lo plane needed at all?)
    tile513_commit_part<P, 0, 16, F16>(v, U, ldu, tid, xf, lo_bits);
    tile513_commit_last<P, PCOLS, F16>(v, U, ldu, tid, xf, lo_bits);
    if (any_lo) *any_lo = lo_bits != 0ull;
}
// ---- the x stash of the 8-wave kernel under the split-fp16 x image (fused_tiles.hpp: struct X16), 256 helper threads ----
// The image in LDS holds split-fp16 planes of x / 8: an ABSOLUTE error floor, which is what a pre-activation needs.  The weight gradient
// dW1 = dpre1^T x feeds Adam, which normalises every element by its own history -- there each x needs its RELATIVE precision, down to the
// quietest bin -- so the stash does not come from the image: the helpers read the tile a second time (it is seconds old: L2 / Infinity
// Cache), column-wise -- lane = feature 64 q + lane, eight frames of one frame group per lane, i.e. a lane holds exactly the eight frames
// of a stash fragment -- and store the split-bf16 planes of the fp32 values: 2 x 512 contiguous bytes per wave instruction, no LDS, no
// transposition, any row stride, gather table or ragged tile, and in whatever phase the helpers have time for it.
// Feature groups [Q0, Q1) of nine: q < 8 = features 64 q .. 64 q + 63 (two 32-feature stash tiles); q = 8 = tile 16 (bin 512 + 31 zero rows).
template <typename P, int Q0, int Q1, typename RowOf>
__device__ __forceinline__ void xstash_reload(const float* __restrict__ src, int ld, RowOf rowof, int64_t b0, int64_t B, typename P::T* xT,
                                              int64_t spl, int64_t Bp, int tid) {
    typedef typename P::Frag Frag;
    static_assert(P::NP == 2 && P::E == 8, "x stash by reload: split-bf16 planes");
    if (xT == nullptr) return;
    const int w = tid >> 6, lane = tid & 63;                              // wave w: frames 8 w .. 8 w + 7 of the tile
    int64_t ro[8];
    bool live[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { ro[k] = rowof(8 * w + k) * (int64_t)ld; live[k] = b0 + 8 * w + k < B; }
    const int64_t kst = (b0 / P::KSTEP + (w >> 1)) * (64 * P::E) + (w & 1) * 32 * P::E;      // this frame group inside a feature tile's k-step blocks
    float v[Q1 - Q0][8];
#pragma unroll
    for (int q = Q0; q < Q1; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int f = 64 * q + lane;
            v[q - Q0][k] = src[ro[k] + (f < XD ? f : XD - 1)];                          // (clamped: features 513 .. 575 of group 8 do not exist)
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        const int f = 64 * q + lane;
        if (q == 8 && lane >= 32) continue;                                // group 8 is one 32-feature tile
        Frag fh, fl;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = (f < XD && live[k]) ? v[q - Q0][k] : 0.f;
            fh[k] = P::cvt(x);
            fl[k] = P::cvt(x - (float)fh[k]);
        }
        // (wave-uniform base + 32-bit per-lane byte offset: the x rows of the stash span 544 Bp elements < 2 GB up to 2^20 frames)
        const int64_t o = (int64_t)(f >> 5) * 32 * Bp + kst + (f & 31) * P::E;
        stash_store16(xT, (int)(o * (int64_t)sizeof(typename P::T)), fh);
        stash_store16(xT + spl, (int)(o * (int64_t)sizeof(typename P::T)), fl);
    }
}
// sum over this thread's share of a full dense tile of log2(x + eps), columns 0 .. 511 (hardware log2: the loss epilogue's log terms,
// taken while the tile sits in registers; utils.py:74 -- the caller scales by ln 2).  Bin 512 is not included: the wave that owns the
// 17th output tile computes that bin's whole term itself.
__device__ __forceinline__ float tile513_log2sum(const f32x4 (&v)[NQ513], float eps, int tid) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s0 += __builtin_amdgcn_logf(v[i][0] + eps) + __builtin_amdgcn_logf(v[i][1] + eps);
        s1 += __builtin_amdgcn_logf(v[i][2] + eps) + __builtin_amdgcn_logf(v[i][3] + eps);
    }
    return s0 + s1;
}

// LDS U[frame][col] -> fragment-major stash (see put_tile), 16 bytes (E frames of one feature) per
// store; feature rows up to `srows` (multiple of 32) are written, columns >= pcols as zeros
template <typename P, bool SRC16 = false>
__device__ __forceinline__ void stash_from_lds(const typename P::T* U, int ldu, int pcols, int srows, typename P::T* stash, int64_t spl,
                                               int64_t Bp, int64_t b0, int tid, int ft0 = 0, int ft1 = 1 << 30, int nplanes = P::NP) {
    typedef typename P::Frag Frag;
    constexpr int E = P::E;
    if constexpr (sizeof(typename P::T) == 2) {
        // one wave per 32-feature tile (wave-uniform loop: EXEC stays all ones for the transposing reads); feature tiles [ft0, ft1)
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int fte = ft1 < srows / 32 ? ft1 : srows / 32;
        for (int ft = ft0 + wave; ft < fte; ft += 4)
            stash_tile<P, SRC16>(U, ldu, 32 * ft, stash + (int64_t)ft * 32 * Bp, spl, b0, lane & 31, lane >> 5, 1.f, pcols, nplanes);
        return;
    }
    constexpr int groups = TB / E;
    const int total = srows * groups;
    for (int idx = tid; idx < total; idx += 256) {
        const int gi = idx / srows, f = idx - gi * srows;     // consecutive threads -> consecutive features
        Frag p;
#pragma unroll
        for (int j = 0; j < E; ++j) p[j] = (f < pcols) ? U[(gi * E + j) * ldu + f] : P::cvt(0.f);
        *reinterpret_cast<Frag*>(stash + (int64_t)(f >> 5) * 32 * Bp + (b0 / P::KSTEP) * (64 * E) + (gi * 32 + (f & 31)) * E) = p;
    }
}


// train_rows2.hip: the 8-wave chain + helper rows kernel (M1 / M2, bf16 and bf16x3 operand policies)
int launch_rows2(int precision, int model, int y_dim, const RowsArgs& a, int grid, hipStream_t s);
bool rows2_supported(int precision, int model);
// train_rows3.hip: 8 GEMM waves (16 x 16 x 32 MFMA, N-split) + 4 helper waves: M1 / M2 train step under the split-bf16 policy
int launch_rows3(int model, int y_dim, const RowsArgs& a, int grid, hipStream_t s);
bool rows3_supported(int precision, int model);

}  // namespace fused
}  // namespace dvae
