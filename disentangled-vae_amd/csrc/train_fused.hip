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
#include <math.h>
#include "common.hpp"
#include "../../include/dvae_train.h"

namespace dvae {
namespace fused {

constexpr int XD = 513, HD = 128, ZD = 16;
constexpr int XP = 528;   // 513 input features padded to a multiple of 16
constexpr int NO = 544;   // 513 output features padded to 17 row tiles of 32
constexpr int TB = 32;    // frames per tile
constexpr int NT_OUT = 17;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct PolF32 {
    typedef float T;
    typedef f32x4 Frag;
    typedef f32x4 Pack4;
    static constexpr int E = 4;        // elements per 16-byte fragment
    static constexpr int KSTEP = 8;    // reduction depth per fragment pair
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ T cvt(float v) { return v; }
    static __device__ __forceinline__ Frag ones() { return Frag{1.f, 1.f, 1.f, 1.f}; }
    static __device__ __forceinline__ float tanh_(float v) { return tanhf(v); }
    static __device__ __forceinline__ float exp_(float v) { return expf(v); }
    static __device__ __forceinline__ float log_(float v) { return logf(v); }
};

struct PolBF16 {
    typedef __bf16 T;
    typedef bf16x8 Frag;
    typedef bf16x4 Pack4;
    static constexpr int E = 8;
    static constexpr int KSTEP = 16;
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ T cvt(float v) { return (__bf16)v; }
    static __device__ __forceinline__ Frag ones() {
        const __bf16 o = (__bf16)1.0f;
        return Frag{o, o, o, o, o, o, o, o};
    }
    // throughput mode: hardware exp2/log2/rcp based transcendentals
    static __device__ __forceinline__ float exp_(float v) { return __expf(v); }
    static __device__ __forceinline__ float log_(float v) { return __logf(v); }
    static __device__ __forceinline__ float tanh_(float v) {
        const float e = __expf(2.f * v);
        return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    }
};

// LDS row strides (elements): an odd number of 16-byte slots per row
template <typename T> struct Ld {
    static constexpr int per16 = 16 / (int)sizeof(T);
    static constexpr int u = NO + per16;          // [frame][544 features]   (x / y / da)
    static constexpr int hh = HD + per16;         // [frame][128]
    static constexpr int z = 32 + per16;          // [frame][32]              (z | pad, dmu | dlv)
    static constexpr int xt = 129;                // fp32 [frame][128] slice of x for the loss epilogue
    static constexpr size_t bytes = (size_t)TB * (u + 2 * hh + z) * sizeof(T) + (size_t)TB * xt * sizeof(float) + 64;
};

__device__ __forceinline__ int feat_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// acc += W[rows of this lane][k-block] * act^T : NSTEPS fragment pairs, weights prefetched
// CH steps ahead straight from global memory, activations from LDS.
template <typename P, int NSTEPS>
__device__ __forceinline__ void gemm_block(f32x16& acc, const typename P::T* __restrict__ wrow, const typename P::T* brow) {
    typedef typename P::Frag Frag;
    constexpr int CH = 8;
    constexpr int STR = 2 * P::E;
    Frag a[CH], an[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i)
        if (i < NSTEPS) a[i] = *reinterpret_cast<const Frag*>(wrow + i * STR);
#pragma unroll
    for (int s0 = 0; s0 < NSTEPS; s0 += CH) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (s0 + CH + i < NSTEPS) an[i] = *reinterpret_cast<const Frag*>(wrow + (s0 + CH + i) * STR);
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (s0 + i < NSTEPS) {
                const Frag b = *reinterpret_cast<const Frag*>(brow + (s0 + i) * STR);
                P::mma(acc, a[i], b);
            }
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (s0 + CH + i < NSTEPS) a[i] = an[i];
    }
}

struct RowsArgs {
    const float* x; const float* y; const float* eps;
    int ldx, ldy, ydim;
    int64_t B, Bp;
    int ntiles;
    float invB, elbo_eps;
    const void *W1s, *W2s, *Wmvs, *W3s, *W4s, *W5s, *W5t, *W4t, *W3zt, *Wmvt, *W2t;
    const float *b1, *b2, *bmu, *blv, *b3, *b4, *b5;
    void *xT, *yT, *h1T, *h2T, *dh1T, *dh2T, *dmlvT, *zT, *d1T, *d2T, *dd1T, *dd2T, *daT;
    double* partials;
};

template <typename P> __device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
}

// write a 32-feature x 32-frame tile (values v[r], feature = fbase + feat_of(r,h), frame = l31)
// to LDS [frame][feature] and/or to the transposed stash [feature][Bp]
template <typename P>
__device__ __forceinline__ void put_tile(const float (&v)[16], typename P::T* lds, int ldl, int fbase,
                                         typename P::T* stash, int64_t Bp, int64_t bcol, int l31, int h) {
    typedef typename P::T T;
    typedef typename P::Pack4 Pack4;
    if (lds) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            Pack4 p;
            p[0] = P::cvt(v[4 * gq]); p[1] = P::cvt(v[4 * gq + 1]); p[2] = P::cvt(v[4 * gq + 2]); p[3] = P::cvt(v[4 * gq + 3]);
            *reinterpret_cast<Pack4*>(lds + l31 * ldl + fbase + 8 * gq + 4 * h) = p;
        }
    }
    if (stash) {
        T* s = stash + bcol + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[(int64_t)(fbase + feat_of(r, h)) * Bp] = P::cvt(v[r]);
    }
}

// global [32 frames][ncols] fp32 (row stride ld) -> LDS U[frame][col] as T, zero padded to `pcols`
template <typename P>
__device__ __forceinline__ void load_rows_to_lds(const float* __restrict__ src, int ld, int ncols, int pcols, int64_t b0, int64_t B,
                                                 typename P::T* U, int ldu, int tid) {
    const int total = TB * pcols;
    for (int idx = tid; idx < total; idx += 256) {
        const int row = idx / pcols, col = idx - row * pcols;
        float v = 0.f;
        if (col < ncols && b0 + row < B) v = src[(b0 + row) * ld + col];
        U[row * ldu + col] = P::cvt(v);
    }
}

// LDS U[frame][col] -> stash [col][Bp] (transposed), 16 bytes (E frames) per store; rows up to `srows`
// (multiple of 32) are written, columns >= pcols as zeros
template <typename P>
__device__ __forceinline__ void stash_from_lds(const typename P::T* U, int ldu, int pcols, int srows, typename P::T* stash,
                                               int64_t Bp, int64_t b0, int tid) {
    typedef typename P::Frag Frag;
    constexpr int E = P::E;
    constexpr int groups = TB / E;
    const int total = srows * groups;
    for (int idx = tid; idx < total; idx += 256) {
        const int gi = idx / srows, f = idx - gi * srows;     // consecutive threads -> consecutive features
        Frag p;
#pragma unroll
        for (int j = 0; j < E; ++j) p[j] = (f < pcols) ? U[(gi * E + j) * ldu + f] : P::cvt(0.f);
        *reinterpret_cast<Frag*>(stash + (int64_t)f * Bp + b0 + gi * E) = p;
    }
}

template <typename P, int YP, bool YENC>
__global__ __launch_bounds__(256, 1) void vae_rows_kernel(const RowsArgs g) {
    typedef typename P::T T;
    constexpr int E = P::E;
    constexpr int KS = P::KSTEP;
    constexpr int LDU = Ld<T>::u, LDH = Ld<T>::hh, LDZ = Ld<T>::z, LDX = Ld<T>::xt;
    constexpr int LD1 = XP + (YENC ? YP : 0);           // W1 shadow row length
    constexpr int LD3 = ZD + YP;                        // W3 shadow row length
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* U = reinterpret_cast<T*>(smem);
    T* Ha = U + TB * LDU;
    T* Hb = Ha + TB * LDH;
    T* Zb = Hb + TB * LDH;
    float* Xt = reinterpret_cast<float*>(Zb + TB * LDZ);
    __shared__ float red[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int fb = 32 * wave;                           // this wave's feature block in 128-wide layers
    const T* const W1s = (const T*)g.W1s; const T* const W2s = (const T*)g.W2s; const T* const Wmvs = (const T*)g.Wmvs;
    const T* const W3s = (const T*)g.W3s; const T* const W4s = (const T*)g.W4s; const T* const W5s = (const T*)g.W5s;
    const T* const W5t = (const T*)g.W5t; const T* const W4t = (const T*)g.W4t; const T* const W3zt = (const T*)g.W3zt;
    const T* const Wmvt = (const T*)g.Wmvt; const T* const W2t = (const T*)g.W2t;

    double tot_rec = 0.0, tot_kl = 0.0;

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int64_t b0 = (int64_t)tile * TB;
        const bool live = (b0 + l31) < g.B;             // this lane's frame exists
        float rec_lane = 0.f, kl_lane = 0.f;

        // ---------------- encoder layer 1: [x | y] -> h1 ----------------
        load_rows_to_lds<P>(g.x, g.ldx, XD, XP, b0, g.B, U, LDU, tid);
        __syncthreads();
        stash_from_lds<P>(U, LDU, XP, NO, (T*)g.xT, g.Bp, b0, tid);
        f32x16 acc;
        zero_acc<P>(acc);
        gemm_block<P, XP / KS>(acc, W1s + (int64_t)(fb + l31) * LD1 + h * E, U + l31 * LDU + h * E);
        if (YP > 0) {
            __syncthreads();
            load_rows_to_lds<P>(g.y, g.ldy, g.ydim, YP, b0, g.B, U, LDU, tid);
            __syncthreads();
            stash_from_lds<P>(U, LDU, YP, (YP + 31) / 32 * 32, (T*)g.yT, g.Bp, b0, tid);
            if (YENC) gemm_block<P, YP / KS>(acc, W1s + (int64_t)(fb + l31) * LD1 + XP + h * E, U + l31 * LDU + h * E);
        }
        float h1r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h1r[r] = P::tanh_(acc[r] + g.b1[fb + feat_of(r, h)]);
        put_tile<P>(h1r, Ha, LDH, fb, (T*)g.h1T, g.Bp, b0, l31, h);
        __syncthreads();

        // ---------------- encoder layer 2 ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, W2s + (int64_t)(fb + l31) * HD + h * E, Ha + l31 * LDH + h * E);
        float h2r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h2r[r] = P::tanh_(acc[r] + g.b2[fb + feat_of(r, h)]);
        put_tile<P>(h2r, Hb, LDH, fb, (T*)g.h2T, g.Bp, b0, l31, h);
        __syncthreads();

        // ---------------- heads + reparametrisation (wave 0): rows 0-15 mu, 16-31 log_var ----------------
        float mu_r[8], lv_r[8], ep_r[8], sd_r[8];
        if (wave == 0) {
            zero_acc<P>(acc);
            gemm_block<P, HD / KS>(acc, Wmvs + (int64_t)l31 * HD + h * E, Hb + l31 * LDH + h * E);
            float zv[16];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int j = feat_of(r, h);                       // latent index 0..15
                mu_r[r] = acc[r] + g.bmu[j];
                lv_r[r] = acc[r + 8] + g.blv[j];
                ep_r[r] = live ? g.eps[(b0 + l31) * ZD + j] : 0.f;
                sd_r[r] = P::exp_(0.5f * lv_r[r]);                 // models.py:17
                zv[r] = fmaf(sd_r[r], ep_r[r], mu_r[r]);           // models.py:20
                zv[r + 8] = 0.f;
                if (live) kl_lane += lv_r[r] - mu_r[r] * mu_r[r] - P::exp_(lv_r[r]);   // utils.py:75
            }
            // z block of the decoder input: features 0..15 valid, 16..31 zero
            put_tile<P>(zv, Zb, LDZ, 0, nullptr, g.Bp, b0, l31, h);
            T* zs = (T*)g.zT + b0 + l31;
#pragma unroll
            for (int r = 0; r < 8; ++r) zs[(int64_t)feat_of(r, h) * g.Bp] = P::cvt(zv[r]);
        }
        __syncthreads();

        // ---------------- decoder layer 1: [z | y] -> d1 ----------------
        zero_acc<P>(acc);
        gemm_block<P, ZD / KS>(acc, W3s + (int64_t)(fb + l31) * LD3 + h * E, Zb + l31 * LDZ + h * E);
        if (YP > 0) gemm_block<P, YP / KS>(acc, W3s + (int64_t)(fb + l31) * LD3 + ZD + h * E, U + l31 * LDU + h * E);
        float d1r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) d1r[r] = P::tanh_(acc[r] + g.b3[fb + feat_of(r, h)]);
        put_tile<P>(d1r, Ha, LDH, fb, (T*)g.d1T, g.Bp, b0, l31, h);
        __syncthreads();

        // ---------------- decoder layer 2 ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, W4s + (int64_t)(fb + l31) * HD + h * E, Ha + l31 * LDH + h * E);
        float d2r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) d2r[r] = P::tanh_(acc[r] + g.b4[fb + feat_of(r, h)]);
        put_tile<P>(d2r, Hb, LDH, fb, (T*)g.d2T, g.Bp, b0, l31, h);
        __syncthreads();

        // ---------------- output layer a = W5 d2 + b5, Itakura-Saito terms, da -> U ----------------
        for (int it = 0; it < (NT_OUT + 3) / 4; ++it) {
            const int f0 = 128 * it;
            for (int idx = tid; idx < TB * 128; idx += 256) {      // x[32 frames][f0 .. f0+127] -> Xt
                const int row = idx >> 7, col = idx & 127;
                float v = 0.f;
                if (f0 + col < XD && b0 + row < g.B) v = g.x[(b0 + row) * g.ldx + f0 + col];
                Xt[row * LDX + col] = v;
            }
            __syncthreads();
            const int t = 4 * it + wave;
            if (t < NT_OUT) {
                zero_acc<P>(acc);
                gemm_block<P, HD / KS>(acc, W5s + (int64_t)(32 * t + l31) * HD + h * E, Hb + l31 * LDH + h * E);
                float da[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fl = feat_of(r, h);
                    const int f = 32 * t + fl;
                    const bool ok = live && f < XD;
                    const float a = acc[r] + (f < XD ? g.b5[f] : 0.f);
                    const float xv = Xt[l31 * LDX + 32 * wave + fl];
                    const float xe = xv * P::exp_(-a);               // x / r,  r = exp(a)  (models.py:122)
                    if (ok) rec_lane += xe - P::log_(xv + g.elbo_eps) + a - 1.f;   // utils.py:74 (log r = a)
                    da[r] = ok ? (1.f - xe) * g.invB : 0.f;          // d recon / d a
                }
                put_tile<P>(da, U, LDU, 32 * t, (T*)g.daT, g.Bp, b0, l31, h);
            }
            __syncthreads();
        }

        // ---------------- backward: d2 <- da ----------------
        zero_acc<P>(acc);
        gemm_block<P, NO / KS>(acc, W5t + (int64_t)(fb + l31) * NO + h * E, U + l31 * LDU + h * E);
        float dv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - d2r[r] * d2r[r]);
        put_tile<P>(dv, Ha, LDH, fb, (T*)g.dd2T, g.Bp, b0, l31, h);
        __syncthreads();

        // ---------------- backward: d1 <- dpre_d2 ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, W4t + (int64_t)(fb + l31) * HD + h * E, Ha + l31 * LDH + h * E);
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - d1r[r] * d1r[r]);
        put_tile<P>(dv, Hb, LDH, fb, (T*)g.dd1T, g.Bp, b0, l31, h);
        __syncthreads();

        // ---------------- backward: z <- dpre_d1 (wave 0), then dmu / dlogvar ----------------
        if (wave == 0) {
            zero_acc<P>(acc);
            gemm_block<P, HD / KS>(acc, W3zt + (int64_t)l31 * HD + h * E, Hb + l31 * LDH + h * E);
            float dml[16];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float dz = acc[r];
                dml[r] = live ? dz + mu_r[r] * g.invB : 0.f;                                                   // dmu
                dml[r + 8] = live ? dz * ep_r[r] * (0.5f * sd_r[r]) - 0.5f * g.invB * (1.f - P::exp_(lv_r[r])) : 0.f;   // dlogvar
            }
            put_tile<P>(dml, Zb, LDZ, 0, (T*)g.dmlvT, g.Bp, b0, l31, h);
        }
        __syncthreads();

        // ---------------- backward: h2 <- [dmu | dlogvar] ----------------
        zero_acc<P>(acc);
        gemm_block<P, 32 / KS>(acc, Wmvt + (int64_t)(fb + l31) * 32 + h * E, Zb + l31 * LDZ + h * E);
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - h2r[r] * h2r[r]);
        put_tile<P>(dv, Ha, LDH, fb, (T*)g.dh2T, g.Bp, b0, l31, h);
        __syncthreads();

        // ---------------- backward: h1 <- dpre_h2 (inputs are data: stop here) ----------------
        zero_acc<P>(acc);
        gemm_block<P, HD / KS>(acc, W2t + (int64_t)(fb + l31) * HD + h * E, Ha + l31 * LDH + h * E);
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = acc[r] * (1.f - h1r[r] * h1r[r]);
        put_tile<P>(dv, nullptr, 0, fb, (T*)g.dh1T, g.Bp, b0, l31, h);

        // ---------------- per-tile loss sums ----------------
        const float rs = wave_sum(rec_lane), ks = wave_sum(kl_lane);
        if (lane == 0) { red[wave] = rs; red[4 + wave] = ks; }
        __syncthreads();
        if (tid == 0) {
            tot_rec += (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
            tot_kl += -0.5 * (double)red[4];
        }
        __syncthreads();
    }
    if (tid == 0) {
        g.partials[2 * blockIdx.x] = tot_rec;
        g.partials[2 * blockIdx.x + 1] = tot_kl;
    }
}

// ---------------------------------------------------------------------------------------------
struct TileDesc {
    const void* A;       // stash rows of dPre^T for this tile's 32 output features
    const void* Bm;      // stash rows of In^T for this tile's 32 input features
    int64_t out_off;     // float offset of element (row 0, col 0) of this tile in a gradient slab
    int64_t bias_off;    // float offset of the bias gradient rows, -1 = none
    int32_t ldo, mvalid, nvalid, pad;
};

template <typename P>
__global__ __launch_bounds__(256) void wgrad_kernel(const TileDesc* __restrict__ tiles, int ntiles, int64_t Bp, int64_t kper,
                                                    float* __restrict__ slabs, int64_t slab_stride) {
    typedef typename P::T T;
    typedef typename P::Frag Frag;
    constexpr int E = P::E, KS = P::KSTEP, CH = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int ti = blockIdx.x * 4 + wave;
    if (ti >= ntiles) return;
    const TileDesc d = tiles[ti];
    const int64_t kbeg = (int64_t)blockIdx.y * kper;
    int64_t kend = kbeg + kper;
    if (kend > Bp) kend = Bp;
    const T* arow = (const T*)d.A + (int64_t)l31 * Bp + h * E;
    const T* brow = (const T*)d.Bm + (int64_t)l31 * Bp + h * E;
    const bool bias = d.bias_off >= 0;
    f32x16 acc, accb;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accb[i] = 0.f; }
    const Frag one = P::ones();
    Frag a[CH], b[CH], an[CH], bn[CH];
    if (kbeg < kend) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            a[i] = *reinterpret_cast<const Frag*>(arow + kbeg + i * KS);
            b[i] = *reinterpret_cast<const Frag*>(brow + kbeg + i * KS);
        }
    }
    for (int64_t k = kbeg; k < kend; k += CH * KS) {
        int64_t kn = k + CH * KS;
        if (kn >= kend) kn = k;                 // last pass: harmless reload instead of a branch around the loads
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            an[i] = *reinterpret_cast<const Frag*>(arow + kn + i * KS);
            bn[i] = *reinterpret_cast<const Frag*>(brow + kn + i * KS);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            P::mma(acc, a[i], b[i]);
            if (bias) P::mma(accb, a[i], one);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) { a[i] = an[i]; b[i] = bn[i]; }
    }
    float* slab = slabs + (int64_t)blockIdx.y * slab_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = feat_of(r, h);
        if (row < d.mvalid) {
            if (l31 < d.nvalid) slab[d.out_off + (int64_t)row * d.ldo + l31] = acc[r];
            if (bias && l31 == 0) slab[d.bias_off + row] = accb[r];
        }
    }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(float* __restrict__ slabs, int64_t n, int nslabs, int64_t stride) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = slabs[i];
        for (int k = 1; k < nslabs; ++k) s += slabs[k * stride + i];
        slabs[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
struct TensorDesc {
    int64_t off;          // float offset in the flat parameter buffer
    int32_t rows, cols;
    int64_t sf_off;       // forward copy: element offset in the weight-copy buffer, -1 = none
    int32_t sf_ld, sf_split, sf_gap, pad0;   // column c lands at c (c < split) or c + gap
    int64_t st_off;       // transposed copy (for backward-data), -1 = none
    int32_t st_ld, st_roff, st_cmax, pad1;   // element (r, c < cmax) lands at [c][r + roff]
};

struct ApplyArgs {
    float* p; float* m; float* v;
    const float* slabs; int64_t slab_stride; int nslabs;
    const TensorDesc* tensors; int ntensors;
    void* wcopy;
    float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale;
    const double* partials; int npartials; int64_t B; float* losses3;
};

template <typename T, bool ADAM>
__global__ __launch_bounds__(256) void apply_kernel(const ApplyArgs g) {
    const int t = blockIdx.y;
    if (t == g.ntensors) {                    // loss finalisation block
        if (!ADAM || blockIdx.x != 0 || g.losses3 == nullptr) return;
        __shared__ double red[4][2];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double a = 0.0, k = 0.0;
        for (int i = threadIdx.x; i < g.npartials; i += 256) { a += g.partials[2 * i]; k += g.partials[2 * i + 1]; }
        a = wave_sum(a); k = wave_sum(k);
        if (lane == 0) { red[wave][0] = a; red[wave][1] = k; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float recon = (float)((red[0][0] + red[1][0] + red[2][0] + red[3][0]) / (double)g.B);
            const float kl = (float)((red[0][1] + red[1][1] + red[2][1] + red[3][1]) / (double)g.B);
            g.losses3[0] = recon + kl; g.losses3[1] = recon; g.losses3[2] = kl;
        }
        return;
    }
    const TensorDesc d = g.tensors[t];
    const int64_t n = (int64_t)d.rows * d.cols;
    T* wc = (T*)g.wcopy;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = d.off + i;
        float pi = g.p[idx];
        if (ADAM) {
            float gi = g.slabs[idx];
            for (int k = 1; k < g.nslabs; ++k) gi += g.slabs[k * g.slab_stride + idx];
            gi *= g.gscale;
            const float mi = g.m[idx] + g.one_minus_b1 * (gi - g.m[idx]);
            const float vi = g.v[idx] * g.b2 + g.one_minus_b2 * (gi * gi);
            const float denom = sqrtf(vi) / g.bc2_sqrt + g.eps;
            pi = pi - g.step_size * (mi / denom);
            g.p[idx] = pi; g.m[idx] = mi; g.v[idx] = vi;
        }
        const int r = (int)(i / d.cols), c = (int)(i - (int64_t)r * d.cols);
        if (d.sf_off >= 0) wc[d.sf_off + (int64_t)r * d.sf_ld + (c < d.sf_split ? c : c + d.sf_gap)] = (T)pi;
        if (d.st_off >= 0 && c < d.st_cmax) wc[d.st_off + (int64_t)c * d.st_ld + r + d.st_roff] = (T)pi;
    }
}

// ---------------------------------------------------------------------------------------------
// host-side planning
struct Layout {
    // weight-copy buffer (elements of T)
    int64_t W1s, W2s, Wmvs, W3s, W4s, W5s, W5t, W4t, W3zt, Wmvt, W2t, wcopy_elems;
    int ld1, ld3, yp, ye, yd;
    // stash (rows of Bp elements)
    int64_t xT, yT, h1T, h2T, dh1T, dh2T, dmlvT, zT, d1T, d2T, dd1T, dd2T, daT, stash_rows;
    // workspace byte offsets
    int64_t o_tiles, o_tensors, o_partials, o_wcopy, o_stash, o_grads, total;
    int ntiles;
};

static inline int64_t al(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static int make_layout(const dvae_train_plan_t& p, Layout& L) {
    const int esz = p.precision == DVAE_PREC_BF16 ? 2 : 4;
    L.yp = p.y_dim == 0 ? 0 : (p.y_dim + 15) / 16 * 16;
    L.ye = p.model == DVAE_MODEL_M2 ? L.yp : 0;
    L.yd = L.yp;
    L.ld1 = XP + L.ye;
    L.ld3 = ZD + L.yd;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += al(n, 128); return r; };
    L.W1s = take((int64_t)HD * L.ld1); L.W2s = take(HD * HD); L.Wmvs = take(32 * HD); L.W3s = take((int64_t)HD * L.ld3);
    L.W4s = take(HD * HD); L.W5s = take((int64_t)NO * HD); L.W5t = take((int64_t)HD * NO); L.W4t = take(HD * HD);
    L.W3zt = take(32 * HD); L.Wmvt = take(HD * 32); L.W2t = take(HD * HD);
    L.wcopy_elems = o;
    int64_t r = 0;
    auto rows = [&](int64_t n) { int64_t q = r; r += n; return q; };
    L.xT = rows(NO); L.yT = rows(L.yp ? al(L.yp, 32) : 0); L.h1T = rows(HD); L.h2T = rows(HD); L.dh1T = rows(HD); L.dh2T = rows(HD);
    L.dmlvT = rows(32); L.zT = rows(32); L.d1T = rows(HD); L.d2T = rows(HD); L.dd1T = rows(HD); L.dd2T = rows(HD); L.daT = rows(NO);
    rows(32);   // slack: the second head job reads 16 rows past dmlvT's 32 (masked on store)
    L.stash_rows = r;
    // tiles: L1x 4x17, L1y 4x(ye/32), L2 4x4, heads 2x4, L3z 4x1, L3y 4x(yd/32), L4 4x4, L5 17x4
    const int nty = L.yp ? (int)(al(L.yp, 32) / 32) : 0;
    L.ntiles = 4 * NT_OUT + (L.ye ? 4 * nty : 0) + 16 + 8 + 4 + (L.yd ? 4 * nty : 0) + 16 + NT_OUT * 4;
    int64_t b = 0;
    auto bytes = [&](int64_t n) { int64_t q = b; b += al(n, 256); return q; };
    L.o_tiles = bytes((int64_t)L.ntiles * sizeof(TileDesc));
    L.o_tensors = bytes(DVAE_TRAIN_MAX_TENSORS * sizeof(TensorDesc));
    L.o_partials = bytes(p.rows_grid * 2 * sizeof(double));
    L.o_wcopy = bytes(L.wcopy_elems * esz);
    L.o_stash = bytes(L.stash_rows * p.Bp * esz);
    L.o_grads = bytes((int64_t)p.ksplit * p.n_params * sizeof(float));
    L.total = b;
    return 0;
}

static bool g_prof = false;
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

extern "C" int dvae_train_plan(int model, int y_dim, int precision, int64_t B, int ksplit_hint, dvae_train_plan_t* plan) {
    DVAE_CHECK_ARG(plan != nullptr && B > 0, "train_plan: bad argument");
    if (!((model == DVAE_MODEL_M1 && y_dim == 0) || (model == DVAE_MODEL_M2 && (y_dim == 1 || y_dim == 513)))) {
        set_error("train_plan: fused kernels cover M1 (y 0) and M2 (y 1 or 513) at x 513 / h [128,128] / z 16; got model %d y_dim %d", model, y_dim);
        return DVAE_E_UNSUPPORTED;
    }
    DVAE_CHECK_ARG(precision == DVAE_PREC_F32 || precision == DVAE_PREC_BF16, "train_plan: unknown precision %d", precision);
    memset(plan, 0, sizeof(*plan));
    plan->model = model; plan->y_dim = y_dim; plan->precision = precision; plan->B = B;
    plan->Bp = al(B, 128);
    const int ye = model == DVAE_MODEL_M2 ? y_dim : 0, yd = y_dim;
    const int rows[14] = {HD, HD, HD, HD, ZD, ZD, ZD, ZD, HD, HD, HD, HD, XD, XD};
    const int cols[14] = {XD + ye, 1, HD, 1, HD, 1, HD, 1, ZD + yd, 1, HD, 1, HD, 1};
    int64_t off = 0;
    plan->n_tensors = 14;
    for (int i = 0; i < 14; ++i) {
        plan->tensor_offset[i] = off; plan->tensor_rows[i] = rows[i]; plan->tensor_cols[i] = cols[i];
        off += al((int64_t)rows[i] * cols[i], 64);
    }
    plan->n_params = off;
    const int64_t ntiles = (B + TB - 1) / TB;
    const int64_t maxg = 256 * (precision == DVAE_PREC_BF16 ? 2 : 1);
    plan->rows_grid = ntiles < maxg ? ntiles : maxg;
    int ks = ksplit_hint;
    if (ks <= 0) { ks = (int)(plan->Bp / 1024); if (ks < 1) ks = 1; if (ks > 8) ks = 8; }
    if (ks > 64) ks = 64;
    plan->ksplit = ks;
    Layout L;
    make_layout(*plan, L);
    plan->workspace_bytes = L.total;
    plan->grad_offset_bytes = L.o_grads;
    const double mac = (double)HD * (XD + ye) + HD * HD + 2.0 * ZD * HD + (double)HD * (ZD + yd) + HD * HD + (double)XD * HD;
    const double dxm = (double)HD * HD + 2.0 * ZD * HD + (double)ZD * HD + HD * HD + (double)XD * HD;
    plan->flops_per_step = 2.0 * (2.0 * mac + dxm) * (double)B;
    plan->min_hbm_bytes_per_step = 4.0 * (XD + y_dim + ZD) * (double)B;
    return 0;
}

static int64_t kper_of(const dvae_train_plan_t* p) {
    const int ks = p->precision == DVAE_PREC_BF16 ? 16 : 8;
    const int64_t unit = 4 * ks;
    return al((p->Bp + p->ksplit - 1) / p->ksplit, unit);
}

template <typename T>
static void fill_tables(const dvae_train_plan_t* p, const Layout& L, char* ws_dev, TileDesc* tiles, TensorDesc* td) {
    const int64_t Bp = p->Bp;
    T* stash = (T*)(ws_dev + L.o_stash);
    auto S = [&](int64_t row) { return (const void*)(stash + row * Bp); };
    int n = 0;
    auto job = [&](int64_t Arow, int M, int64_t Brow, int N, int tensor, int col0, int bias_tensor) {
        const int ldo = p->tensor_cols[tensor];
        for (int m0 = 0; m0 < M; m0 += 32)
            for (int n0 = 0; n0 < N; n0 += 32) {
                TileDesc d;
                d.A = S(Arow + m0); d.Bm = S(Brow + n0);
                d.out_off = p->tensor_offset[tensor] + (int64_t)m0 * ldo + col0 + n0;
                d.bias_off = (bias_tensor >= 0 && n0 == 0) ? p->tensor_offset[bias_tensor] + m0 : -1;
                d.ldo = ldo; d.mvalid = M - m0 < 32 ? M - m0 : 32; d.nvalid = N - n0 < 32 ? N - n0 : 32; d.pad = 0;
                tiles[n++] = d;
            }
    };
    const int ye = p->model == DVAE_MODEL_M2 ? p->y_dim : 0, yd = p->y_dim;
    job(L.dh1T, HD, L.xT, XD, 0, 0, 1);
    if (ye) job(L.dh1T, HD, L.yT, ye, 0, XD, -1);
    job(L.dh2T, HD, L.h1T, HD, 2, 0, 3);
    job(L.dmlvT, ZD, L.h2T, HD, 4, 0, 5);
    job(L.dmlvT + ZD, ZD, L.h2T, HD, 6, 0, 7);
    job(L.dd1T, HD, L.zT, ZD, 8, 0, 9);
    if (yd) job(L.dd1T, HD, L.yT, yd, 8, ZD, -1);
    job(L.dd2T, HD, L.d1T, HD, 10, 0, 11);
    job(L.daT, XD, L.d2T, HD, 12, 0, 13);
    // tensors -> kernel-layout copies
    for (int i = 0; i < 14; ++i) {
        TensorDesc t;
        memset(&t, 0, sizeof(t));
        t.off = p->tensor_offset[i]; t.rows = p->tensor_rows[i]; t.cols = p->tensor_cols[i];
        t.sf_off = -1; t.st_off = -1; t.sf_split = 1 << 30;
        td[i] = t;
    }
    td[0].sf_off = L.W1s; td[0].sf_ld = L.ld1; td[0].sf_split = XD; td[0].sf_gap = XP - XD;
    td[2].sf_off = L.W2s; td[2].sf_ld = HD; td[2].st_off = L.W2t; td[2].st_ld = HD; td[2].st_cmax = HD;
    td[4].sf_off = L.Wmvs; td[4].sf_ld = HD; td[4].st_off = L.Wmvt; td[4].st_ld = 32; td[4].st_cmax = HD;
    td[6].sf_off = L.Wmvs + 16 * HD; td[6].sf_ld = HD; td[6].st_off = L.Wmvt; td[6].st_ld = 32; td[6].st_roff = 16; td[6].st_cmax = HD;
    td[8].sf_off = L.W3s; td[8].sf_ld = L.ld3; td[8].st_off = L.W3zt; td[8].st_ld = HD; td[8].st_cmax = ZD;
    td[10].sf_off = L.W4s; td[10].sf_ld = HD; td[10].st_off = L.W4t; td[10].st_ld = HD; td[10].st_cmax = HD;
    td[12].sf_off = L.W5s; td[12].sf_ld = HD; td[12].st_off = L.W5t; td[12].st_ld = NO; td[12].st_cmax = HD;
}

static int launch_apply(const dvae_train_plan_t* plan, const Layout& L, float* params, float* m, float* v, char* ws, int n_slabs,
                        bool adam, int step, double lr, double beta1, double beta2, double adam_eps, double grad_scale,
                        float* losses3, hipStream_t s) {
    ApplyArgs a;
    memset(&a, 0, sizeof(a));
    a.p = params; a.m = m; a.v = v;
    a.slabs = (const float*)(ws + L.o_grads); a.slab_stride = plan->n_params; a.nslabs = n_slabs;
    a.tensors = (const TensorDesc*)(ws + L.o_tensors); a.ntensors = plan->n_tensors;
    a.wcopy = ws + L.o_wcopy;
    if (adam) {
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        a.one_minus_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_minus_b2 = (float)(1.0 - beta2);
        a.step_size = (float)(lr / bc1); a.bc2_sqrt = (float)sqrt(bc2); a.eps = (float)adam_eps; a.gscale = (float)grad_scale;
    }
    a.partials = (const double*)(ws + L.o_partials); a.npartials = (int)plan->rows_grid; a.B = plan->B; a.losses3 = losses3;
    const dim3 grid(64, plan->n_tensors + 1);
    if (plan->precision == DVAE_PREC_BF16) {
        if (adam) hipLaunchKernelGGL((apply_kernel<__bf16, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((apply_kernel<__bf16, false>), grid, dim3(256), 0, s, a);
    } else {
        if (adam) hipLaunchKernelGGL((apply_kernel<float, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((apply_kernel<float, false>), grid, dim3(256), 0, s, a);
    }
    DVAE_LAUNCH_OK("apply_kernel");
    return 0;
}

extern "C" int dvae_train_repack(const dvae_train_plan_t* plan, const float* params, void* ws, void* stream) {
    DVAE_CHECK_ARG(plan && params && ws, "train_repack: bad argument");
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
    TileDesc* tiles = new TileDesc[L.ntiles + 8];
    TensorDesc td[DVAE_TRAIN_MAX_TENSORS];
    memset(td, 0, sizeof(td));
    if (plan->precision == DVAE_PREC_BF16) fill_tables<__bf16>(plan, L, w, tiles, td);
    else fill_tables<float>(plan, L, w, tiles, td);
    hipError_t e1 = hipMemcpyAsync(w + L.o_tiles, tiles, (size_t)L.ntiles * sizeof(TileDesc), hipMemcpyHostToDevice, s);
    hipError_t e2 = hipMemcpyAsync(w + L.o_tensors, td, sizeof(td), hipMemcpyHostToDevice, s);
    hipError_t e3 = hipStreamSynchronize(s);
    delete[] tiles;
    DVAE_HIP(e1); DVAE_HIP(e2); DVAE_HIP(e3);
    return dvae_train_repack(plan, params, ws, stream);
}

template <typename P, int YP, bool YENC>
static int launch_rows(const RowsArgs& a, int grid, hipStream_t s) {
    const size_t lds = Ld<typename P::T>::bytes;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)vae_rows_kernel<P, YP, YENC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute(rows kernel, %zu B LDS): %s", lds, hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    hipLaunchKernelGGL((vae_rows_kernel<P, YP, YENC>), dim3(grid), dim3(256), lds, s, a);
    DVAE_LAUNCH_OK("vae_rows_kernel");
    return 0;
}

extern "C" int dvae_train_grads(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                                const float* y, int ldy, const float* eps_noise, float elbo_eps, int reduce_slabs, void* stream) {
    DVAE_CHECK_ARG(plan && params && ws && x && eps_noise && ldx >= XD, "train_grads: bad argument");
    DVAE_CHECK_ARG(plan->y_dim == 0 || (y != nullptr && ldy >= plan->y_dim), "train_grads: y missing or ldy < y_dim");
    Layout L;
    make_layout(*plan, L);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)ws;
    const bool bf = plan->precision == DVAE_PREC_BF16;
    const int esz = bf ? 2 : 4;
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.eps = eps_noise; a.ldx = ldx; a.ldy = plan->y_dim ? ldy : 0; a.ydim = plan->y_dim;
    a.B = plan->B; a.Bp = plan->Bp; a.ntiles = (int)((plan->B + TB - 1) / TB);
    a.invB = (float)(1.0 / (double)plan->B); a.elbo_eps = elbo_eps;
    char* wc = w + L.o_wcopy;
    auto WC = [&](int64_t off) { return (const void*)(wc + off * esz); };
    a.W1s = WC(L.W1s); a.W2s = WC(L.W2s); a.Wmvs = WC(L.Wmvs); a.W3s = WC(L.W3s); a.W4s = WC(L.W4s); a.W5s = WC(L.W5s);
    a.W5t = WC(L.W5t); a.W4t = WC(L.W4t); a.W3zt = WC(L.W3zt); a.Wmvt = WC(L.Wmvt); a.W2t = WC(L.W2t);
    a.b1 = params + plan->tensor_offset[1]; a.b2 = params + plan->tensor_offset[3];
    a.bmu = params + plan->tensor_offset[5]; a.blv = params + plan->tensor_offset[7];
    a.b3 = params + plan->tensor_offset[9]; a.b4 = params + plan->tensor_offset[11]; a.b5 = params + plan->tensor_offset[13];
    a.partials = (double*)(w + L.o_partials);
    char* st = w + L.o_stash;
    auto ST = [&](int64_t row) { return (void*)(st + row * plan->Bp * esz); };
    a.xT = ST(L.xT); a.yT = ST(L.yT); a.h1T = ST(L.h1T); a.h2T = ST(L.h2T); a.dh1T = ST(L.dh1T); a.dh2T = ST(L.dh2T);
    a.dmlvT = ST(L.dmlvT); a.zT = ST(L.zT); a.d1T = ST(L.d1T); a.d2T = ST(L.d2T); a.dd1T = ST(L.dd1T); a.dd2T = ST(L.dd2T); a.daT = ST(L.daT);
    const int grid = (int)plan->rows_grid;
    const bool m2 = plan->model == DVAE_MODEL_M2;
    int rc;
    {
        ProfScope ps(s, 0);
        if (bf) {
            if (!m2) rc = launch_rows<PolBF16, 0, false>(a, grid, s);
            else if (plan->y_dim == 1) rc = launch_rows<PolBF16, 16, true>(a, grid, s);
            else rc = launch_rows<PolBF16, 528, true>(a, grid, s);
        } else {
            if (!m2) rc = launch_rows<PolF32, 0, false>(a, grid, s);
            else if (plan->y_dim == 1) rc = launch_rows<PolF32, 16, true>(a, grid, s);
            else rc = launch_rows<PolF32, 528, true>(a, grid, s);
        }
    }
    if (rc) return rc;
    const int64_t kper = kper_of(plan);
    const int ks = (int)((plan->Bp + kper - 1) / kper);
    DVAE_CHECK_ARG(ks <= plan->ksplit, "train_grads: internal k-split mismatch");
    const dim3 g2((unsigned)((L.ntiles + 3) / 4), (unsigned)ks);
    float* slabs = (float*)(w + L.o_grads);
    {
        ProfScope ps(s, 1);
        if (bf) hipLaunchKernelGGL((wgrad_kernel<PolBF16>), g2, dim3(256), 0, s, (const TileDesc*)(w + L.o_tiles), L.ntiles, plan->Bp, kper, slabs, plan->n_params);
        else hipLaunchKernelGGL((wgrad_kernel<PolF32>), g2, dim3(256), 0, s, (const TileDesc*)(w + L.o_tiles), L.ntiles, plan->Bp, kper, slabs, plan->n_params);
    }
    DVAE_LAUNCH_OK("wgrad_kernel");
    if (reduce_slabs && ks > 1) {
        ProfScope ps(s, 2);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(512), dim3(256), 0, s, slabs, plan->n_params, ks, plan->n_params);
        DVAE_LAUNCH_OK("slab_reduce_kernel");
    }
    return 0;
}

static int used_slabs(const dvae_train_plan_t* plan) {
    const int64_t kper = kper_of(plan);
    return (int)((plan->Bp + kper - 1) / kper);
}

extern "C" int dvae_train_apply(const dvae_train_plan_t* plan, float* params, float* m, float* v, void* ws, int n_slabs,
                                int step, double lr, double beta1, double beta2, double adam_eps, double grad_scale,
                                float* losses3, void* stream) {
    DVAE_CHECK_ARG(plan && params && m && v && ws && step >= 1, "train_apply: bad argument");
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
    int rc = dvae_train_grads(plan, params, ws, x, ldx, y, ldy, eps_noise, elbo_eps, 0, stream);
    if (rc) return rc;
    return dvae_train_apply(plan, params, m, v, ws, 0, step, lr, beta1, beta2, adam_eps, 1.0, losses3, stream);
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
