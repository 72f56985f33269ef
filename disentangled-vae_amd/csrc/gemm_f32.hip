// Layer-level fp32 kernels: fused Linear+activation forward, its two backward
// GEMMs and the activation backward.  One LDS-tiled kernel template on the
// exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32
// fma chain, so results sit at fp32 rounding distance from the reference's
// CPU sgemm).  64x64 output tile per 256-thread workgroup (4 waves, 2x2, one
// 32x32 accumulator each), K consumed in 32-deep slabs staged through LDS with
// a register prefetch of the next slab.
//
// Operand layouts (row-major global memory, no transposes materialised):
//   forward     out[B,N]  = act([x0|x1] W^T + b)   A=[M][K]      B=W [N][K]
//   bwd-data    din[B,K]  = dpre W[:,koff:+K]      A=[M][Kred]   B=[Kred][N]   (k-major B)
//   bwd-weight  dW[N,Kin] = dpre^T [x0|x1]         A=[Kred][M]   B=[Kred][N]   (both k-major)
// The 32x32x2 f32 MFMA wants lane l to hold A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]:
// a k-major LDS slab serves that with unit-stride ds_read_b32; an [M][K] slab
// is padded to 33 floats per row so the 32 rows land on 32 distinct banks.
#include "common.hpp"

namespace dvae {

// Row-major matrix made of two column blocks: implements torch.cat([x, y], dim=1)
// (packages/models/models.py:201-202) without materialising the concat.
struct MatCat {
    const float* p0;
    const float* p1;
    int c0, c1;
    int ld0, ld1;
    int64_t rows;
    __device__ __forceinline__ float get(int64_t r, int c) const {
        if (r >= rows) return 0.f;
        if (c < c0) return p0[r * ld0 + c];
        c -= c0;
        if (c < c1) return p1[r * ld1 + c];
        return 0.f;
    }
};

struct GemmArgs {
    MatCat A, B;
    int64_t M;       // output rows
    int N;           // output cols
    int64_t K;       // reduction length
    int64_t kper;    // reduction rows per grid.z slice (multiple of 32)
    float* out;      // output matrix
    int ldo;
    const float* bias;   // EPI 0
    int act;             // EPI 0
    int accumulate;      // EPI 1
    float* db;           // EPI 2 (may be null)
    int atomic;          // EPI 2: combine grid.z slices with atomics
    float* part;         // EPI 2, deterministic form: slice z writes its partial tile to part + z * part_stride (same [M][ldo] layout), db partials behind the matrices
    int64_t part_stride;
    float* dbpart;       // [slices][M]
};

constexpr int BM = 64, BN = 64, BK = 32;

template <bool KM>
__device__ __forceinline__ void tile_load(const MatCat& X, int64_t mn0, int64_t kc, int64_t kend, int tid, float (&r)[8]) {
    if (KM) {  // X is [K][MN]: 32 k-rows x 64 columns
        const int c = tid & 63, rr = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t k = kc + rr + 4 * i;
            r[i] = (k < kend) ? X.get(k, (int)(mn0 + c)) : 0.f;
        }
    } else {   // X is [MN][K]: 64 rows x 32 k-columns
        const int c = tid & 31, rr = tid >> 5;
        const int64_t k = kc + c;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            r[i] = (k < kend) ? X.get(mn0 + rr + 8 * i, (int)k) : 0.f;
        }
    }
}

template <bool KM>
__device__ __forceinline__ void tile_store(float* S, int tid, const float (&r)[8]) {
    if (KM) {
        const int c = tid & 63, rr = tid >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) S[(rr + 4 * i) * BM + c] = r[i];
    } else {
        const int c = tid & 31, rr = tid >> 5;
#pragma unroll
        for (int i = 0; i < 8; ++i) S[(rr + 8 * i) * (BK + 1) + c] = r[i];
    }
}

template <bool AKM, bool BKM, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ float As[AKM ? BK * BM : BM * (BK + 1)];
    __shared__ float Bs[BKM ? BK * BN : BN * (BK + 1)];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int64_t n0 = (int64_t)blockIdx.y * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.kper;
    const int64_t kend = (kbeg + g.kper < g.K) ? kbeg + g.kper : g.K;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float dbacc = 0.f;

    float ra[8], rb[8];
    if (kbeg < kend) {
        tile_load<AKM>(g.A, m0, kbeg, kend, tid, ra);
        tile_load<BKM>(g.B, n0, kbeg, kend, tid, rb);
    }
    for (int64_t kc = kbeg; kc < kend; kc += BK) {
        tile_store<AKM>(As, tid, ra);
        tile_store<BKM>(Bs, tid, rb);
        __syncthreads();
        if (kc + BK < kend) {
            tile_load<AKM>(g.A, m0, kc + BK, kend, tid, ra);
            tile_load<BKM>(g.B, n0, kc + BK, kend, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + h;
            const float a = AKM ? As[k * BM + wm * 32 + l31] : As[(wm * 32 + l31) * (BK + 1) + k];
            const float b = BKM ? Bs[k * BN + wn * 32 + l31] : Bs[(wn * 32 + l31) * (BK + 1) + k];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (EPI == 2 && AKM) {
            if (g.db != nullptr && blockIdx.y == 0 && tid < BM) {
#pragma unroll 8
                for (int k = 0; k < BK; ++k) dbacc += As[k * BM + tid];
            }
        }
        __syncthreads();
    }

    // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int64_t col = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < g.M && col < g.N) {
            float v = acc[r];
            float* dst = (EPI == 2 && g.part != nullptr ? g.part + (int64_t)blockIdx.z * g.part_stride : g.out) + row * g.ldo + col;
            if (EPI == 0) {
                if (g.bias) v += g.bias[col];
                *dst = act_apply(v, g.act);
            } else if (EPI == 1) {
                if (g.accumulate) v += *dst;
                *dst = v;
            } else {
                if (g.atomic && g.part == nullptr) atomicAdd(dst, v); else *dst = v;
            }
        }
    }
    if (EPI == 2 && AKM) {
        if (g.db != nullptr && blockIdx.y == 0 && tid < BM && m0 + tid < g.M) {
            if (g.part != nullptr) g.dbpart[(int64_t)blockIdx.z * g.M + m0 + tid] = dbacc;
            else if (g.atomic) atomicAdd(g.db + m0 + tid, dbacc); else g.db[m0 + tid] = dbacc;
        }
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dout, int ldd,
                                                       const float* __restrict__ out, int ldo,
                                                       float* __restrict__ dpre, int ldp,
                                                       int64_t B, int N, int act) {
    const int64_t total = B * (int64_t)N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / N;
        const int c = (int)(i - r * N);
        dpre[r * ldp + c] = dout[r * ldd + c] * act_grad_from_out(out[r * ldo + c], act);
    }
}

static inline dim3 gemm_grid(int64_t M, int N, int ksplit) {
    return dim3((unsigned)cdiv(M, BM), (unsigned)cdiv(N, BN), (unsigned)ksplit);
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_linear_act_fwd(const float* x0, int k0, int ld0, const float* x1, int k1, int ld1,
                                   const float* W, int ldw, const float* bias, float* out, int ldo,
                                   int64_t B, int N, int act, void* stream) {
    DVAE_CHECK_ARG(x0 && W && out && B >= 0 && N > 0 && k0 > 0 && k1 >= 0, "linear_act_fwd: bad pointer or size");
    DVAE_CHECK_ARG(ld0 >= k0 && (k1 == 0 || (x1 && ld1 >= k1)) && ldw >= k0 + k1 && ldo >= N, "linear_act_fwd: bad leading dimension");
    DVAE_CHECK_ARG(act >= DVAE_ACT_NONE && act <= DVAE_ACT_EXP, "linear_act_fwd: unknown activation %d", act);
    DVAE_CHECK_ARG(cdiv(N, BN) <= 65535, "linear_act_fwd: N too large");
    if (B == 0) return 0;
    GemmArgs g{};
    g.A = MatCat{x0, x1, k0, k1, ld0, ld1, B};
    g.B = MatCat{W, nullptr, k0 + k1, 0, ldw, 0, (int64_t)N};
    g.M = B; g.N = N; g.K = k0 + k1; g.kper = cdiv(g.K, BK) * BK;
    g.out = out; g.ldo = ldo; g.bias = bias; g.act = act;
    hipLaunchKernelGGL((gemm_f32_kernel<false, false, 0>), gemm_grid(B, N, 1), dim3(256), 0, (hipStream_t)stream, g);
    DVAE_LAUNCH_OK("gemm_f32<fwd>");
    return 0;
}

extern "C" int dvae_act_bwd(const float* dout, int ldd, const float* out, int ldo, float* dpre, int ldp,
                            int64_t B, int N, int act, void* stream) {
    DVAE_CHECK_ARG(dout && out && dpre && B >= 0 && N > 0 && ldd >= N && ldo >= N && ldp >= N, "act_bwd: bad argument");
    if (B == 0) return 0;
    const int64_t total = B * (int64_t)N;
    int blocks = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, ldd, out, ldo, dpre, ldp, B, N, act);
    DVAE_LAUNCH_OK("act_bwd");
    return 0;
}

extern "C" int dvae_linear_bwd_data(const float* dpre, int ldp, const float* W, int ldw, int koff,
                                    float* din, int ldi, int64_t B, int N, int K, int accumulate, void* stream) {
    DVAE_CHECK_ARG(dpre && W && din && B >= 0 && N > 0 && K > 0 && koff >= 0, "linear_bwd_data: bad pointer or size");
    DVAE_CHECK_ARG(ldp >= N && ldw >= koff + K && ldi >= K, "linear_bwd_data: bad leading dimension");
    DVAE_CHECK_ARG(cdiv(K, BN) <= 65535, "linear_bwd_data: K too large");
    if (B == 0) return 0;
    GemmArgs g{};
    g.A = MatCat{dpre, nullptr, N, 0, ldp, 0, B};
    g.B = MatCat{W + koff, nullptr, K, 0, ldw, 0, (int64_t)N};
    g.M = B; g.N = K; g.K = N; g.kper = cdiv(g.K, BK) * BK;
    g.out = din; g.ldo = ldi; g.accumulate = accumulate;
    hipLaunchKernelGGL((gemm_f32_kernel<false, true, 1>), gemm_grid(B, K, 1), dim3(256), 0, (hipStream_t)stream, g);
    DVAE_LAUNCH_OK("gemm_f32<bwd_data>");
    return 0;
}

// deterministic combination of the slice partials: slices summed in ascending order, one thread per output element
__global__ __launch_bounds__(256) void slices_sum_kernel(const float* __restrict__ part, int64_t part_stride, int nslices, float* __restrict__ dW, int ldw, int ldp,
                                                        int N, int Kin, const float* __restrict__ dbpart, float* __restrict__ db) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)N * Kin) {
        const int64_t r = i / Kin, c = i - r * Kin;
        float s = 0.f;
        for (int z = 0; z < nslices; ++z) s += part[(int64_t)z * part_stride + r * ldp + c];
        dW[r * ldw + c] = s;
    } else if (db != nullptr && i < (int64_t)N * Kin + N) {
        const int64_t r = i - (int64_t)N * Kin;
        float s = 0.f;
        for (int z = 0; z < nslices; ++z) s += dbpart[(int64_t)z * N + r];
        db[r] = s;
    }
}

static int bwd_weight_slices(int64_t B, int N, int Kin, int ksplit, int64_t* kper_out) {
    const int64_t tiles = cdiv(N, BM) * cdiv(Kin, BN);
    if (ksplit <= 0) {
        int64_t want = cdiv(1024, tiles);
        int64_t maxs = cdiv(B, 256);
        ksplit = (int)(want < maxs ? want : maxs);
        if (ksplit < 1) ksplit = 1;
    }
    if (ksplit > 65535) ksplit = 65535;
    int64_t kper = cdiv(cdiv(B, ksplit), BK) * BK;
    if (kper < BK) kper = BK;
    ksplit = (int)cdiv(B, kper);
    if (ksplit < 1) ksplit = 1;
    *kper_out = kper;
    return ksplit;
}

extern "C" size_t dvae_linear_bwd_weight_workspace_bytes(int64_t B, int N, int Kin, int ksplit) {
    int64_t kper;
    const int ns = bwd_weight_slices(B > 0 ? B : 1, N, Kin, ksplit, &kper);
    return ns > 1 ? (size_t)ns * ((size_t)N * Kin + N) * sizeof(float) : 0;
}

extern "C" int dvae_linear_bwd_weight_det(const float* dpre, int ldp, const float* x0, int k0, int ld0,
                                          const float* x1, int k1, int ld1, float* dW, int ldw, float* db,
                                          int64_t B, int N, int ksplit, void* workspace, void* stream) {
    DVAE_CHECK_ARG(dpre && x0 && dW && B >= 0 && N > 0 && k0 > 0 && k1 >= 0, "linear_bwd_weight_det: bad pointer or size");
    DVAE_CHECK_ARG(ldp >= N && ld0 >= k0 && (k1 == 0 || (x1 && ld1 >= k1)) && ldw >= k0 + k1, "linear_bwd_weight_det: bad leading dimension");
    const int Kin = k0 + k1;
    DVAE_CHECK_ARG(cdiv(Kin, BN) <= 65535, "linear_bwd_weight_det: fan-in too large");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        DVAE_HIP(hipMemset2DAsync(dW, (size_t)ldw * sizeof(float), 0, (size_t)Kin * sizeof(float), (size_t)N, s));
        if (db) DVAE_HIP(hipMemsetAsync(db, 0, (size_t)N * sizeof(float), s));
        return 0;
    }
    int64_t kper;
    const int ns = bwd_weight_slices(B, N, Kin, ksplit, &kper);
    DVAE_CHECK_ARG(ns == 1 || workspace != nullptr, "linear_bwd_weight_det: %d slices need the workspace of dvae_linear_bwd_weight_workspace_bytes", ns);
    GemmArgs g{};
    g.A = MatCat{dpre, nullptr, N, 0, ldp, 0, B};
    g.B = MatCat{x0, x1, k0, k1, ld0, ld1, B};
    g.M = N; g.N = Kin; g.K = B; g.kper = kper;
    g.out = dW; g.ldo = ldw; g.db = db; g.atomic = 0;
    if (ns > 1) {
        g.part = (float*)workspace; g.part_stride = (int64_t)N * Kin; g.ldo = Kin;
        g.dbpart = (float*)workspace + (int64_t)ns * N * Kin;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<true, true, 2>), gemm_grid(N, Kin, ns), dim3(256), 0, s, g);
    DVAE_LAUNCH_OK("gemm_f32<bwd_weight, slices>");
    if (ns > 1) {
        const int64_t total = (int64_t)N * Kin + (db ? N : 0);
        hipLaunchKernelGGL(slices_sum_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, g.part, g.part_stride, ns, dW, ldw, Kin, N, Kin, g.dbpart, db);
        DVAE_LAUNCH_OK("slices_sum_kernel");
    }
    return 0;
}

extern "C" int dvae_linear_bwd_weight(const float* dpre, int ldp, const float* x0, int k0, int ld0,
                                      const float* x1, int k1, int ld1, float* dW, int ldw, float* db,
                                      int64_t B, int N, int ksplit, void* stream) {
    DVAE_CHECK_ARG(dpre && x0 && dW && B >= 0 && N > 0 && k0 > 0 && k1 >= 0, "linear_bwd_weight: bad pointer or size");
    DVAE_CHECK_ARG(ldp >= N && ld0 >= k0 && (k1 == 0 || (x1 && ld1 >= k1)) && ldw >= k0 + k1, "linear_bwd_weight: bad leading dimension");
    const int Kin = k0 + k1;
    DVAE_CHECK_ARG(cdiv(Kin, BN) <= 65535, "linear_bwd_weight: fan-in too large");
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = cdiv(N, BM) * cdiv(Kin, BN);
    if (ksplit <= 0) {
        int64_t want = cdiv(1024, tiles);
        int64_t maxs = cdiv(B, 256);
        ksplit = (int)(want < maxs ? want : maxs);
        if (ksplit < 1) ksplit = 1;
    }
    if (ksplit > 65535) ksplit = 65535;
    GemmArgs g{};
    g.A = MatCat{dpre, nullptr, N, 0, ldp, 0, B};
    g.B = MatCat{x0, x1, k0, k1, ld0, ld1, B};
    g.M = N; g.N = Kin; g.K = B;
    g.kper = cdiv(cdiv(B, ksplit), BK) * BK;
    if (g.kper < BK) g.kper = BK;
    ksplit = (int)cdiv(B, g.kper);
    if (ksplit < 1) ksplit = 1;
    g.out = dW; g.ldo = ldw; g.db = db; g.atomic = ksplit > 1;
    if (g.atomic || B == 0) {
        DVAE_HIP(hipMemset2DAsync(dW, (size_t)ldw * sizeof(float), 0, (size_t)Kin * sizeof(float), (size_t)N, s));
        if (db) DVAE_HIP(hipMemsetAsync(db, 0, (size_t)N * sizeof(float), s));
        if (B == 0) return 0;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<true, true, 2>), gemm_grid(N, Kin, ksplit), dim3(256), 0, s, g);
    DVAE_LAUNCH_OK("gemm_f32<bwd_weight>");
    return 0;
}
