// STFT / ISTFT for packages/processing/stft.py (librosa semantics restated in
// oracle/stft_oracle.py).  One workgroup walks frames; per frame the nfft real
// samples are windowed and packed into an nfft/2-point complex FFT that runs
// entirely in LDS (double precision: the reference transforms float64 audio
// and only then casts to complex64), followed by the real-FFT split step.
// Twiddles and the window are staged in LDS once per workgroup.
// ISTFT = inverse of the same split + FFT, windowed frames to a scratch
// buffer, then a gather overlap-add that replays librosa's float32
// frame-by-frame accumulation order exactly (deterministic, no atomics).
#include <float.h>
#include <stdlib.h>
#include "common.hpp"

namespace dvae {

struct cd { double x, y; };
__device__ __forceinline__ cd cmul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd cadd(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cconj(cd a) { return cd{a.x, -a.y}; }

// in-LDS radix-2 DIT FFT of M = 1 << logM points already stored in bit-reversed order.
// tw[k] = exp(-2 pi i k / (2M)), k < M.  inverse != 0 conjugates the twiddles.
__device__ __forceinline__ void fft_lds(cd* z, const cd* tw, int logM, int inverse) {
    const int M = 1 << logM;
    for (int s = 1; s <= logM; ++s) {
        const int half = 1 << (s - 1);
        for (int j = threadIdx.x; j < (M >> 1); j += blockDim.x) {
            const int grp = j >> (s - 1), pos = j & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            cd w = tw[2 * pos * (M >> s)];
            if (inverse) w.y = -w.y;
            const cd a = z[i0], b = cmul(w, z[i1]);
            z[i0] = cadd(a, b);
            z[i1] = csub(a, b);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void stage_tables(cd* tw, double* win, const double* window, int nfft) {
    const int M = nfft >> 1;
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)nfft, &s, &c);
        tw[k] = cd{c, s};
    }
    for (int i = threadIdx.x; i < nfft; i += blockDim.x) win[i] = window[i];
}

__device__ __forceinline__ void store_bin(void* out, int layout, int64_t T, int F, int64_t t, int f, cd v) {
    float* o = (float*)out;
    const float re = (float)v.x, im = (float)v.y;
    if (layout == 0) {            // complex64 [F][T]  (column = frame)
        o[(f * T + t) * 2] = re;
        o[(f * T + t) * 2 + 1] = im;
    } else if (layout == 2) {     // complex64 [T][F]  (row = frame: the memory order of librosa's Fortran-ordered result)
        o[(t * F + f) * 2] = re;
        o[(t * F + f) * 2 + 1] = im;
    } else {                      // power [T][F] float32: np.abs(complex64)**2
        const float a = hypotf(re, im);
        o[t * F + f] = a * a;
    }
}

template <typename TIN>
__global__ __launch_bounds__(256) void stft_pow2_kernel(const TIN* __restrict__ x, int64_t n, const double* __restrict__ window,
                                                         int nfft, int logM, int hop, int64_t T, void* out, int layout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = nfft >> 1, F = M + 1;
    cd* z = (cd*)smem;
    cd* tw = z + M;
    double* win = (double*)(tw + M);
    stage_tables(tw, win, window, nfft);
    __syncthreads();
    for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
        const int64_t base = t * hop;
        for (int i = threadIdx.x; i < M; i += blockDim.x) {
            const int64_t s0 = base + 2 * i;
            const double a = (s0 < n) ? (double)x[s0] * win[2 * i] : 0.0;
            const double b = (s0 + 1 < n) ? (double)x[s0 + 1] * win[2 * i + 1] : 0.0;
            z[__brev((unsigned)i) >> (32 - logM)] = cd{a, b};
        }
        __syncthreads();
        fft_lds(z, tw, logM, 0);
        // split: X[k] = E + W^k O, X[M-k] = conj(E - W^k O)
        for (int k = threadIdx.x; k <= (M >> 1); k += blockDim.x) {
            const cd zk = z[k], zc = cconj(z[(M - k) & (M - 1)]);
            const cd e = cd{0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y)};
            const cd d = csub(zk, zc);
            const cd o = cd{0.5 * d.y, -0.5 * d.x};          // -0.5 i (zk - zc)
            const cd wo = cmul(tw[k], o);
            store_bin(out, layout, T, F, t, k, cadd(e, wo));
            store_bin(out, layout, T, F, t, M - k, cconj(csub(e, wo)));
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// nfft = 1024 (every caller of the reference): ONE WAVE per frame.  The 512-point complex FFT is three radix-8
// Stockham passes with 8 points per lane in registers; lanes exchange data through a private LDS buffer between
// passes (no workgroup barrier anywhere: a wave's LDS accesses are ordered), the window and all twiddles live in
// registers for the whole launch.  Double precision throughout (the reference transforms float64 audio and only
// then casts to complex64).  Power frames ([T][513], the training layout) leave as 256-byte runs per wave; the
// complex [513][T] layout is staged through LDS 16 frames at a time so each bin's 16 frames leave as one 128-byte run.
__device__ __forceinline__ cd cmulc(cd a, double wr, double wi) { return cd{a.x * wr - a.y * wi, a.x * wi + a.y * wr}; }

// in-place 8-point DFT (forward), natural-order output
__device__ __forceinline__ void dft8(cd (&a)[8]) {
    constexpr double H = 0.70710678118654752440;
    cd b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = cadd(a[i], a[i + 4]); }
    { const cd d = csub(a[0], a[4]); b[4] = d; }
    { const cd d = csub(a[1], a[5]); b[5] = cd{(d.x + d.y) * H, (d.y - d.x) * H}; }        // * (1 - i)/sqrt2
    { const cd d = csub(a[2], a[6]); b[6] = cd{d.y, -d.x}; }                               // * -i
    { const cd d = csub(a[3], a[7]); b[7] = cd{(d.y - d.x) * H, -(d.x + d.y) * H}; }       // * (-1 - i)/sqrt2
    cd c[8];
#pragma unroll
    for (int q = 0; q < 8; q += 4) {
        c[q] = cadd(b[q], b[q + 2]); c[q + 1] = cadd(b[q + 1], b[q + 3]);
        c[q + 2] = csub(b[q], b[q + 2]);
        const cd d = csub(b[q + 1], b[q + 3]); c[q + 3] = cd{d.y, -d.x};                   // * -i
    }
    a[0] = cadd(c[0], c[1]); a[4] = csub(c[0], c[1]); a[2] = cadd(c[2], c[3]); a[6] = csub(c[2], c[3]);
    a[1] = cadd(c[4], c[5]); a[5] = csub(c[4], c[5]); a[3] = cadd(c[6], c[7]); a[7] = csub(c[6], c[7]);
}

__device__ __forceinline__ int padidx(int i) { return i + (i >> 3); }     // one double of padding per 8: strides 8 and 64 both conflict-free

// 512-point complex forward FFT of one wave: v[r] = x[lane + 64 r] in, X[lane + 64 r] out (both natural order), three
// radix-8 Stockham passes; re / im: the wave's private padded LDS exchange buffers (512 + 64 doubles each).
struct Fft512 {
    double t1r[8], t1i[8], t2r[8], t2i[8];
    __device__ __forceinline__ void init(int lane) {
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            sincospi(-2.0 * (double)(r * (lane & 7)) / 64.0, &t1i[r], &t1r[r]);      // pass 1 (Ns = 8): exp(-2 pi i r k / 64), k = lane & 7
            sincospi(-2.0 * (double)(r * lane) / 512.0, &t2i[r], &t2r[r]);           // pass 2 (Ns = 64): exp(-2 pi i r lane / 512)
        }
    }
    __device__ __forceinline__ void run(cd (&v)[8], double* re, double* im, int lane) const {
        // pass 0 (Ns = 1): no twiddles; outputs to index lane*8 + r
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane * 8 + r); re[i] = v[r].x; im[i] = v[r].y; }
        __builtin_amdgcn_wave_barrier();
        // pass 1 (Ns = 8)
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane + 64 * r); v[r] = cd{re[i], im[i]}; }
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmulc(v[r], t1r[r], t1i[r]);
        dft8(v);
        __builtin_amdgcn_wave_barrier();
        {
            const int j0 = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int i = padidx(j0 + 8 * r); re[i] = v[r].x; im[i] = v[r].y; }
        }
        __builtin_amdgcn_wave_barrier();
        // pass 2 (Ns = 64): outputs X[lane + 64 r] stay in registers
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane + 64 * r); v[r] = cd{re[i], im[i]}; }
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmulc(v[r], t2r[r], t2i[r]);
        dft8(v);
    }
};

// The same transform with the pass-1 twiddles read from an LDS table t1l[(lane & 7) * 8 + r] = exp(-2 pi i r (lane & 7) / 64) (they depend on
// lane & 7 only): 28 registers less per wave (three waves per SIMD: stft1024_walk_kernel, OCC3)
struct Fft512L {
    double t2r[8], t2i[8];
    __device__ __forceinline__ void init(int lane) {
#pragma unroll
        for (int r = 1; r < 8; ++r) sincospi(-2.0 * (double)(r * lane) / 512.0, &t2i[r], &t2r[r]);
    }
    __device__ __forceinline__ void run(cd (&v)[8], double* re, double* im, int lane, const double2* t1l) const {
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane * 8 + r); re[i] = v[r].x; im[i] = v[r].y; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane + 64 * r); v[r] = cd{re[i], im[i]}; }
#pragma unroll
        for (int r = 1; r < 8; ++r) { const double2 t = t1l[(lane & 7) * 8 + r]; v[r] = cmulc(v[r], t.x, t.y); }
        dft8(v);
        __builtin_amdgcn_wave_barrier();
        {
            const int j0 = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int i = padidx(j0 + 8 * r); re[i] = v[r].x; im[i] = v[r].y; }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane + 64 * r); v[r] = cd{re[i], im[i]}; }
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmulc(v[r], t2r[r], t2i[r]);
        dft8(v);
    }
};

constexpr int STFT_FR = 16;        // frames staged per workgroup pass of the complex layout

template <typename TIN, int LAYOUT>
__global__ __launch_bounds__(256) void stft1024_kernel(const TIN* __restrict__ x, int64_t n, const double* __restrict__ window,
                                                        int hop, int64_t T, int chunk, void* out) {
    constexpr int M = 512, F = 513;
    __shared__ double lre[4][M + 64], lim[4][M + 64];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* stage = reinterpret_cast<float2*>(smem);              // LAYOUT 0: [F][STFT_FR + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* re = lre[wave];
    double* im = lim[wave];
    // per-lane constants
    double wa[8], wb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { wa[r] = window[2 * (lane + 64 * r)]; wb[r] = window[2 * (lane + 64 * r) + 1]; }
    Fft512 fft;
    fft.init(lane);
    double sr[5], si[5];                                                           // split twiddles exp(-2 pi i k / 1024), k = lane + 64 r; [4]: k = 256
#pragma unroll
    for (int r = 0; r < 4; ++r) sincospi(-2.0 * (double)(lane + 64 * r) / 1024.0, &si[r], &sr[r]);
    sr[4] = 0.0; si[4] = -1.0;

    // raw sample pairs of frame t (the zero end-pad is implied past n); requested one frame ahead of the transform
    struct TIN2 { TIN a, b; };
    auto fetch = [&](int64_t t, TIN2 (&raw)[8]) {
        const int64_t base = t * hop;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t s0 = base + 2 * (lane + 64 * r);
            if (s0 + 1 < n) raw[r] = *reinterpret_cast<const TIN2*>(x + s0);      // hop and the pair offset are even: aligned pair loads
            else { raw[r].a = s0 < n ? x[s0] : (TIN)0; raw[r].b = (TIN)0; }
        }
    };
    // consecutive frames overlap by nfft - hop samples: with hop = 256 (128 pairs = 2 slots of 64 lanes) pair slot r of
    // frame t+1 is slot r+2 of frame t IN THE SAME LANE, so a wave walking consecutive frames loads only slots 6 and 7
    auto advance = [&](int64_t tnext, const TIN2 (&prev)[8], TIN2 (&raw)[8]) {
#pragma unroll
        for (int r = 0; r < 6; ++r) raw[r] = prev[r + 2];
        const int64_t base = tnext * hop;
#pragma unroll
        for (int r = 6; r < 8; ++r) {
            const int64_t s0 = base + 2 * (lane + 64 * r);
            if (s0 + 1 < n) raw[r] = *reinterpret_cast<const TIN2*>(x + s0);
            else { raw[r].a = s0 < n ? x[s0] : (TIN)0; raw[r].b = (TIN)0; }
        }
    };
    auto one_frame = [&](int64_t t, const TIN2 (&raw)[8], auto&& emit) {
        cd v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = cd{(double)raw[r].a * wa[r], (double)raw[r].b * wb[r]};
        fft.run(v, re, im, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane + 64 * r); re[i] = v[r].x; im[i] = v[r].y; }
        __builtin_amdgcn_wave_barrier();
        // real-FFT split: X[k] = E + W^k O, X[M-k] = conj(E - W^k O), partner z[M-k] from LDS
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const int pi = padidx((M - k) & (M - 1));
            const cd zk = v[r], zc = cd{re[pi], -im[pi]};
            const cd e = cd{0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y)};
            const cd d = csub(zk, zc);
            const cd wo = cmulc(cd{0.5 * d.y, -0.5 * d.x}, sr[r], si[r]);
            emit(k, cadd(e, wo));
            emit(M - k, cconj(csub(e, wo)));
        }
        if (lane == 0) {                                         // k = 256 (its own partner): X = E + (-i) O
            const cd zk = v[4];
            emit(256, cd{zk.x, -zk.y});
        }
        __builtin_amdgcn_wave_barrier();
    };

    TIN2 cur[8], nxt[8];
    if (LAYOUT == 2) {
        // complex frames, frame-major: the walk of the power layout, each bin leaving as its complex64 value (512-byte runs per wave)
        float2* o = (float2*)out;
        const int64_t tb = ((int64_t)blockIdx.x * 4 + wave) * chunk;
        const int64_t te = tb + chunk < T ? tb + chunk : T;
        if (tb < te) fetch(tb, cur);
        for (int64_t t = tb; t < te; ++t) {
            if (t + 1 < te) { if (hop == 256) advance(t + 1, cur, nxt); else fetch(t + 1, nxt); }
            one_frame(t, cur, [&](int f, cd X) { o[t * F + f] = float2{(float)X.x, (float)X.y}; });
#pragma unroll
            for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
        }
    } else if (LAYOUT == 1) {
        float* o = (float*)out;
        // each wave walks `chunk` consecutive frames (chunk chosen by the host so that the launch still fills the chip)
        const int64_t tb = ((int64_t)blockIdx.x * 4 + wave) * chunk;
        const int64_t te = tb + chunk < T ? tb + chunk : T;
        if (tb < te) fetch(tb, cur);
        for (int64_t t = tb; t < te; ++t) {
            if (t + 1 < te) { if (hop == 256) advance(t + 1, cur, nxt); else fetch(t + 1, nxt); }   // in flight under this frame's transform
            one_frame(t, cur, [&](int f, cd X) {
                // np.abs(complex64) ** 2: float32 magnitude, then its square.  The magnitudes here are far from
                // overflow, so the correctly rounded square root of re^2 + im^2 stands in for hypotf (30 instructions)
                const float re32 = (float)X.x, im32 = (float)X.y;
                const float a = __fsqrt_rn(fmaf(re32, re32, im32 * im32));
                o[t * F + f] = a * a;
            });
#pragma unroll
            for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
        }
    } else {
        float2* o = (float2*)out;
        for (int64_t t0 = (int64_t)blockIdx.x * STFT_FR; t0 < T; t0 += (int64_t)gridDim.x * STFT_FR) {
            constexpr int PW = STFT_FR / 4;                           // consecutive frames per wave
            if (t0 + wave * PW < T) fetch(t0 + wave * PW, cur);
            for (int q = wave * PW; q < (wave + 1) * PW; ++q) {
                const int64_t t = t0 + q;
                if (q + 1 < (wave + 1) * PW && t + 1 < T) { if (hop == 256) advance(t + 1, cur, nxt); else fetch(t + 1, nxt); }
                if (t < T) one_frame(t, cur, [&](int f, cd X) { stage[f * (STFT_FR + 1) + q] = float2{(float)X.x, (float)X.y}; });
#pragma unroll
                for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
            }
            __syncthreads();
            const int nq = (int)(T - t0 < STFT_FR ? T - t0 : STFT_FR);
            for (int idx = threadIdx.x; idx < F * STFT_FR; idx += 256) {
                const int f = idx / STFT_FR, q = idx - f * STFT_FR;
                if (q < nq) o[(int64_t)f * T + t0 + q] = stage[f * (STFT_FR + 1) + q];
            }
            __syncthreads();
        }
    }
}

// hop = 256 (every caller of the reference), frame-major outputs: the walk of stft1024_kernel<., 1 / 2> with the per-frame VALU work that is
// not the transform taken out.  Round 3 measured the walk at ~100 % VALU issue (2 waves per SIMD, 520 VALU instructions per frame, 282 of them
// fp64 transform arithmetic); the rest was (a) 48 v_mov_b64 rotating the six carried sample pairs from one frame's slots into the next, (b) ~80
// instructions of 64-bit address arithmetic and end-of-signal compares around 10 loads and 9 stores, (c) the denormal-input scaling hipcc wraps
// around v_sqrt_f32 (5 of 10 instructions per bin).  Here: (a) the pairs live in a ring of 8 registers indexed by frame number mod 4 -- the loop
// is unrolled by four and nothing moves; (b) loads and stores go through buffer descriptors: one per-lane byte offset, the frame offset in an
// SGPR (the scalar offset is outside the descriptor's range check on gfx9: every access is in range by the host's own check that all T
// frames fit in n samples, dvae_stft: "(T - 1) hop + nfft <= n" -- the caller passes the end-padded signal); (c) the raw v_sqrt_f32 (1 ulp,
// not the correctly rounded sqrtf: its square is within ~2 ulp of np.abs(complex64) ** 2, inside the 4e-7 relative bound at which
// tests/test_gpu_stft.py pins the reference's HDF5 power frames; |X|^2 below 1.2e-38 -- where the reference's own result is a denormal
// or zero -- gives 0: parity for denormal magnitudes is unpinned by any reference fixture).
template <typename TIN, bool POWER, bool OCC3>
__global__ __launch_bounds__(256, OCC3 ? 3 : 2) void stft1024_walk_kernel(const TIN* __restrict__ x, int64_t n, const double* __restrict__ window, int64_t T, int chunk,
                                                             void* out) {
    constexpr int M = 512, F = 513;
    constexpr int ESZ = POWER ? 4 : 8;
    __shared__ double lre[4][M + 64], lim[4][M + 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform frame numbers: scalar buffer offsets)
    double* re = lre[wave];
    double* im = lim[wave];
    // OCC3 (diagnostic build, measured slower -- see the launcher): three waves per SIMD (<= 168 registers): the window (32 registers), the
    // pass-1 twiddles (28) and the split twiddles (16) are read from LDS tables every frame instead (19 ds_read_b128)
    __shared__ double2 lwin[OCC3 ? M : 1], lsp[OCC3 ? M / 2 : 1], lt1[OCC3 ? 64 : 1];
    double wa[OCC3 ? 1 : 8], wb[OCC3 ? 1 : 8], sr[OCC3 ? 1 : 4], si[OCC3 ? 1 : 4];
    typename std::conditional<OCC3, Fft512L, Fft512>::type fft;
    fft.init(lane);
    if constexpr (OCC3) {
        for (int k = threadIdx.x; k < M; k += 256) lwin[k] = double2{window[2 * k], window[2 * k + 1]};
        for (int k = threadIdx.x; k < M / 2; k += 256) { double sn, cs; sincospi(-2.0 * (double)k / 1024.0, &sn, &cs); lsp[k] = double2{cs, sn}; }
        if (threadIdx.x < 64) { double sn, cs; sincospi(-2.0 * (double)((threadIdx.x & 7) * (threadIdx.x >> 3)) / 64.0, &sn, &cs); lt1[threadIdx.x] = double2{cs, sn}; /* entry (k = tid >> 3, r = tid & 7) */ }
        __syncthreads();
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) { wa[r] = window[2 * (lane + 64 * r)]; wb[r] = window[2 * (lane + 64 * r) + 1]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) sincospi(-2.0 * (double)(lane + 64 * r) / 1024.0, &si[r], &sr[r]);   // split twiddles exp(-2 pi i k / 1024), k = lane + 64 r
    }

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<TIN*>(x), 0, (int)(n * (int64_t)sizeof(TIN)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(T * F * ESZ), 0x00020000);
    struct TIN2 { TIN a, b; };
    const int vx = lane * 2 * (int)sizeof(TIN);
    // pair slot r of frame t: samples 256 t + 128 r + 2 lane, + 1 (always inside n: the host checks that every frame fits)
    auto ldpair = [&](int64_t t, int r) __attribute__((always_inline)) {
        const int so = (int)((t * 256 + 128 * r) * (int64_t)sizeof(TIN));
        if constexpr (sizeof(TIN) == 8) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vx, so, 0);
            return __builtin_bit_cast(TIN2, v);
        } else {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, vx, so, 0);
            return __builtin_bit_cast(TIN2, v);
        }
    };
    const int vk = lane * ESZ, vm = (M - 192 - lane) * ESZ;                        // bins lane + 64 r / 512 - lane - 64 r: + 64 r ESZ / + 64 (3 - r) ESZ
    auto put = [&](int voff, int so, cd X) __attribute__((always_inline)) {
        const float re32 = (float)X.x, im32 = (float)X.y;
        if constexpr (POWER) {
            // np.abs(complex64) ** 2: float32 magnitude (v_sqrt_f32 of re^2 + im^2, 1 ulp, stands in for hypotf: far from overflow), squared
            const float a = __builtin_amdgcn_sqrtf(fmaf(re32, re32, im32 * im32));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a * a), rs_o, voff, so, 0);
        } else {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u2{__builtin_bit_cast(unsigned, re32), __builtin_bit_cast(unsigned, im32)}, rs_o, voff, so, 0);
        }
    };

    const int64_t tb = ((int64_t)blockIdx.x * 4 + wave) * chunk;
    const int64_t te = tb + chunk < T ? tb + chunk : T;
    TIN2 buf[8];                                                                   // ring: slot r of a frame with t - tb = p (mod 4) is buf[(r + 2 p) & 7]
    if (tb < te) {
#pragma unroll
        for (int r = 0; r < 8; ++r) buf[r] = ldpair(tb, r);
    }
    auto frame = [&](auto phc, int64_t t) __attribute__((always_inline)) {
        constexpr int PH = decltype(phc)::value;
        cd v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const TIN2 q = buf[(r + 2 * PH) & 7];
            if constexpr (OCC3) { const double2 w = lwin[lane + 64 * r]; v[r] = cd{(double)q.a * w.x, (double)q.b * w.y}; }
            else v[r] = cd{(double)q.a * wa[r], (double)q.b * wb[r]};
        }
        if (t + 1 < te) {                                                          // slots 6, 7 of the next frame take the places of this frame's slots 0, 1
            buf[(2 * PH) & 7] = ldpair(t + 1, 6);
            buf[(2 * PH + 1) & 7] = ldpair(t + 1, 7);
        }
        if constexpr (OCC3) fft.run(v, re, im, lane, lt1); else fft.run(v, re, im, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = padidx(lane + 64 * r); re[i] = v[r].x; im[i] = v[r].y; }
        __builtin_amdgcn_wave_barrier();
        const int so = (int)(t * F * ESZ);
        // real-FFT split: X[k] = E + W^k O, X[M-k] = conj(E - W^k O), partner z[M-k] from LDS
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const int pi = padidx((M - k) & (M - 1));
            const cd zk = v[r], zc = cd{re[pi], -im[pi]};
            const cd e = cd{0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y)};
            const cd d = csub(zk, zc);
            double swr, swi;
            if constexpr (OCC3) { const double2 w = lsp[k]; swr = w.x; swi = w.y; } else { swr = sr[r]; swi = si[r]; }
            const cd wo = cmulc(cd{0.5 * d.y, -0.5 * d.x}, swr, swi);
            put(vk + 64 * r * ESZ, so, cadd(e, wo));
            put(vm + 64 * (3 - r) * ESZ, so, cconj(csub(e, wo)));
        }
        if (lane == 0) put(256 * ESZ, so, cd{v[4].x, -v[4].y});                    // k = 256 (its own partner): X = E + (-i) O
        __builtin_amdgcn_wave_barrier();
    };
    for (int64_t t = tb; t < te; t += 4) {
        frame(std::integral_constant<int, 0>{}, t);
        if (t + 1 < te) frame(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < te) frame(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < te) frame(std::integral_constant<int, 3>{}, t + 3);
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 ARITHMETIC, nfft = 1024 / hop = 256: the transform of stft_pytorch (packages/processing/stft.py:123-152 = torch.stft on an fp32
// tensor with torch.hann_window(1024): window product, FFT and output all in float32; the caller squares and adds in float32 as well,
// packages/data_handling.py:136).  The float64 walk above stays the transform of stft() (librosa multiplies by a float64 window, so its
// FFT runs in double whatever the audio's type) and of every bit-level pin.  Same walk -- one wave per frame, three radix-8 Stockham
// passes, 8 points per lane, six of the eight sample pairs carried over in a register ring -- with what the narrower type buys: a point
// is ONE 8-byte LDS slot (re, im) instead of two 8-byte doubles (half the exchange instructions, half the bytes), no fp64 VALU (half
// rate on gfx950), and under 128 registers, i.e. four waves per SIMD instead of two.
struct cf { float x, y; };
__device__ __forceinline__ cf cfadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf cfsub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cfconj(cf a) { return cf{a.x, -a.y}; }
__device__ __forceinline__ cf cfmulc(cf a, float wr, float wi) { return cf{a.x * wr - a.y * wi, a.x * wi + a.y * wr}; }

__device__ __forceinline__ void dft8f(cf (&a)[8]) {
    constexpr float H = 0.70710678118654752440f;
    cf b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = cfadd(a[i], a[i + 4]); }
    { const cf d = cfsub(a[0], a[4]); b[4] = d; }
    { const cf d = cfsub(a[1], a[5]); b[5] = cf{(d.x + d.y) * H, (d.y - d.x) * H}; }
    { const cf d = cfsub(a[2], a[6]); b[6] = cf{d.y, -d.x}; }
    { const cf d = cfsub(a[3], a[7]); b[7] = cf{(d.y - d.x) * H, -(d.x + d.y) * H}; }
    cf c[8];
#pragma unroll
    for (int q = 0; q < 8; q += 4) {
        c[q] = cfadd(b[q], b[q + 2]); c[q + 1] = cfadd(b[q + 1], b[q + 3]);
        c[q + 2] = cfsub(b[q], b[q + 2]);
        const cf d = cfsub(b[q + 1], b[q + 3]); c[q + 3] = cf{d.y, -d.x};
    }
    a[0] = cfadd(c[0], c[1]); a[4] = cfsub(c[0], c[1]); a[2] = cfadd(c[2], c[3]); a[6] = cfsub(c[2], c[3]);
    a[1] = cfadd(c[4], c[5]); a[5] = cfsub(c[4], c[5]); a[3] = cfadd(c[6], c[7]); a[7] = cfsub(c[6], c[7]);
}

struct Fft512F {
    float t1r[8], t1i[8], t2r[8], t2i[8];
    __device__ __forceinline__ void init(int lane) {
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            double sn, cs;
            sincospi(-2.0 * (double)(r * (lane & 7)) / 64.0, &sn, &cs); t1r[r] = (float)cs; t1i[r] = (float)sn;
            sincospi(-2.0 * (double)(r * lane) / 512.0, &sn, &cs); t2r[r] = (float)cs; t2i[r] = (float)sn;
        }
    }
    // z: the wave's private exchange buffer, 512 + 64 (re, im) slots, padded like the double buffers (one slot per 8)
    __device__ __forceinline__ void run(cf (&v)[8], cf* z, int lane) const {
        dft8f(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) z[padidx(lane * 8 + r)] = v[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = z[padidx(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cfmulc(v[r], t1r[r], t1i[r]);
        dft8f(v);
        __builtin_amdgcn_wave_barrier();
        {
            const int j0 = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) z[padidx(j0 + 8 * r)] = v[r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = z[padidx(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cfmulc(v[r], t2r[r], t2i[r]);
        dft8f(v);
    }
};

// waves per SIMD: 3 (168 registers).  Same box, alternating, ten minutes of float32 audio: complex frames 45.8 / 38.3 / 39.5 us and power
// frames 39.0 / 37.0 / 39.2 us at 4 / 3 / 2 (at 4 the complex form spills 8 registers)
#ifndef STFT_F32_OCC
#define STFT_F32_OCC 3
#endif
template <bool POWER>
__global__ __launch_bounds__(256, STFT_F32_OCC) void stft1024_walk_f32_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ window, int64_t T,
                                                                              int chunk, void* out) {
    constexpr int M = 512, F = 513;
    constexpr int ESZ = POWER ? 4 : 8;
    __shared__ __attribute__((aligned(8))) cf lz[4][M + 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cf* z = lz[wave];
    float wa[8], wb[8], sr[4], si[4];
    Fft512F fft;
    fft.init(lane);
#pragma unroll
    for (int r = 0; r < 8; ++r) { wa[r] = window[2 * (lane + 64 * r)]; wb[r] = window[2 * (lane + 64 * r) + 1]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { double sn, cs; sincospi(-2.0 * (double)(lane + 64 * r) / 1024.0, &sn, &cs); sr[r] = (float)cs; si[r] = (float)sn; }

    // every access in range by the host's check that all T frames fit in n samples (the scalar offset is not range-checked)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(n * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(T * F * ESZ), 0x00020000);
    const int vx = lane * 8;
    auto ldpair = [&](int64_t t, int r) __attribute__((always_inline)) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, vx, (int)((t * 256 + 128 * r) * 4), 0);
        return __builtin_bit_cast(cf, v);                                          // (samples 2 lane, 2 lane + 1 of the slot)
    };
    const int vk = lane * ESZ, vm = (M - 192 - lane) * ESZ;
    auto put = [&](int voff, int so, cf X) __attribute__((always_inline)) {
        if constexpr (POWER) {
            // x_tf[..., 0] ** 2 + x_tf[..., 1] ** 2 (packages/data_handling.py:136): two rounded squares, one rounded sum
            // (contraction switched off for the expression: __fmul_rn / __fadd_rn are plain operators in HIP's headers and would fuse)
            float pw;
            {
#pragma clang fp contract(off)
                const float a2 = X.x * X.x, b2 = X.y * X.y;
                pw = a2 + b2;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pw), rs_o, voff, so, 0);
        } else {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, X), rs_o, voff, so, 0);
        }
    };
    const int64_t tb = ((int64_t)blockIdx.x * 4 + wave) * chunk;
    const int64_t te = tb + chunk < T ? tb + chunk : T;
    cf buf[8];                                                                     // ring: slot r of a frame with t - tb = p (mod 4) is buf[(r + 2 p) & 7]
    if (tb < te) {
#pragma unroll
        for (int r = 0; r < 8; ++r) buf[r] = ldpair(tb, r);
    }
    auto frame = [&](auto phc, int64_t t) __attribute__((always_inline)) {
        constexpr int PH = decltype(phc)::value;
        cf v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { const cf q = buf[(r + 2 * PH) & 7]; v[r] = cf{q.x * wa[r], q.y * wb[r]}; }
        if (t + 1 < te) {
            buf[(2 * PH) & 7] = ldpair(t + 1, 6);
            buf[(2 * PH + 1) & 7] = ldpair(t + 1, 7);
        }
        fft.run(v, z, lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) z[padidx(lane + 64 * r)] = v[r];
        __builtin_amdgcn_wave_barrier();
        const int so = (int)(t * F * ESZ);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const cf zk = v[r], zc = cfconj(z[padidx((M - k) & (M - 1))]);
            const cf e = cf{0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y)};
            const cf d = cfsub(zk, zc);
            const cf wo = cfmulc(cf{0.5f * d.y, -0.5f * d.x}, sr[r], si[r]);
            put(vk + 64 * r * ESZ, so, cfadd(e, wo));
            put(vm + 64 * (3 - r) * ESZ, so, cfconj(cfsub(e, wo)));
        }
        if (lane == 0) put(256 * ESZ, so, cf{v[4].x, -v[4].y});
        __builtin_amdgcn_wave_barrier();
    };
    for (int64_t t = tb; t < te; t += 4) {
        frame(std::integral_constant<int, 0>{}, t);
        if (t + 1 < te) frame(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < te) frame(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < te) frame(std::integral_constant<int, 3>{}, t + 3);
    }
}

// generic O(N^2) DFT for non power-of-two window lengths (e.g. the wrapper's never-used
// default 50 ms = 800 samples): API completeness only.
template <typename TIN>
__global__ __launch_bounds__(256) void stft_dft_kernel(const TIN* __restrict__ x, int64_t n, const double* __restrict__ window,
                                                        int nfft, int hop, int64_t T, void* out, int layout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cd* tw = (cd*)smem;                 // nfft entries exp(-2 pi i k / nfft)
    double* fr = (double*)(tw + nfft);  // windowed frame
    const int F = nfft / 2 + 1;
    for (int k = threadIdx.x; k < nfft; k += blockDim.x) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)nfft, &s, &c);
        tw[k] = cd{c, s};
    }
    for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < nfft; i += blockDim.x) {
            const int64_t s0 = t * hop + i;
            fr[i] = (s0 < n) ? (double)x[s0] * window[i] : 0.0;
        }
        __syncthreads();
        for (int f = threadIdx.x; f < F; f += blockDim.x) {
            double re = 0.0, im = 0.0;
            int idx = 0;
            for (int i = 0; i < nfft; ++i) {
                re += fr[i] * tw[idx].x;
                im += fr[i] * tw[idx].y;
                idx += f; if (idx >= nfft) idx -= nfft;
            }
            store_bin(out, layout, T, F, t, f, cd{re, im});
        }
    }
}

// nfft = 1024: frames[t][m] = window[m] * irfft(S[:, t])[m] with ONE WAVE per frame (same FFT core, run on the
// conjugate: ifft(Z) = conj(fft(conj Z)) / M).  S is [bin][T]: a workgroup stages 16 consecutive frames through LDS
// (one 128-byte run per bin) and its four waves take four frames each.
constexpr int ISTFT_FR = 8;       // frames staged per workgroup pass: 37 KB + 37 KB of exchange buffers = two workgroups per CU
__global__ __launch_bounds__(256) void istft1024_frames_kernel(const float2* __restrict__ S, int64_t T, int64_t sf, int64_t st,
                                                               const double* __restrict__ window, double* __restrict__ frames) {
    constexpr int M = 512, F = 513, PW = ISTFT_FR / 4;
    __shared__ double lre[4][M + 64], lim[4][M + 64];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* stage = reinterpret_cast<float2*>(smem);              // [F][ISTFT_FR + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* re = lre[wave];
    double* im = lim[wave];
    double wa[8], wb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { wa[r] = window[2 * (lane + 64 * r)] * (1.0 / M); wb[r] = window[2 * (lane + 64 * r) + 1] * (1.0 / M); }
    Fft512 fft;
    fft.init(lane);
    double sr[8], si[8];                                          // exp(+2 pi i k / 1024), k = lane + 64 r
#pragma unroll
    for (int r = 0; r < 8; ++r) sincospi(2.0 * (double)(lane + 64 * r) / 1024.0, &si[r], &sr[r]);
    for (int64_t t0 = (int64_t)blockIdx.x * ISTFT_FR; t0 < T; t0 += (int64_t)gridDim.x * ISTFT_FR) {
        const int nq = (int)(T - t0 < ISTFT_FR ? T - t0 : ISTFT_FR);
        __syncthreads();                                          // the previous block's readers are done with `stage`
        for (int idx = threadIdx.x; idx < F * ISTFT_FR; idx += 256) {
            const int f = idx / ISTFT_FR, q = idx - f * ISTFT_FR;
            if (q < nq) stage[f * (ISTFT_FR + 1) + q] = S[(int64_t)f * sf + (t0 + q) * st];
        }
        __syncthreads();
        for (int q = wave * PW; q < (wave + 1) * PW && q < nq; ++q) {
            cd v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = lane + 64 * r;
                const float2 a = stage[k * (ISTFT_FR + 1) + q], b = stage[(M - k) * (ISTFT_FR + 1) + q];
                cd xk = cd{(double)a.x, (double)a.y}, xc = cd{(double)b.x, -(double)b.y};   // X[k], conj(X[M-k])
                if (k == 0) { xk.y = 0.0; xc.y = 0.0; }                                      // C2R ignores imag of DC / Nyquist
                const cd e = cd{0.5 * (xk.x + xc.x), 0.5 * (xk.y + xc.y)};
                const cd o = cmulc(cd{0.5 * (xk.x - xc.x), 0.5 * (xk.y - xc.y)}, sr[r], si[r]);
                v[r] = cd{e.x - o.y, -(e.y + o.x)};               // conj(E + i O)
            }
            fft.run(v, re, im, lane);
            double* dst = frames + (t0 + q) * 1024;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = lane + 64 * r;
                reinterpret_cast<double2*>(dst)[i] = double2{wa[r] * v[r].x, -wb[r] * v[r].y};   // conj, 1/M folded into the window
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// nfft = 1024, hop = 256 (every caller of the reference): inverse FFT AND overlap-add in one kernel, no frame scratch.
// The two-kernel form above writes every windowed frame to HBM in double (8 KB per frame: 307 MB for ten minutes of audio) and
// gathers it back; here a workgroup owns a chunk of IF_K consecutive frames and the IF_K * 256 output samples they complete.
// It computes the chunk's frames plus the three frames in front of it (whose tails reach into the chunk: 10 % more FFTs, no
// exchange between workgroups), one wave per frame, four consecutive frames per round; after each round the four frames sit in
// LDS and all 256 threads add them into the chunk's float output image IN FRAME ORDER with one float rounding per addition --
// exactly the arithmetic of librosa's in-place `y[...] += ytmp` and of istft_ola_kernel (the results are bit-identical).
// Chunk sizes: NPASS staging passes of IF_FR frames (one 8 * IF_FR-byte run per bin), three of the frames halo; short
// utterances take small chunks so that the launch still covers the CUs.  The next pass's S values are requested into registers
// before the current pass's FFT rounds and committed to LDS after them.
constexpr int IF_H = 3;
template <int IF_FR, int NPASS> struct IstftFusedLds {
    static constexpr int R = NPASS * IF_FR, K = R - IF_H;   // frames computed / owned per chunk
    static constexpr int ACC = R * 256 + 768;                  // floats of the output image: frame R - 1 ends at (R - 1) * 256 + 1023
    double ex[4][1152];                                        // per wave: FFT exchange buffers (re: 576, im: 576), then its windowed frame (1024)
    float2 stage[513 * (IF_FR + 1)];                        // IF_FR frames of S: [bin][IF_FR + 1] (S bin-major, one (8 * IF_FR)-byte run per bin)
                                                               // or [frame][516] (S frame-major: whole 4104-byte frames, conflict-free readers)
    float acc[ACC];
    float wss4[256];                                           // window sum of squares of a sample covered by four frames, by src mod hop
};
template <int IF_FR, int NPASS, bool TF>
__global__ __launch_bounds__(256) void istft1024_fused_kernel(const float2* __restrict__ S, int64_t T, int64_t ld,
                                                              const double* __restrict__ window, int64_t start,
                                                              float* __restrict__ y, int64_t out_len) {
    typedef IstftFusedLds<IF_FR, NPASS> LT;
    constexpr int M = 512, F = 513, HOP = 256, NF = 1024, IF_K = LT::K, IF_ACC = LT::ACC;
    constexpr int NPRE = (F * IF_FR + 255) / 256;              // staged values per thread and pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LT& L = *reinterpret_cast<LT*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* re = L.ex[wave];
    double* im = L.ex[wave] + 576;
    double wa[8], wb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { wa[r] = window[2 * (lane + 64 * r)] * (1.0 / M); wb[r] = window[2 * (lane + 64 * r) + 1] * (1.0 / M); }
    Fft512 fft;
    fft.init(lane);
    double sr[8], si[8];                                          // exp(+2 pi i k / 1024), k = lane + 64 r
#pragma unroll
    for (int r = 0; r < 8; ++r) sincospi(2.0 * (double)(lane + 64 * r) / 1024.0, &si[r], &sr[r]);
    const int64_t ntot = (int64_t)NF + (int64_t)HOP * (T - 1);
    const int64_t nchunks = (T + IF_K - 1) / IF_K;
    {   // frames in ascending order: window positions m0 + 768, + 512, + 256, + 0; float rounding after every addition (as the generic loop)
        float w4 = 0.f;
#pragma unroll
        for (int f = 3; f >= 0; --f) { const double w = window[tid + f * HOP]; w4 = (float)((double)w4 + w * w); }
        L.wss4[tid] = w4;
    }
    float2 pre[NPRE];
    // staged value idx of a pass -> (bin, frame of the pass): consecutive threads take consecutive frames of a bin when S is
    // [bin][ld] and consecutive bins of a frame when S is [frame][ld]; `slot`: where (bin, frame) lives in L.stage
    auto split = [&](int idx, int& f, int& q) __attribute__((always_inline)) {
        if (TF) { q = idx / F; f = idx - q * F; } else { f = idx / IF_FR; q = idx - f * IF_FR; }
    };
    auto slot = [&](int f, int q) __attribute__((always_inline)) { return TF ? q * 516 + f : f * (IF_FR + 1) + q; };
    auto request = [&](int64_t ts) __attribute__((always_inline)) {   // IF_FR frames starting at ts (frames outside [0, T): zeros)
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = tid + 256 * u;
            int f, q;
            split(idx, f, q);
            const int64_t t = ts + q;
            pre[u] = (idx < F * IF_FR && t >= 0 && t < T) ? (TF ? S[t * ld + f] : S[(int64_t)f * ld + t]) : float2{0.f, 0.f};
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = tid + 256 * u;
            int f, q;
            split(idx, f, q);
            if (idx < F * IF_FR) L.stage[slot(f, q)] = pre[u];
        }
    };
    if ((int64_t)blockIdx.x < nchunks) request((int64_t)blockIdx.x * IF_K - IF_H);
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t t0 = c * IF_K, tb = t0 - IF_H;              // first own frame, first computed frame (may be < 0)
        __syncthreads();                                          // the previous chunk's output pass is done with acc
        for (int i = tid; i < IF_ACC; i += 256) L.acc[i] = 0.f;
        for (int pass = 0; pass < NPASS; ++pass) {
            const int64_t ts = tb + (int64_t)pass * IF_FR;     // first frame of this staging pass
            commit();                                             // every reader of `stage` passed the barrier that closed the last round
            __syncthreads();
            // next pass (of this chunk or of this workgroup's next chunk): in flight during the FFT rounds
            if (pass + 1 < NPASS) request(ts + IF_FR);
            else if (c + gridDim.x < nchunks) request((c + gridDim.x) * IF_K - IF_H);
            for (int rr = 0; rr < IF_FR / 4; ++rr) {
                const int q = 4 * rr + wave;
                const int64_t t = ts + q;
                const bool valid = t >= 0 && t < T;               // wave-uniform
                if (valid) {
                    cd v[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int k = lane + 64 * r;
                        const float2 a = L.stage[slot(k, q)], b = L.stage[slot(M - k, q)];
                        cd xk = cd{(double)a.x, (double)a.y}, xc = cd{(double)b.x, -(double)b.y};   // X[k], conj(X[M-k])
                        if (k == 0) { xk.y = 0.0; xc.y = 0.0; }                                      // C2R ignores imag of DC / Nyquist
                        const cd e = cd{0.5 * (xk.x + xc.x), 0.5 * (xk.y + xc.y)};
                        const cd o = cmulc(cd{0.5 * (xk.x - xc.x), 0.5 * (xk.y - xc.y)}, sr[r], si[r]);
                        v[r] = cd{e.x - o.y, -(e.y + o.x)};       // conj(E + i O)
                    }
                    fft.run(v, re, im, lane);
                    __builtin_amdgcn_wave_barrier();              // every lane has read its pass-2 inputs: the buffer becomes the frame
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int i = lane + 64 * r;
                        reinterpret_cast<double2*>(L.ex[wave])[i] = double2{wa[r] * v[r].x, -wb[r] * v[r].y};   // conj, 1/M folded into the window
                    }
                }
                __syncthreads();                                  // the round's four frames are in LDS
                // overlap-add of frames ts + 4 rr .. + 3, in frame order, one float rounding per addition
                const int j0 = pass * IF_FR + 4 * rr;          // index of the round's first frame in the chunk
#pragma unroll
                for (int sidx0 = 0; sidx0 < 3 * HOP + NF; sidx0 += 256) {     // seven independent chains per thread (the float <-> double conversions are slow and dependent)
                    const int sidx = sidx0 + tid;
                    float a = L.acc[j0 * HOP + sidx];
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int m = sidx - f * HOP;
                        const int64_t tf = ts + 4 * rr + f;
                        if (m >= 0 && m < NF && tf >= 0 && tf < T) a = (float)((double)a + L.ex[f][m]);
                    }
                    L.acc[j0 * HOP + sidx] = a;
                }
                __syncthreads();                                  // before the next round reuses the exchange buffers (and `stage`, after the last round)
            }
        }
        // output: the samples this chunk completes (the last chunk also owns everything behind its frames)
        const int64_t s_lo = t0 * HOP;
        const bool last = c == nchunks - 1;
        const int64_t s_hi = last ? start + out_len : (t0 + IF_K) * HOP;
        for (int64_t src = s_lo + tid; src < s_hi; src += 256) {
            const int64_t i = src - start;
            if (i < 0 || i >= out_len) continue;
            float a = 0.f, wss = 0.f;
            if (src < ntot) {
                a = L.acc[src - tb * HOP];
                if (src >= NF - HOP && src / HOP <= T - 1) wss = L.wss4[src & (HOP - 1)];   // covered by four frames: the window sum depends on src mod hop only
                else {
                    int64_t tlo = (src - NF + HOP) / HOP;
                    if (src < NF) tlo = 0;
                    int64_t thi = src / HOP;
                    if (thi > T - 1) thi = T - 1;
                    for (int64_t t = tlo; t <= thi; ++t) {
                        const int m = (int)(src - t * HOP);
                        wss = (float)((double)wss + window[m] * window[m]);
                    }
                }
                if (wss > FLT_MIN) a = a / wss;
            }
            y[i] = a;
        }
    }
}

// nfft = 1024, hop = 256, S FRAME-major ([T][ld], row t = frame t): the mirror image of the forward kernel's walk.  A frame is one
// contiguous 4104-byte row, so a wave reads its frame straight into registers (two 512-byte runs per instruction: bins
// lane + 64 r ascending and 512 - lane - 64 r descending) -- no staging through LDS, no workgroup barrier.  Each wave walks
// `chunk` consecutive frames plus the three in front of them and keeps the overlap-add IN REGISTERS: after the inverse FFT lane l
// holds samples 2 l + 128 r + {0, 1} of the frame (r = 0..7), the running float image of the next 1024 output samples lives in
// the same lanes, and advancing one hop (256 samples) is a shift by two registers.  After frame t has been added, samples
// [256 t, 256 t + 256) have received all their frames in frame order with one float rounding per addition -- the arithmetic of
// librosa's `y[...] += ytmp`, of istft_ola_kernel and of the fused kernel above (bit-identical results) -- and leave as 512-byte runs.
// LDS: the FFT exchange buffers only (9 KB per wave).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void istft1024_walk_kernel(const float2* __restrict__ S, int64_t T, int64_t ld,
                                                             const double* __restrict__ window, int64_t start,
                                                             float* __restrict__ y, int64_t out_len, int chunk) {
    constexpr int M = 512, HOP = 256, NF = 1024;
    __shared__ double lre[4][M + 64], lim[4][M + 64];
    // per-bin constants of the whole workgroup in LDS (two waves per SIMD need the kernel under 256 registers):
    // tw[k] = exp(+2 pi i k / 1024); wn[k] = (window[2 k], -window[2 k + 1]) / M  (conj and 1/M folded into the window)
    __shared__ double2 tw[M], wn[M];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform chunk and frame numbers: scalar loop control and addresses
    double* re = lre[wave];
    double* im = lim[wave];
    for (int k = threadIdx.x; k < M; k += 256) {
        double sn, cs;
        sincospi(2.0 * (double)k / 1024.0, &sn, &cs);
        tw[k] = double2{cs, sn};
        wn[k] = double2{window[2 * k] * (1.0 / M), -(window[2 * k + 1] * (1.0 / M))};
    }
    __syncthreads();                                              // the only workgroup barrier
    const int64_t nchunks = (T + chunk - 1) / chunk;
    const int64_t c = (int64_t)blockIdx.x * 4 + wave;
    if (c >= nchunks) return;
    Fft512 fft;
    fft.init(lane);
    // window sum of squares of a sample covered by four frames, at the positions this lane emits (p = 2 lane + 128 j + e within the
    // hop): frames in ascending order see window positions p + 768, + 512, + 256, + 0; float rounding after every addition
    float w4[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float w = 0.f;
#pragma unroll
            for (int f = 3; f >= 0; --f) { const double ww = window[2 * lane + 128 * j + e + f * HOP]; w = (float)((double)w + ww * ww); }
            w4[j][e] = w;
        }
    const int64_t ntot = (int64_t)NF + (int64_t)HOP * (T - 1);
    const int64_t t0 = c * chunk;
    const int64_t te = t0 + chunk < T ? t0 + chunk : T;
    const bool last = c == nchunks - 1;
    const int64_t t_emit_end = last ? T + 3 : te;                 // the last chunk also flushes the three hops behind frame T - 1
    float2 ra[8], rb[8], na[8], nb[8];                            // X[k] and X[512 - k] of the current / the next frame
    auto fetch = [&](int64_t t, float2 (&a)[8], float2 (&b)[8]) __attribute__((always_inline)) {
        const float2* row = S + t * ld;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = row[lane + 64 * r]; b[r] = row[M - lane - 64 * r]; }
    };
    float acc[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; }
    int64_t t = t0 - 3 < 0 ? 0 : t0 - 3;                          // frames in front of the signal do not exist (nothing to add, nothing to emit)
    if (t < T) fetch(t, ra, rb);
    for (; t < t_emit_end; ++t) {
        if (t < T) {
            if (t + 1 < te) fetch(t + 1, na, nb);                 // in flight under this frame's transform
            cd v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = lane + 64 * r;
                cd xk = cd{(double)ra[r].x, (double)ra[r].y}, xc = cd{(double)rb[r].x, -(double)rb[r].y};   // X[k], conj(X[M-k])
                if (k == 0) { xk.y = 0.0; xc.y = 0.0; }                                                      // C2R ignores imag of DC / Nyquist
                const cd e = cd{0.5 * (xk.x + xc.x), 0.5 * (xk.y + xc.y)};
                const double2 w = tw[k];
                const cd o = cmulc(cd{0.5 * (xk.x - xc.x), 0.5 * (xk.y - xc.y)}, w.x, w.y);
                v[r] = cd{e.x - o.y, -(e.y + o.x)};               // conj(E + i O)
            }
            fft.run(v, re, im, lane);
            __builtin_amdgcn_wave_barrier();                      // the next frame's first exchange writes come after every lane's last reads
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                // windowed frame values (conj, 1/M folded into the window) as the doubles the other kernels store, then one
                // float rounding per addition; __dmul_rn / __dadd_rn: never contracted into an fma
                const double2 w = wn[lane + 64 * r];
                acc[r][0] = (float)__dadd_rn((double)acc[r][0], __dmul_rn(w.x, v[r].x));
                acc[r][1] = (float)__dadd_rn((double)acc[r][1], __dmul_rn(w.y, v[r].y));
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { ra[r] = na[r]; rb[r] = nb[r]; }
        }
        if (t >= t0) {
            // samples [256 t, 256 t + 256) are complete
            const bool inner = t >= 3 && t <= T - 1;              // covered by four frames: the window sum depends on the position in the hop only
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int64_t src = t * HOP + 2 * lane + 128 * j + e;
                    float a = acc[j][e], wss = 0.f;
                    if (inner) wss = w4[j][e];
                    else {
                        int64_t tlo = (src - NF + HOP) / HOP;
                        if (src < NF) tlo = 0;
                        int64_t thi = src / HOP;
                        if (thi > T - 1) thi = T - 1;
                        for (int64_t tt = tlo; tt <= thi; ++tt) {
                            const int m = (int)(src - tt * HOP);
                            wss = (float)((double)wss + window[m] * window[m]);
                        }
                    }
                    if (wss > FLT_MIN) a = a / wss;
                    o[e] = a;
                }
                const int64_t i = t * HOP + 2 * lane + 128 * j - start;
                if (i >= 0 && i + 1 < out_len && ((start & 1) == 0)) *reinterpret_cast<float2*>(y + i) = float2{o[0], o[1]};
                else {
                    if (i >= 0 && i < out_len) y[i] = o[0];
                    if (i + 1 >= 0 && i + 1 < out_len) y[i + 1] = o[1];
                }
            }
        }
        // advance one hop: two registers down, zeros in behind
#pragma unroll
        for (int r = 0; r < 6; ++r) { acc[r][0] = acc[r + 2][0]; acc[r][1] = acc[r + 2][1]; }
        acc[6][0] = acc[6][1] = acc[7][0] = acc[7][1] = 0.f;
    }
    if (last)                                                      // behind the signal: zeros up to out_len
        for (int64_t src = ntot + lane; src < start + out_len; src += 64)
            if (src >= start) y[src - start] = 0.f;
}

// The same walk in the arithmetic of istft_pytorch (packages/processing/stft.py:154-190: torch.istft of a complex64 tensor with
// torch.hann_window): inverse FFT, window product, overlap-add and the division by the window envelope in float32 (the kernel above
// computes in double whatever the input: the arithmetic of istft(), where librosa transforms with numpy's double FFT).  What the narrower
// type buys: a point is ONE 8-byte LDS slot, packed float32 VALU instead of fp64 (ten minutes of audio: 73 -> 57-61 us).  Two waves per
// SIMD as the double walk: a round of 2048 waves is two per SIMD whatever the kernel allows, more and shorter chunks transform more halo
// frames and measured slower at every occupancy (profiles/r05_istft_f32_ab.txt; at three waves per SIMD the kernel spills: 72 us; the next
// row by LDS-direct loads instead of 32 registers, spill-free at three and four: 63-75 us at every occupancy -- the walk is bound by its
// VALU / LDS work per transform, not by latency).
#ifndef ISTFT_F32_OCC
#define ISTFT_F32_OCC 2
#endif
__global__ __launch_bounds__(256, ISTFT_F32_OCC) void istft1024_walk_f32_kernel(const float2* __restrict__ S, int64_t T, int64_t ld,
                                                                                const float* __restrict__ window, int64_t start,
                                                                                float* __restrict__ y, int64_t out_len, int chunk) {
    constexpr int M = 512, HOP = 256, NF = 1024;
    __shared__ __attribute__((aligned(8))) cf lz[4][M + 64];
    // tw[k] = exp(+2 pi i k / 1024); wn[k] = (window[2 k], -window[2 k + 1]) / M  (conj and 1/M folded into the window: exact scalings)
    __shared__ float2 tw[M], wn[M];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cf* z = lz[wave];
    for (int k = threadIdx.x; k < M; k += 256) {
        double sn, cs;
        sincospi(2.0 * (double)k / 1024.0, &sn, &cs);
        tw[k] = float2{(float)cs, (float)sn};
        wn[k] = float2{window[2 * k] * (1.0f / M), -(window[2 * k + 1] * (1.0f / M))};
    }
    __syncthreads();                                              // the only workgroup barrier
    const int64_t nchunks = (T + chunk - 1) / chunk;
    const int64_t c = (int64_t)blockIdx.x * 4 + wave;
    if (c >= nchunks) return;
    Fft512F fft;
    fft.init(lane);
    // window envelope of a sample covered by four frames, at the positions this lane emits (frames in ascending order)
    float w4[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float w = 0.f;
#pragma unroll
            for (int f = 3; f >= 0; --f) { const float ww = window[2 * lane + 128 * j + e + f * HOP]; w = fmaf(ww, ww, w); }
            w4[j][e] = w;
        }
    const int64_t ntot = (int64_t)NF + (int64_t)HOP * (T - 1);
    const int64_t t0 = c * chunk;
    const int64_t te = t0 + chunk < T ? t0 + chunk : T;
    const bool last = c == nchunks - 1;
    const int64_t t_emit_end = last ? T + 3 : te;                 // the last chunk also flushes the three hops behind frame T - 1
    float2 ra[8], rb[8], na[8], nb[8];                            // X[k] and X[512 - k] of the current / the next frame
    // one descriptor, two per-lane offsets, the frame's row as the scalar offset, the bin group as the instruction offset (sixteen 64-bit
    // addresses per frame cost 32 registers); the launcher checks T * ld * 8 < 2^31
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(S), 0, (int)(T * ld * 8), 0x00020000);
    const int va = lane * 8, vb = (64 - lane) * 8;
    auto fetch = [&](int64_t t, float2 (&a)[8], float2 (&b)[8]) __attribute__((always_inline)) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const int so = (int)(t * ld * 8);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            a[r] = __builtin_bit_cast(float2, (u2)__builtin_amdgcn_raw_buffer_load_b64(rs_s, va + 512 * r, so, 0));
            b[r] = __builtin_bit_cast(float2, (u2)__builtin_amdgcn_raw_buffer_load_b64(rs_s, vb + 512 * (7 - r), so, 0));
        }
    };
    float acc[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; }
    int64_t t = t0 - 3 < 0 ? 0 : t0 - 3;
    if (t < T) fetch(t, ra, rb);
    for (; t < t_emit_end; ++t) {
        if (t < T) {
            if (t + 1 < te) fetch(t + 1, na, nb);                 // in flight under this frame's transform
            cf v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = lane + 64 * r;
                cf xk = cf{ra[r].x, ra[r].y}, xc = cf{rb[r].x, -rb[r].y};                        // X[k], conj(X[M-k])
                if (k == 0) { xk.y = 0.f; xc.y = 0.f; }                                           // C2R ignores imag of DC / Nyquist
                const cf e = cf{0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y)};
                const float2 w = tw[k];
                const cf o = cfmulc(cf{0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y)}, w.x, w.y);
                v[r] = cf{e.x - o.y, -(e.y + o.x)};               // conj(E + i O)
            }
            fft.run(v, z, lane);
            __builtin_amdgcn_wave_barrier();                      // the next frame's first exchange writes come after every lane's last reads
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 w = wn[lane + 64 * r];
                acc[r][0] = fmaf(w.x, v[r].x, acc[r][0]);
                acc[r][1] = fmaf(w.y, v[r].y, acc[r][1]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { ra[r] = na[r]; rb[r] = nb[r]; }
        }
        if (t >= t0) {
            // samples [256 t, 256 t + 256) are complete
            const bool inner = t >= 3 && t <= T - 1;              // covered by four frames: the envelope depends on the position in the hop only
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int64_t src = t * HOP + 2 * lane + 128 * j + e;
                    float a = acc[j][e], wss = 0.f;
                    if (inner) wss = w4[j][e];
                    else {
                        int64_t tlo = (src - NF + HOP) / HOP;
                        if (src < NF) tlo = 0;
                        int64_t thi = src / HOP;
                        if (thi > T - 1) thi = T - 1;
                        for (int64_t tt = tlo; tt <= thi; ++tt) {
                            const float ww = window[(int)(src - tt * HOP)];
                            wss = fmaf(ww, ww, wss);
                        }
                    }
                    if (wss > 1e-11f) a = a / wss;                // torch.istft: window_envelop.abs() > 1e-11 is asserted over the kept range
                    o[e] = a;
                }
                const int64_t i = t * HOP + 2 * lane + 128 * j - start;
                if (i >= 0 && i + 1 < out_len && ((start & 1) == 0)) *reinterpret_cast<float2*>(y + i) = float2{o[0], o[1]};
                else {
                    if (i >= 0 && i < out_len) y[i] = o[0];
                    if (i + 1 >= 0 && i + 1 < out_len) y[i + 1] = o[1];
                }
            }
        }
        // advance one hop: two registers down, zeros in behind
#pragma unroll
        for (int r = 0; r < 6; ++r) { acc[r][0] = acc[r + 2][0]; acc[r][1] = acc[r + 2][1]; }
        acc[6][0] = acc[6][1] = acc[7][0] = acc[7][1] = 0.f;
    }
    if (last)                                                      // behind the signal: zeros up to out_len
        for (int64_t src = ntot + lane; src < start + out_len; src += 64)
            if (src >= start) y[src - start] = 0.f;
}

// [513][ld] (bin-major rows, the legacy layout) -> [T][513] frame rows for the walk kernel: 64 x 64 tiles through LDS, 512-byte runs both ways
constexpr int64_t ISTFT_TR_MIN_T = 1024;      // shorter spectrograms go through the staged kernel directly (a second launch costs more than it saves)
__global__ __launch_bounds__(256) void c64_transpose_kernel(const float2* __restrict__ S, int64_t T, int64_t ld, float2* __restrict__ out) {
    __shared__ float2 tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    const int b0 = blockIdx.y * 64;
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int b = b0 + ty + 4 * p;
        if (b < 513 && t0 + tx < T) tile[ty + 4 * p][tx] = S[(int64_t)b * ld + t0 + tx];
    }
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int64_t t = t0 + ty + 4 * p;
        if (t < T && b0 + tx < 513) out[t * 513 + b0 + tx] = tile[tx][ty + 4 * p];
    }
}

template <int IF_FR, int NPASS, bool TF>
static int launch_istft_fused(const float2* S, int64_t T, int64_t ld, const double* window, int64_t start, float* y, int64_t out_len, hipStream_t s) {
    typedef IstftFusedLds<IF_FR, NPASS> LT;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        DVAE_HIP(hipFuncSetAttribute((const void*)(istft1024_fused_kernel<IF_FR, NPASS, TF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LT)));
        attr_done[dev] = true;
    }
    const int64_t nchunks = cdiv(T, LT::K);
    const int per_cu = sizeof(LT) <= 80 * 1024 ? 2 : 1;           // workgroups resident per CU (LDS)
    const int wb = (int)(nchunks < 256 * per_cu ? nchunks : 256 * per_cu);
    hipLaunchKernelGGL((istft1024_fused_kernel<IF_FR, NPASS, TF>), dim3(wb), dim3(256), sizeof(LT), s, S, T, ld, window, start, y, out_len);
    DVAE_LAUNCH_OK("istft1024_fused_kernel");
    return 0;
}

// frames[t][m] = window[m] * irfft(S[:, t])[m]   (double scratch)
__global__ __launch_bounds__(256) void istft_frames_pow2_kernel(const float* __restrict__ S, int64_t T, int64_t sf, int64_t st,
                                                                 const double* __restrict__ window, int nfft, int logM,
                                                                 double* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = nfft >> 1;
    cd* z = (cd*)smem;
    cd* tw = z + M;
    double* win = (double*)(tw + M);
    stage_tables(tw, win, window, nfft);
    __syncthreads();
    const double scale = 1.0 / (double)M;
    for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
        for (int k = threadIdx.x; k <= (M >> 1); k += blockDim.x) {
            cd xk = cd{(double)S[(k * sf + t * st) * 2], (double)S[(k * sf + t * st) * 2 + 1]};
            cd xm = cd{(double)S[((int64_t)(M - k) * sf + t * st) * 2], (double)S[((int64_t)(M - k) * sf + t * st) * 2 + 1]};
            if (k == 0) { xk.y = 0.0; xm.y = 0.0; }              // C2R ignores imag of DC / Nyquist
            const cd xc = cconj(xm);
            const cd e = cd{0.5 * (xk.x + xc.x), 0.5 * (xk.y + xc.y)};
            const cd d = csub(xk, xc);
            const cd o = cmul(cconj(tw[k]), cd{0.5 * d.x, 0.5 * d.y});
            const cd zk = cd{e.x - o.y, e.y + o.x};              // E + i O
            const cd zm = cd{e.x + o.y, -e.y + o.x};             // conj(E) + i conj(O)
            z[__brev((unsigned)k) >> (32 - logM)] = zk;
            if (k != 0 && k != (M >> 1)) z[__brev((unsigned)(M - k)) >> (32 - logM)] = zm;
        }
        __syncthreads();
        fft_lds(z, tw, logM, 1);
        for (int i = threadIdx.x; i < M; i += blockDim.x) {
            const cd v = z[i];
            frames[t * nfft + 2 * i] = win[2 * i] * (v.x * scale);
            frames[t * nfft + 2 * i + 1] = win[2 * i + 1] * (v.y * scale);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void istft_frames_dft_kernel(const float* __restrict__ S, int64_t T, int64_t sf, int64_t st,
                                                                const double* __restrict__ window, int nfft,
                                                                double* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cd* tw = (cd*)smem;                  // exp(+2 pi i k / nfft)
    cd* X = tw + nfft;                   // half spectrum of this frame
    const int F = nfft / 2 + 1;
    for (int k = threadIdx.x; k < nfft; k += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)k / (double)nfft, &s, &c);
        tw[k] = cd{c, s};
    }
    for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
        __syncthreads();
        for (int f = threadIdx.x; f < F; f += blockDim.x) {
            cd v = cd{(double)S[(f * sf + t * st) * 2], (double)S[(f * sf + t * st) * 2 + 1]};
            if (f == 0 || (2 * f == nfft)) v.y = 0.0;
            X[f] = v;
        }
        __syncthreads();
        for (int m = threadIdx.x; m < nfft; m += blockDim.x) {
            double acc = X[0].x;
            int idx = 0;
            for (int f = 1; f < F; ++f) {
                idx += m; if (idx >= nfft) idx -= nfft;
                const double term = X[f].x * tw[idx].x - X[f].y * tw[idx].y;
                acc += (2 * f == nfft) ? term : 2.0 * term;
            }
            frames[t * nfft + m] = window[m] * (acc / (double)nfft);
        }
    }
}

// y[i] = sum over frames (float32 accumulation in frame order, as librosa's in-place +=) / wss
__global__ __launch_bounds__(256) void istft_ola_kernel(const double* __restrict__ frames, const double* __restrict__ window,
                                                         int64_t T, int nfft, int hop, int64_t start, float* __restrict__ y, int64_t out_len) {
    const int64_t ntot = (int64_t)nfft + (int64_t)hop * (T - 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < out_len; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = i + start;
        float acc = 0.f, wss = 0.f;
        if (src < ntot) {
            int64_t tlo = (src - nfft + hop) / hop;     // ceil((src - nfft + 1) / hop) for src >= nfft - 1
            if (src < nfft) tlo = 0;
            int64_t thi = src / hop;
            if (thi > T - 1) thi = T - 1;
            for (int64_t t = tlo; t <= thi; ++t) {
                const int m = (int)(src - t * hop);
                acc = (float)((double)acc + frames[t * nfft + m]);
                wss = (float)((double)wss + window[m] * window[m]);
            }
            if (wss > FLT_MIN) acc = acc / wss;
        }
        y[i] = acc;
    }
}

static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_stft(const void* x, int in_f64, int64_t n, const double* window, int nfft, int hop,
                         int64_t T, void* out, int layout, void* stream) {
    DVAE_CHECK_ARG(x && window && out && n > 0 && nfft >= 4 && (nfft % 2) == 0 && hop > 0 && T >= 0, "stft: bad argument");
    DVAE_CHECK_ARG(layout >= 0 && layout <= 2, "stft: unknown output layout %d", layout);
    DVAE_CHECK_ARG(T == 0 || (T - 1) * (int64_t)hop + nfft <= n, "stft: %lld frames do not fit in %lld samples", (long long)T, (long long)n);
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)(T < 2048 ? T : 2048);
    const int lg = ilog2_exact(nfft);
    static const bool legacy = getenv("DVAE_STFT_LEGACY") != nullptr;       // A/B switch for the workgroup-per-frame kernel
    if (nfft == 1024 && !legacy) {
        if (layout != 0) {
            // one round of waves: 256 CUs x 4 SIMDs x 2 resident waves = 2048 slots; a wave takes ceil(T / 2048) frames
            int chunk = (int)cdiv(T, 2048);
            chunk = chunk < 1 ? 1 : chunk;
            const int wb = (int)cdiv(T, (int64_t)4 * chunk);
            // hop 256 and everything addressable with 32-bit byte offsets: the walk with buffer addressing and the register ring
            static const bool oldwalk = getenv("DVAE_STFT_WALK") != nullptr && !strcmp(getenv("DVAE_STFT_WALK"), "r3");
            const bool ring = hop == 256 && !oldwalk && n * (in_f64 ? 8 : 4) < ((int64_t)1 << 31) && T * 513 * (layout == 1 ? 4 : 8) < ((int64_t)1 << 31);
            if (ring) {
                // DVAE_STFT_OCC=3 (diagnostic build): three waves per SIMD with the window and two twiddle tables read from LDS every frame
                // (166 registers, 50 KB of LDS per workgroup) -- measured SLOWER, 63.3 / 64.9 us against 57.8 / 61.4 (complex / power frames,
                // ten minutes of float64 audio, alternating on one box): the 19 extra ds_read_b128 per frame cost more than the third wave hides
                static const bool occ3 = kDiagBuild && getenv("DVAE_STFT_OCC") != nullptr && atoi(getenv("DVAE_STFT_OCC")) == 3;
                if (occ3) {
                    if constexpr (kDiagBuild) {
                    int chunk3 = (int)cdiv(T, 3072);                   // one round of 256 CUs x 4 SIMDs x 3 resident waves
                    chunk3 = chunk3 < 1 ? 1 : chunk3;
                    const int wb3 = (int)cdiv(T, (int64_t)4 * chunk3);
                    if (layout == 1) {
                        if (in_f64) hipLaunchKernelGGL((stft1024_walk_kernel<double, true, true>), dim3(wb3), dim3(256), 0, s, (const double*)x, n, window, T, chunk3, out);
                        else hipLaunchKernelGGL((stft1024_walk_kernel<float, true, true>), dim3(wb3), dim3(256), 0, s, (const float*)x, n, window, T, chunk3, out);
                    } else {
                        if (in_f64) hipLaunchKernelGGL((stft1024_walk_kernel<double, false, true>), dim3(wb3), dim3(256), 0, s, (const double*)x, n, window, T, chunk3, out);
                        else hipLaunchKernelGGL((stft1024_walk_kernel<float, false, true>), dim3(wb3), dim3(256), 0, s, (const float*)x, n, window, T, chunk3, out);
                    }
                    }
                } else if (layout == 1) {
                    if (in_f64) hipLaunchKernelGGL((stft1024_walk_kernel<double, true, false>), dim3(wb), dim3(256), 0, s, (const double*)x, n, window, T, chunk, out);
                    else hipLaunchKernelGGL((stft1024_walk_kernel<float, true, false>), dim3(wb), dim3(256), 0, s, (const float*)x, n, window, T, chunk, out);
                } else {
                    if (in_f64) hipLaunchKernelGGL((stft1024_walk_kernel<double, false, false>), dim3(wb), dim3(256), 0, s, (const double*)x, n, window, T, chunk, out);
                    else hipLaunchKernelGGL((stft1024_walk_kernel<float, false, false>), dim3(wb), dim3(256), 0, s, (const float*)x, n, window, T, chunk, out);
                }
            } else if (layout == 1) {
                if (in_f64) hipLaunchKernelGGL((stft1024_kernel<double, 1>), dim3(wb), dim3(256), 0, s, (const double*)x, n, window, hop, T, chunk, out);
                else hipLaunchKernelGGL((stft1024_kernel<float, 1>), dim3(wb), dim3(256), 0, s, (const float*)x, n, window, hop, T, chunk, out);
            } else {
                if (in_f64) hipLaunchKernelGGL((stft1024_kernel<double, 2>), dim3(wb), dim3(256), 0, s, (const double*)x, n, window, hop, T, chunk, out);
                else hipLaunchKernelGGL((stft1024_kernel<float, 2>), dim3(wb), dim3(256), 0, s, (const float*)x, n, window, hop, T, chunk, out);
            }
        } else {
            const int wb = (int)(cdiv(T, STFT_FR) < 2048 ? cdiv(T, STFT_FR) : 2048);
            const size_t lds = (size_t)513 * (STFT_FR + 1) * sizeof(float2);
            static bool attr_done = false;
            if (!attr_done) {
                DVAE_HIP(hipFuncSetAttribute((const void*)stft1024_kernel<double, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                DVAE_HIP(hipFuncSetAttribute((const void*)stft1024_kernel<float, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_done = true;
            }
            if (in_f64) hipLaunchKernelGGL((stft1024_kernel<double, 0>), dim3(wb), dim3(256), lds, s, (const double*)x, n, window, hop, T, 1, out);
            else hipLaunchKernelGGL((stft1024_kernel<float, 0>), dim3(wb), dim3(256), lds, s, (const float*)x, n, window, hop, T, 1, out);
        }
    } else if (lg >= 3 && nfft <= 2048) {
        const size_t lds = (size_t)(nfft / 2) * 2 * sizeof(cd) + (size_t)nfft * sizeof(double);
        if (in_f64)
            hipLaunchKernelGGL((stft_pow2_kernel<double>), dim3(blocks), dim3(256), lds, s, (const double*)x, n, window, nfft, lg - 1, hop, T, out, layout);
        else
            hipLaunchKernelGGL((stft_pow2_kernel<float>), dim3(blocks), dim3(256), lds, s, (const float*)x, n, window, nfft, lg - 1, hop, T, out, layout);
    } else {
        DVAE_CHECK_ARG(nfft <= 2048, "stft: window length %d not supported (max 2048)", nfft);
        const size_t lds = (size_t)nfft * sizeof(cd) + (size_t)nfft * sizeof(double);
        if (in_f64)
            hipLaunchKernelGGL((stft_dft_kernel<double>), dim3(blocks), dim3(256), lds, s, (const double*)x, n, window, nfft, hop, T, out, layout);
        else
            hipLaunchKernelGGL((stft_dft_kernel<float>), dim3(blocks), dim3(256), lds, s, (const float*)x, n, window, nfft, hop, T, out, layout);
    }
    DVAE_LAUNCH_OK("stft");
    return 0;
}

extern "C" int dvae_stft_f32(const float* x, int64_t n, const float* window, int nfft, int hop, int64_t T, void* out, int layout, void* stream) {
    DVAE_CHECK_ARG(x && window && out && n > 0 && T >= 0, "stft_f32: bad argument");
    DVAE_CHECK_ARG(nfft == 1024 && hop == 256, "stft_f32: the float32-arithmetic transform exists for nfft 1024 / hop 256 (got %d / %d): use dvae_stft", nfft, hop);
    DVAE_CHECK_ARG(layout == 1 || layout == 2, "stft_f32: frame-major layouts only (1 power frames, 2 complex frames), got %d", layout);
    DVAE_CHECK_ARG(T == 0 || (T - 1) * (int64_t)hop + nfft <= n, "stft_f32: %lld frames do not fit in %lld samples", (long long)T, (long long)n);
    DVAE_CHECK_ARG(n * 4 < ((int64_t)1 << 31) && T * 513 * (layout == 1 ? 4 : 8) < ((int64_t)1 << 31), "stft_f32: signal or spectrogram beyond 2 GB (32-bit buffer offsets)");
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // one round of waves: 256 CUs x 4 SIMDs x STFT_F32_OCC resident waves
    int chunk = (int)cdiv(T, (int64_t)1024 * STFT_F32_OCC);
    chunk = chunk < 1 ? 1 : chunk;
    const int wb = (int)cdiv(T, (int64_t)4 * chunk);
    if (layout == 1) hipLaunchKernelGGL((stft1024_walk_f32_kernel<true>), dim3(wb), dim3(256), 0, s, x, n, window, T, chunk, out);
    else hipLaunchKernelGGL((stft1024_walk_f32_kernel<false>), dim3(wb), dim3(256), 0, s, x, n, window, T, chunk, out);
    DVAE_LAUNCH_OK("stft1024_walk_f32_kernel");
    return 0;
}

extern "C" size_t dvae_istft_workspace_bytes(int64_t T, int nfft) {
    return (size_t)(T > 0 ? T : 0) * (size_t)nfft * sizeof(double);
}

// the 1024 / 256 transform (every caller of the reference) runs as one kernel and needs no frame scratch
extern "C" size_t dvae_istft_workspace_bytes_hop(int64_t T, int nfft, int hop) {
    if (nfft == 1024 && hop == 256 && getenv("DVAE_STFT_LEGACY") == nullptr && getenv("DVAE_ISTFT_2PASS") == nullptr)
        return T >= ISTFT_TR_MIN_T ? (size_t)T * 513 * sizeof(float2) + 16 : 16;      // long bin-major input: its frame-major copy
    return dvae_istft_workspace_bytes(T, nfft);
}

// S(bin f, frame t) = S[f * ld + t] (tf = false: bin-major, ld >= T) or S[t * ld + f] (tf = true: frame-major, ld >= nfft / 2 + 1)
static int istft_run(const void* S, int64_t T, int64_t ld, bool tf, const double* window, int nfft, int hop,
                     int64_t start, float* y, int64_t out_len, void* ws, void* stream) {
    DVAE_CHECK_ARG(S && window && y && ws && T > 0 && nfft >= 4 && (nfft % 2) == 0 && hop > 0 && start >= 0 && out_len >= 0,
                   "istft: bad argument");
    DVAE_CHECK_ARG(ld >= (tf ? (int64_t)(nfft / 2 + 1) : T), "istft: leading dimension %lld too small", (long long)ld);
    DVAE_CHECK_ARG(nfft <= 2048, "istft: window length %d not supported (max 2048)", nfft);
    hipStream_t s = (hipStream_t)stream;
    const int64_t sf = tf ? 1 : ld, st = tf ? ld : 1;
    const int blocks = (int)(T < 2048 ? T : 2048);
    const int lg = ilog2_exact(nfft);
    static const bool legacy = getenv("DVAE_STFT_LEGACY") != nullptr;
    const bool two_pass = getenv("DVAE_ISTFT_2PASS") != nullptr;             // A/B switch (read per call): frames to scratch + gather overlap-add
    if (nfft == 1024 && hop == 256 && !legacy && !two_pass) {
        if (out_len == 0) return 0;
        // chunk size by length: enough chunks to cover the CUs first, then the least halo work (3 of 8 / 16 / 32 frames)
        // chunk size by length: enough chunks to cover the CUs first, then wider runs per bin and less halo work (3 of 8 / 16 / 32 frames)
        if (tf && getenv("DVAE_ISTFT_STAGED") == nullptr) {                  // A/B switch (read per call): frame-major input through the staged kernel
            // one round of waves (2048 slots, as the forward transform): the shortest walk per wave, ceil(T / 2048) own frames + 3 halo
            // frames (short utterances: 4 transforms for 1 own frame, all waves side by side -- 12 us at 309 frames against 19 us with 4 own)
            const char* const slots_s = getenv("DVAE_ISTFT_SLOTS");          // experiment switch: waves the frames are dealt to (default: one round of 2048)
            const int slots = slots_s && atoi(slots_s) >= 64 ? atoi(slots_s) : 2048;
            int chunk = (int)cdiv(T, slots);
            chunk = chunk < 1 ? 1 : chunk;
            const int wb = (int)cdiv(cdiv(T, chunk), 4);
            hipLaunchKernelGGL(istft1024_walk_kernel, dim3(wb), dim3(256), 0, s, (const float2*)S, T, ld, window, start, y, out_len, chunk);
            DVAE_LAUNCH_OK("istft1024_walk_kernel");
            return 0;
        }
        if (!tf && T >= ISTFT_TR_MIN_T && getenv("DVAE_ISTFT_STAGED") == nullptr) {
            // long bin-major spectrograms: one transposing pass into the workspace, then the frame-major walk (ten minutes of audio:
            // 88 + 81 us against the staged kernel's 207; the same arithmetic, bit-identical)
            hipLaunchKernelGGL(c64_transpose_kernel, dim3((unsigned)cdiv(T, 64), 9), dim3(256), 0, s, (const float2*)S, T, ld, (float2*)ws);
            DVAE_LAUNCH_OK("c64_transpose_kernel");
            int chunk = (int)cdiv(T, 2048);
            chunk = chunk < 1 ? 1 : chunk;
            const int wb = (int)cdiv(cdiv(T, chunk), 4);
            hipLaunchKernelGGL(istft1024_walk_kernel, dim3(wb), dim3(256), 0, s, (const float2*)ws, T, (int64_t)513, window, start, y, out_len, chunk);
            DVAE_LAUNCH_OK("istft1024_walk_kernel");
            return 0;
        }
        if (tf) {                                                                // frame-major input through the staged kernel: diagnostic builds (DVAE_ISTFT_STAGED)
#ifdef DVAE_DIAG
            if (T <= 5 * 512) return launch_istft_fused<8, 1, true>((const float2*)S, T, ld, window, start, y, out_len, s);
            if (T <= 13 * 512) return launch_istft_fused<16, 1, true>((const float2*)S, T, ld, window, start, y, out_len, s);
            return launch_istft_fused<16, 2, true>((const float2*)S, T, ld, window, start, y, out_len, s);
#else
            set_error("istft: DVAE_ISTFT_STAGED on frame-major input exists in the diagnostic build only (build.py --diag)");
            return DVAE_E_UNSUPPORTED;
#endif
        }
        if (T <= 5 * 512) return launch_istft_fused<8, 1, false>((const float2*)S, T, ld, window, start, y, out_len, s);
        if (T <= 13 * 512) return launch_istft_fused<16, 1, false>((const float2*)S, T, ld, window, start, y, out_len, s);
        return launch_istft_fused<16, 2, false>((const float2*)S, T, ld, window, start, y, out_len, s);
    }
    if (nfft == 1024 && !legacy) {
        const size_t lds = (size_t)513 * (ISTFT_FR + 1) * sizeof(float2);
        static bool attr_done = false;
        if (!attr_done) {
            DVAE_HIP(hipFuncSetAttribute((const void*)istft1024_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_done = true;
        }
        const int wb = (int)(cdiv(T, ISTFT_FR) < 4096 ? cdiv(T, ISTFT_FR) : 4096);
        hipLaunchKernelGGL(istft1024_frames_kernel, dim3(wb), dim3(256), lds, s, (const float2*)S, T, sf, st, window, (double*)ws);
    } else if (lg >= 3) {
        const size_t lds = (size_t)(nfft / 2) * 2 * sizeof(cd) + (size_t)nfft * sizeof(double);
        hipLaunchKernelGGL(istft_frames_pow2_kernel, dim3(blocks), dim3(256), lds, s, (const float*)S, T, sf, st, window, nfft, lg - 1, (double*)ws);
    } else {
        const size_t lds = (size_t)nfft * sizeof(cd) + (size_t)(nfft / 2 + 1) * sizeof(cd);
        hipLaunchKernelGGL(istft_frames_dft_kernel, dim3(blocks), dim3(256), lds, s, (const float*)S, T, sf, st, window, nfft, (double*)ws);
    }
    DVAE_LAUNCH_OK("istft_frames");
    if (out_len == 0) return 0;
    const int ob = (int)(cdiv(out_len, 256) < 2048 ? cdiv(out_len, 256) : 2048);
    hipLaunchKernelGGL(istft_ola_kernel, dim3(ob), dim3(256), 0, s, (const double*)ws, window, T, nfft, hop, start, y, out_len);
    DVAE_LAUNCH_OK("istft_ola");
    return 0;
}

extern "C" int dvae_istft(const void* S, int64_t T, int64_t ldT, const double* window, int nfft, int hop,
                          int64_t start, float* y, int64_t out_len, void* ws, void* stream) {
    return istft_run(S, T, ldT, false, window, nfft, hop, start, y, out_len, ws, stream);
}

extern "C" int dvae_istft_frames(const void* S, int64_t T, int64_t ldF, const double* window, int nfft, int hop,
                                 int64_t start, float* y, int64_t out_len, void* ws, void* stream) {
    return istft_run(S, T, ldF, true, window, nfft, hop, start, y, out_len, ws, stream);
}

// float32-arithmetic inverse transform (istft_pytorch): S bin-major ([513][ld], frames = 0: transposed into ws first, T * 513 complex64)
// or frame-major ([T][ld], frames = 1: read in place, ws unused)
extern "C" int dvae_istft_f32(const void* S, int64_t T, int64_t ld, int frames, const float* window, int nfft, int hop,
                              int64_t start, float* y, int64_t out_len, void* ws, void* stream) {
    DVAE_CHECK_ARG(S && window && y && T > 0 && start >= 0 && out_len >= 0, "istft_f32: bad argument");
    DVAE_CHECK_ARG(nfft == 1024 && hop == 256, "istft_f32: window length 1024 / hop 256 only (got %d / %d): use dvae_istft", nfft, hop);
    DVAE_CHECK_ARG(ld >= (frames ? (int64_t)513 : T), "istft_f32: leading dimension %lld too small", (long long)ld);
    DVAE_CHECK_ARG(frames || ws, "istft_f32: bin-major input needs the workspace (T * 513 complex64)");
    DVAE_CHECK_ARG(T * (frames ? ld : (int64_t)513) * 8 < ((int64_t)1 << 31), "istft_f32: spectrograms of 2 GB and more are not addressed (use dvae_istft)");
    hipStream_t s = (hipStream_t)stream;
    if (out_len == 0) return 0;
    const float2* Sf = (const float2*)S;
    int64_t ldf = ld;
    if (!frames) {
        hipLaunchKernelGGL(c64_transpose_kernel, dim3((unsigned)cdiv(T, 64), 9), dim3(256), 0, s, (const float2*)S, T, ld, (float2*)ws);
        DVAE_LAUNCH_OK("c64_transpose_kernel");
        Sf = (const float2*)ws;
        ldf = 513;
    }
    const char* const slots_s = getenv("DVAE_ISTFT_SLOTS");          // experiment switch, as in the double walk
    const int slots = slots_s && atoi(slots_s) >= 64 ? atoi(slots_s) : 2048;
    int chunk = (int)cdiv(T, slots);                               // one round of waves, as the double walk
    chunk = chunk < 1 ? 1 : chunk;
    const int wb = (int)cdiv(cdiv(T, chunk), 4);
    hipLaunchKernelGGL(istft1024_walk_f32_kernel, dim3(wb), dim3(256), 0, s, Sf, T, ldf, window, start, y, out_len, chunk);
    DVAE_LAUNCH_OK("istft1024_walk_f32_kernel");
    return 0;
}
