// Round 5: the k-step micro-benchmark of tools/r03/kstep_bench.hip with ONE more switch -- 2048: every weight fragment pair feeds TWO 32-frame
// operand sets (a 64-frame tile: twice the LDS operand reads and MFMAs per 2 KB of weights pulled into the CU, two accumulators).  What it
// answers: does the k-step of a 64-frame rows kernel become MFMA-bound (6 x 32 = 192 clocks for twice the frames) as DESIGN section 5
// estimates, or does something else bind first?  (The LDS image here is 64 frames x 552 columns x 2 planes = 141 KB.)
// Micro-benchmark of the rows kernel's k-step outside the kernel: one workgroup per CU, four waves (one per
// SIMD), each wave streaming 2 KB of "weight" fragments per k-step from an L2-resident buffer through a register ring of depth D,
// reading its B operand (2 x 1 KB) from LDS and issuing the k-step's MFMAs.  What is switched on is a bit mask:
//   1 weight loads   2 LDS operand reads   4 MFMAs   8 MFMAs on three independent accumulators (instead of one dependent chain)
//   16 MFMAs as 16x16x32 tiles (four per 32x32x16's worth of work; a quarter of the result bytes per instruction)
//   32 weight loads as one plane only (1 KB per k-step)   64 a second wave per SIMD (8 waves per workgroup)
// Output: ns per k-step (median over workgroups, wall clock 100 MHz) and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/r03/kstep_bench.hip -o gpurun_out/kstep_bench && gpurun_out/kstep_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef PLANE_STRIDE
#define PLANE_STRIDE (2u * 1048576u + 4352u)
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int D>
__global__ __launch_bounds__(512, 1) void kstep_kernel(const void* __restrict__ w, unsigned wbytes, int nsteps, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool WL = MODE & 1, BL = MODE & 2, MF = MODE & 4, IND = MODE & 8, SMALL = MODE & 16, ONEP = MODE & 32, F64 = MODE & 2048;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: a divergent soffset puts a waterfall loop around every load
    // LDS image: 32 frames x 552 columns x 2 planes of bf16 (as U of the rows kernel)
    for (int i = threadIdx.x; i < (F64 ? 2 : 1) * 2 * 32 * 552 / 2; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u + i;
    __syncthreads();
    // MODE & 256: the per-lane offset (lane * 16) comes from the buffer descriptor (ADD_TID_ENABLE, stride 16) instead of a VGPR
    constexpr bool TID = MODE & 256;
    // (with ADD_TID_ENABLE the DATA_FORMAT field of word 3 is read as stride bits 17:14: it must be zero, or lane 63 lands 4 MB away)
    const __amdgpu_buffer_rsrc_t rs = TID ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 16, (int)wbytes, (1 << 23))
                                          : __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, (int)wbytes, 0x00020000);
    const int voff = TID ? 0 : lane * 16;
    // fragment (tile = wave & 3, k-step s): byte offset ((s * 4 + tile) * 1024), lo plane at + wbytes / 2
    // second wave of a SIMD (waves 4-7): its own fragments, 216 k-steps further on (never the lines its partner has just pulled into L1)
    const unsigned tile_off = (unsigned)(wave & 3) * 1024u + (unsigned)(wave >> 2) * 216u * 4096u;
    const unsigned plane = PLANE_STRIDE;                          // not a power of two (as the kernel's weight-copy planes)
    const unsigned span = 216u;                                   // k-steps before wrapping: the 216 k-steps of a tile (0.86 MB per plane)
    bf16x8 ring[D][2];
    auto wload = [&](int s, bf16x8 (&r)[2]) __attribute__((always_inline)) {
        const unsigned so = (unsigned)(s % (int)span) * 4096u + tile_off;
        if constexpr (MODE & 512) {                                  // global_load_dwordx4 (scalar base + per-lane offset) instead of buffer loads
            const char* base = reinterpret_cast<const char*>(w) + so;
            r[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + lane * 16));
            r[1] = ONEP ? r[0] : __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + plane + lane * 16));
            return;
        }
        r[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)so, 0));
        if (!ONEP) r[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)(so + plane), 0));
        else r[1] = r[0];
    };
    const __bf16* brow = reinterpret_cast<const __bf16*>(smem) + (lane & 31) * 552 + (lane >> 5) * 8;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    auto bload = [&](int s, bf16x8 (&b)[2]) __attribute__((always_inline)) {
        const int c = (s % 33) * 16;
        if constexpr (MODE & 128) {                                  // the same 2 x 16 bytes per lane as four 8-byte LDS reads
            const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(brow + c), a1 = *reinterpret_cast<const bf16x4*>(brow + c + 4);
            const bf16x4 b0 = *reinterpret_cast<const bf16x4*>(brow + 32 * 552 + c), b1 = *reinterpret_cast<const bf16x4*>(brow + 32 * 552 + c + 4);
            b[0] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            b[1] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        } else {
            b[0] = *reinterpret_cast<const bf16x8*>(brow + c);
            b[1] = *reinterpret_cast<const bf16x8*>(brow + 32 * 552 + c);
        }
    };
    // frames 32 .. 63 of a 64-frame tile: a second image behind the first (2 x 32 x 552 elements further on)
    auto bload2 = [&](int s, bf16x8 (&b)[2]) __attribute__((always_inline)) {
        const int c = (s % 33) * 16;
        b[0] = *reinterpret_cast<const bf16x8*>(brow + 2 * 32 * 552 + c);
        b[1] = *reinterpret_cast<const bf16x8*>(brow + 3 * 32 * 552 + c);
    };
    f32x16 acc0, acc1, acc2;
    f32x4 sa[4];
    unsigned isum = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) sa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 zero8;
#pragma unroll
    for (int i = 0; i < 8; ++i) zero8[i] = (__bf16)(1.0f + lane);
#pragma unroll
    for (int i = 0; i < D; ++i) { if (WL) wload(i, ring[i]); else { ring[i][0] = zero8; ring[i][1] = zero8; } }
    constexpr int BD = 3;                                         // B operand ring (the kernel prefetches its LDS operands as well)
    bf16x8 bqr[BD][2];
#pragma unroll
    for (int i = 0; i < BD; ++i) { if (BL) bload(i, bqr[i]); else { bqr[i][0] = zero8; bqr[i][1] = zero8; } }
    bf16x8 bqr2[BD][2];
#pragma unroll
    for (int i = 0; i < BD; ++i) { if (BL && F64) bload2(i, bqr2[i]); else { bqr2[i][0] = zero8; bqr2[i][1] = zero8; } }
    static_assert(D % BD == 0 || BD % D == 0 || true, "");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    // MODE & 1024: the instruction cache is invalidated before every lap of the loop (18 k-steps): every k-step's code comes from L2
    // again -- the rows kernel's situation, whose 110 KB of straight-line code (every k-step of a tile its own piece of code, executed
    // once per tile) does not fit the 64 KB instruction cache two CUs share
    constexpr int U = D * BD;                                     // unroll: both rings on compile-time indices
    for (int s0 = 0; s0 < nsteps; s0 += U) {
        if constexpr (MODE & 1024) asm volatile("s_icache_inv" ::: "memory");
#pragma unroll
        for (int ii = 0; ii < U; ++ii) {
            const int i = ii % D;
            bf16x8 (&bq)[2] = bqr[ii % BD];
            if (MF) {
                if (!SMALL) {
                    if (F64) {
                        bf16x8 (&bq2)[2] = bqr2[ii % BD];
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq2[0], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][1], bq[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][1], bq2[0], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq[1], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq2[1], acc1, 0, 0, 0);
                    } else if (!IND) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq[0], acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][1], bq[0], acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq[1], acc0, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][1], bq[0], acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i][0], bq[1], acc2, 0, 0, 0);
                    }
                } else {
                    // the same FLOPs as three 32x32x16 (3 x 32 768) on 16x16x32 tiles (16 384 each): six instructions, four accumulators
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        sa[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[i][q & 1], bq[(q >> 1) & 1], sa[q & 3], 0, 0, 0);
                }
            } else {
                // every byte of every operand is consumed (a narrower use lets the compiler shrink the LDS reads to 2 and 4 bytes)
                const u32x4 q0 = __builtin_bit_cast(u32x4, ring[i][0]) ^ __builtin_bit_cast(u32x4, ring[i][1]) ^ __builtin_bit_cast(u32x4, bq[0]) ^ __builtin_bit_cast(u32x4, bq[1]);
                isum ^= q0[0] ^ q0[1] ^ q0[2] ^ q0[3];
            }
            if (BL) bload(s0 + ii + BD, bqr[ii % BD]);
            if (BL && F64) bload2(s0 + ii + BD, bqr2[ii % BD]);
            if (WL) wload(s0 + ii + D, ring[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long t1 = wall_clock64();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += sa[i][0] + sa[i][1] + sa[i][2] + sa[i][3];
#pragma unroll
    for (int i = 0; i < D; ++i) r += (float)ring[i][0][0] + (float)ring[i][1][0];
#pragma unroll
    for (int i = 0; i < BD; ++i) r += (float)bqr2[i][0][0] + (float)bqr2[i][1][0];
    if (r == 12345.678f || isum == 0x12345u) sink[0] = r + (float)isum;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = c1 - c0; }
}

template <int MODE, int D>
static void run(const char* name, const void* w, unsigned wbytes, unsigned long long* dout, float* sink) {
    const int nsteps = 2016, grid = 256;      // a multiple of every D * 3 used below
    const int threads = (MODE & 64) ? 512 : 256;
    const size_t lds = ((MODE & 2048) ? 2 : 1) * 2 * 32 * 552 * 2 + 64;
    CK(hipFuncSetAttribute((const void*)kstep_kernel<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kstep_kernel<MODE, D>), dim3(grid), dim3(threads), lds, 0, w, wbytes, nsteps, dout, sink);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(2 * grid);
    CK(hipMemcpy(h.data(), dout, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    std::vector<double> ns(grid), clk(grid);
    for (int i = 0; i < grid; ++i) { ns[i] = 10.0 * (double)h[2 * i] / nsteps; clk[i] = (double)h[2 * i + 1] / nsteps; }
    std::sort(ns.begin(), ns.end()); std::sort(clk.begin(), clk.end());
    const double per_cu_bytes = ((MODE & 1) ? ((MODE & 32) ? 1024.0 : 2048.0) : 0.0) * (threads / 64);
    const double frames = (MODE & 2048) ? 64.0 : 32.0;
    printf("%-72s D=%2d  %6.1f ns/k-step (max %6.1f)  %6.1f clk  %5.2f GHz  %6.1f GB/s per CU  %5.2f ns per frame and k-step\n", name, D, ns[grid / 2], ns[grid - 1], clk[grid / 2],
           clk[grid / 2] / ns[grid / 2], per_cu_bytes / ns[grid / 2], ns[grid / 2] / frames);
}

int main() {
    const unsigned wbytes = 2u * PLANE_STRIDE;                         // two planes of 2 x 216 k-steps x 4 tiles x 1 KB, L2-resident
    void* w; unsigned long long* dout; float* sink;
    CK(hipMalloc(&w, 16 * (size_t)wbytes)); CK(hipMemset(w, 0x3f, 16 * (size_t)wbytes));      // slack: whatever a descriptor variant makes of the range, it stays inside
    setvbuf(stdout, nullptr, _IOLBF, 0);
    CK(hipMalloc(&dout, sizeof(unsigned long long) * 1024)); CK(hipMalloc(&sink, 64));
    // warm the clocks
    for (int i = 0; i < 30; ++i) run<7, 6>("warm-up", w, wbytes, dout, sink);
    printf("---\n");
    run<7, 6>("32 frames: everything (the rows kernel's k-step)", w, wbytes, dout, sink);
    run<15, 6>("32 frames: everything, independent accumulators", w, wbytes, dout, sink);
    run<2048 + 4, 6>("64 frames: MFMAs only (6 per k-step, two accumulators)", w, wbytes, dout, sink);
    run<2048 + 2, 6>("64 frames: LDS operand reads only (4 KB per k-step and wave)", w, wbytes, dout, sink);
    run<2048 + 6, 6>("64 frames: MFMAs + LDS reads", w, wbytes, dout, sink);
    run<2048 + 5, 6>("64 frames: weight loads + MFMAs (no LDS)", w, wbytes, dout, sink);
    run<2048 + 7, 4>("64 frames: everything", w, wbytes, dout, sink);
    run<2048 + 7, 6>("64 frames: everything", w, wbytes, dout, sink);
    run<2048 + 7, 8>("64 frames: everything", w, wbytes, dout, sink);
    run<2048 + 7 + 1024, 6>("64 frames: everything, instruction cache invalidated every 18 k-steps", w, wbytes, dout, sink);
    return 0;
}
