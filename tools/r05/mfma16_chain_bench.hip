// Round 5: what a dependent chain of v_mfma_f32_16x16x32_bf16 costs, alone and with the MCEM likelihood epilogue's VALU work between the
// MFMAs (csrc/mcem_resident16.hip: 12 dependent MFMAs per output tile, three transcendentals + four plain VALU per bin).  One wave per SIMD,
// one workgroup, s_memtime ticks (= shader clocks here) per MFMA slot, measured around 96 x REP of them.  Variants:
//   0 dependent chain   1 two independent chains, alternating   2 dependent chain, one v_exp_f32 behind every MFMA
//   3 dependent chain, v_exp_f32 + two v_fma_f32 behind every MFMA   4 two chains alternating, v_exp_f32 + two v_fma_f32 behind every MFMA
//   5 no MFMA, the VALU work of 3 alone   6 dependent chain, two v_fma_f32 behind every MFMA
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/r05/mfma16_chain_bench.hip -o tools/r05/bin/mfma16_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, int rep) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    float e = 0.5f + lane * 1e-3f, f = 1.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int i = 0; i < 96; ++i) {
            if constexpr (V != 5) {
                if constexpr (V == 1 || V == 4) {
                    if (i & 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
                } else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            }
            if constexpr (V == 2 || V == 3 || V == 4 || V == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(e));
            if constexpr (V == 3 || V == 4 || V == 5 || V == 6) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(e));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[1] + e + f;
}

// fp32 MFMAs (the exact-fp32 MCEM chains): W = 0: v_mfma_f32_16x16x4_f32 (8 passes), W = 1: v_mfma_f32_32x32x2_f32 (16 passes); NCH chains alternating
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int W, int NCH>
__global__ __launch_bounds__(256, 1) void kf(unsigned long long* out, float* sink, int rep) {
    const int lane = threadIdx.x & 63;
    float a = 0.001f * lane, b = 0.002f * lane;
    f32x4 c4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x16 c16[2];
    for (int i = 0; i < 16; ++i) { c16[0][i] = 0.f; c16[1][i] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int i = 0; i < 96; ++i) {
            if constexpr (W == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4[i % NCH]) : "v"(a), "v"(b));
            else if constexpr (W == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c4[i % NCH]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16[i % NCH]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = c4[0][0] + c4[1][1] + c4[2][2] + c4[3][3] + c16[0][0] + c16[1][1];
}

template <int W, int NCH> void runf(const char* what) {
    unsigned long long* out; float* sink;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4096));
    const int rep = 200;
    hipLaunchKernelGGL((kf<W, NCH>), dim3(1), dim3(256), 0, 0, out, sink, rep);
    hipLaunchKernelGGL((kf<W, NCH>), dim3(1), dim3(256), 0, 0, out, sink, rep);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    printf("f %-70s %.1f s_memtime ticks per 96 slots = %.2f per slot\n", what, (double)h[0] / rep, (double)h[0] / rep / 96.0);
    CK(hipFree(out)); CK(hipFree(sink));
}

// the epilogue's instruction mix with no dependences between consecutive instructions: per slot NT transcendentals (v_exp_f32 on eight rotating
// registers) and NF v_fma_f32 (eight rotating registers), with MF = 0 no MFMA, 1 a dependent v_mfma_f32_16x16x32_bf16, 2 a dependent v_mfma_f32_16x16x4_f32
template <int MF, int NT, int NF>
__global__ __launch_bounds__(1024, 1) void km(unsigned long long* out, float* sink, int rep) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    float af = 0.001f * lane, bf = 0.002f * lane;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    float e[8], f[8];
    for (int i = 0; i < 8; ++i) { e[i] = 0.1f * i + lane * 1e-3f; f[i] = 1.f + i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int i = 0; i < 96; ++i) {
            if constexpr (MF == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            if constexpr (MF == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(af), "v"(bf));
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(i * NT + t) & 7]));
#pragma unroll
            for (int t = 0; t < NF; ++t) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(i * NF + t) & 7]) : "v"(af));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
    float sm = c0[0];
    for (int i = 0; i < 8; ++i) sm += e[i] + f[i];
    sink[threadIdx.x] = sm;
}
template <int MF, int NT, int NF> void runm(const char* what, int threads = 256) {
    unsigned long long* out; float* sink;
    CK(hipMalloc(&out, 256)); CK(hipMalloc(&sink, 16384));
    const int rep = 200;
    hipLaunchKernelGGL((km<MF, NT, NF>), dim3(1), dim3(threads), 0, 0, out, sink, rep);
    hipLaunchKernelGGL((km<MF, NT, NF>), dim3(1), dim3(threads), 0, 0, out, sink, rep);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    printf("m %-70s %.1f s_memtime ticks per 96 slots = %.2f per slot", what, (double)h[0] / rep, (double)h[0] / rep / 96.0);
    if (threads > 256) {
        unsigned long long h2[16];
        CK(hipMemcpy(h2, out, 8 * (threads / 64), hipMemcpyDeviceToHost));
        printf("  (every wave:");
        for (int w = 0; w < threads / 64; ++w) printf(" %.1f", (double)h2[w] / rep / 96.0);
        printf(")");
    }
    printf("\n");
    CK(hipFree(out)); CK(hipFree(sink));
}

template <int V> void run(const char* what) {
    unsigned long long* out; float* sink;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4096));
    const int rep = 200;
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, out, sink, rep);
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, out, sink, rep);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    // (s_memtime ticks at the shader clock here: a back-to-back chain of these 4-pass MFMAs reads 16.15 per MFMA)
    printf("%d %-70s %.1f s_memtime ticks per 96 slots = %.2f per slot\n", V, what, (double)h[0] / rep, (double)h[0] / rep / 96.0);
    CK(hipFree(out)); CK(hipFree(sink));
}

int main() {
    run<0>("dependent chain");
    run<1>("two independent chains, alternating");
    run<2>("dependent chain + v_exp_f32");
    run<3>("dependent chain + v_exp_f32 + 2 v_fma_f32");
    run<4>("two chains alternating + v_exp_f32 + 2 v_fma_f32");
    run<5>("v_exp_f32 + 2 v_fma_f32, no MFMA");
    run<6>("dependent chain + 2 v_fma_f32");
    runm<0, 1, 0>("independent VALU: 1 v_exp_f32 per slot, no MFMA");
    runm<0, 0, 4>("independent VALU: 4 v_fma_f32 per slot, no MFMA");
    runm<0, 1, 2>("independent VALU: 1 v_exp_f32 + 2 v_fma_f32 per slot, no MFMA");
    runm<1, 1, 2>("bf16 16x16x32 chain + 1 v_exp_f32 + 2 v_fma_f32 (independent) per slot");
    runm<1, 0, 4>("bf16 16x16x32 chain + 4 v_fma_f32 (independent) per slot");
    runm<1, 1, 0>("bf16 16x16x32 chain + 1 v_exp_f32 per slot");
    runm<2, 1, 2>("fp32 16x16x4 chain + 1 v_exp_f32 + 2 v_fma_f32 (independent) per slot");
    runm<2, 2, 4>("fp32 16x16x4 chain + 2 v_exp_f32 + 4 v_fma_f32 (independent) per slot");
    runm<2, 0, 6>("fp32 16x16x4 chain + 6 v_fma_f32 (independent) per slot");
    // two waves per SIMD (512 threads): does one wave's VALU work run beside the other's MFMAs?  (time per slot of EACH wave; both do the same work)
    runm<1, 1, 2>("TWO waves per SIMD: bf16 16x16x32 chain + 1 v_exp_f32 + 2 v_fma_f32 per slot", 512);
    runm<1, 0, 0>("TWO waves per SIMD: bf16 16x16x32 chain alone", 512);
    runm<0, 1, 2>("TWO waves per SIMD: 1 v_exp_f32 + 2 v_fma_f32 per slot, no MFMA", 512);
    runm<2, 1, 2>("TWO waves per SIMD: fp32 16x16x4 chain + 1 v_exp_f32 + 2 v_fma_f32 per slot", 512);
    runm<1, 1, 2>("FOUR waves per SIMD: bf16 16x16x32 chain + 1 v_exp_f32 + 2 v_fma_f32 per slot", 1024);
    runm<1, 0, 0>("FOUR waves per SIMD: bf16 16x16x32 chain alone", 1024);
    runm<0, 1, 2>("FOUR waves per SIMD: 1 v_exp_f32 + 2 v_fma_f32 per slot, no MFMA", 1024);
    runf<0, 1>("v_mfma_f32_16x16x4_f32 (8 passes): dependent chain");
    runf<0, 2>("v_mfma_f32_16x16x4_f32: two chains alternating");
    runf<0, 4>("v_mfma_f32_16x16x4_f32: four chains alternating");
    runf<2, 1>("v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4, K = 1): dependent chain");
    runf<2, 2>("v_mfma_f32_4x4x1_16b_f32: two chains alternating");
    runf<2, 4>("v_mfma_f32_4x4x1_16b_f32: four chains alternating");
    runf<1, 1>("v_mfma_f32_32x32x2_f32 (16 passes): dependent chain");
    runf<1, 2>("v_mfma_f32_32x32x2_f32: two chains alternating");
    return 0;
}
