// Micro-benchmark: cycles per k-step of the rows-kernel GEMM inner loop under different weight address patterns.
// One workgroup of NW waves on an otherwise idle chip; s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int D = 16, NSTEPS = 528;   // 33 laps

template <int MODE>   // 0 row-major [128][ld], 1 fragment-major contiguous, 2 no global loads, 3 loads only (no mfma), 4 row-major but LDS reads removed
__global__ __launch_bounds__(256) void k(const __bf16* W, int ld, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) __bf16 U[32 * 552];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 32 * 552; i += blockDim.x) U[i] = (__bf16)(0.001f * (i % 7));
    __syncthreads();
    const __bf16* wrow; int wstr;
    if (MODE == 1) { wrow = W + wave * 512 + lane * 8; wstr = 4 * 512; }
    else { wrow = W + (long)(32 * wave + l31) * ld + h * 8; wstr = 16; }
    const __bf16* brow = U + l31 * 552 + h * 8;
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 a[D], bq[4];
    for (int i = 0; i < D; ++i) a[i] = *(const bf16x8*)(wrow + i * wstr);
    for (int i = 0; i < 4; ++i) bq[i] = *(const bf16x8*)(brow + i * 16);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int c = 0; c < NSTEPS / D - 1; ++c) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if (MODE != 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[i % 4], acc, 0, 0, 0);
            else acc[0] += (float)a[i][0] + (float)bq[i % 4][0];
            if (MODE != 4) bq[i % 4] = *(const bf16x8*)(brow + ((c * D + i + 4) % 32) * 16);
            if (MODE != 2) a[i] = *(const bf16x8*)(wrow + ((c + 1) * D + i) * wstr);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < D; ++i) s += (float)a[i][1];
    sink[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE> void run(const char* name, const __bf16* W, int ld, int blocks, int threads) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 4 * 8); hipMalloc(&sink, blocks * 256 * 4);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, W, ld, out, sink);
    hipDeviceSynchronize();
    unsigned long long h[4096]; hipMemcpy(h, out, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double m = 0; int n = blocks * (threads / 64); for (int i = 0; i < n; ++i) m += h[i / (threads / 64) * 4 + i % (threads / 64)];
    printf("%-44s blocks %4d waves/blk %d : %.1f clk/step\n", name, blocks, threads / 64, m / n / (NSTEPS - D));
    hipFree(out); hipFree(sink);
}

int main() {
    __bf16* W; size_t bytes = (size_t)16 << 20;   // mode 1 walks 528 steps x 4 KB = 2.2 MB; mode 0 stays inside 0.3 MB
    hipMalloc(&W, bytes); hipMemset(W, 0, bytes);
    for (int blocks : {1, 256}) for (int threads : {64, 256}) {
        run<0>("row-major 32 rows x 32 B per load", W, 1056, blocks, threads);
        run<1>("fragment-major 1 KB contiguous per load", W, 1056, blocks, threads);
        run<2>("no global loads (LDS + MFMA)", W, 1056, blocks, threads);
        run<3>("loads only (no MFMA)", W, 1056, blocks, threads);
        run<4>("row-major, no LDS reads", W, 1056, blocks, threads);
    }
    return 0;
}
