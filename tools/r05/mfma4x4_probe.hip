// Round 5: operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4, K = 1) and what update_dpp row_ror gives:
// A = 100 + lane, B = 1 at ONE lane (others 0), C = 0 -> prints which D registers of which lanes become non-zero and their values.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int hot) {
    const int l = threadIdx.x;
    float a = 100.f + l, b = l == hot ? 1.f : 0.f;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 3" : "+v"(c) : "v"(a), "v"(b));
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
    const unsigned r4 = __builtin_amdgcn_update_dpp(0u, (unsigned)l, 0x124, 0xf, 0xf, false);   // row_ror:4
    const unsigned r12 = __builtin_amdgcn_update_dpp(0u, (unsigned)l, 0x12c, 0xf, 0xf, false);  // row_ror:12
    const unsigned r8 = __builtin_amdgcn_update_dpp(0u, (unsigned)l, 0x128, 0xf, 0xf, false);   // row_ror:8
    out[256 + l] = (float)r4; out[320 + l] = (float)r12; out[384 + l] = (float)r8;
}
int main() {
    float* d; float h[448];
    (void)hipMalloc(&d, sizeof(h));
    for (int hot : {0, 1, 5, 18}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, hot);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("B = 1 at lane %d: non-zero D:", hot);
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (h[l * 4 + i] != 0.f) printf("  lane %d reg %d = %.0f", l, i, h[l * 4 + i]);
        printf("\n");
    }
    printf("row_ror:4  of lane id:"); for (int l = 0; l < 20; ++l) printf(" %.0f", h[256 + l]); printf("\n");
    printf("row_ror:12 of lane id:"); for (int l = 0; l < 20; ++l) printf(" %.0f", h[320 + l]); printf("\n");
    printf("row_ror:8  of lane id:"); for (int l = 0; l < 20; ++l) printf(" %.0f", h[384 + l]); printf("\n");
    return 0;
}
