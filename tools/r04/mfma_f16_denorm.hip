// Does v_mfma_f32_32x32x16_f16 on gfx950 keep fp16 SUBNORMAL inputs (or flush them to zero)?
// One wave: A = 2^-20 (subnormal in fp16) everywhere, B = 2^10 everywhere: C = 16 * 2^-10 = 2^-6 when kept, 0 when flushed.
// Also bf16 for comparison (A = bf16 subnormal 2^-130, B = 2^100 -> 16 * 2^-30).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1024.f; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    b8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)7.3468396926392969e-40f; bb[i] = (__bf16)1.2676506002282294e30f; }
    f16v d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, d, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8); hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("f16 subnormal x 2^10, K=16: got %g (kept: %g)\n", h[0], 0.015625);
    printf("bf16 subnormal 2^-130 x 2^100, K=16: got %g (kept: %g)\n", h[1], 16 * 9.313225746154785e-10);
    return 0;
}
