// Round 5: what __builtin_amdgcn_permlane16_swap / permlane32_swap return (vdst', src') for vdst = lane id, src = 100 + lane id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap vdst'", "permlane16_swap src'", "permlane32_swap vdst'", "permlane32_swap src'"};
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * r + i]); printf("\n"); }
    return 0;
}
