// Frame-store helpers for the GPU-resident training set (replacement of the reference's per-column HDF5 reads,
// packages/data_handling.py:45-60): row gather (epoch shuffle) and the (F, N) -> [N][F] relayout of the on-disk
// format (scripts/create_train_set.py:116: X_<split> is (513, N), one frame per column).  HBM-bound copies.
#include "common.hpp"

namespace dvae {

// dst[i][0..cols) = src[idx[i]][0..cols): one wave per row, lanes stride the columns (coalesced 256 B per instruction;
// rows of 513 floats are only 4-byte aligned, so there is no wider vector access to be had)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ idx,
                                                          int64_t n, int cols, float* __restrict__ dst, int64_t ldd, int64_t nsrc,
                                                          int* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t r = idx[i];
        if (r < 0 || r >= nsrc) { if (lane == 0 && bad) atomicAdd(bad, 1); continue; }    // never read out of bounds
        const float* s = src + r * ld;
        float* d = dst + i * ldd;
        for (int c = lane; c < cols; c += 64) d[c] = s[c];
    }
}

// out[n][f] = in[f][n] through a 32 x 33 LDS tile
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t rows, int64_t cols, int64_t ldi,
                                                        float* __restrict__ out, int64_t ldo) {
    __shared__ float t[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) t[j][tx] = in[(r0 + j) * ldi + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) out[(c0 + j) * ldo + r0 + tx] = t[tx][j];
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_gather_rows(const float* src, int64_t ld, int64_t nsrc, const int64_t* idx, int64_t n, int cols, float* dst,
                                int64_t ldd, int* bad_count, void* stream) {
    DVAE_CHECK_ARG(src && idx && dst, "gather_rows: null argument");
    DVAE_CHECK_ARG(cols > 0 && ld >= cols && ldd >= cols && n >= 0 && nsrc > 0, "gather_rows: bad shape (cols %d ld %lld ldd %lld)", cols, (long long)ld, (long long)ldd);
    if (n == 0) return 0;
    const int64_t blocks = cdiv(n, 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, src, ld, idx, n,
                       cols, dst, ldd, nsrc, bad_count);
    DVAE_LAUNCH_OK("gather_rows_kernel");
    return 0;
}

extern "C" int dvae_transpose(const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo, void* stream) {
    DVAE_CHECK_ARG(in && out, "transpose: null argument");
    DVAE_CHECK_ARG(rows > 0 && cols > 0 && ldi >= cols && ldo >= rows, "transpose: bad shape");
    DVAE_CHECK_ARG(cdiv(rows, 32) <= 65535, "transpose: more than 2M rows (put the long axis on the columns)");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, in, rows, cols,
                       ldi, out, ldo);
    DVAE_LAUNCH_OK("transpose_kernel");
    return 0;
}
