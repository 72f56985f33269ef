// Reparametrisation, ELBO (Itakura-Saito + KL), BCE family and Adam: the
// HBM-bound elementwise / reduction kernels of the layer-level path.
// One wave64 per frame for the 513-bin Itakura-Saito row sum (wavefront
// shuffle reduction), per-block partials in double, a one-block final pass:
// deterministic (no atomics), no host sync.
#include <float.h>
#include "common.hpp"

namespace dvae {

constexpr int kMaxPartials = 1024;

__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                           const float* __restrict__ eps, float* __restrict__ z, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = expf(lv[i] * 0.5f);          // models.py:17  std = log_var.mul(0.5).exp_()
        z[i] = fmaf(s, eps[i], mu[i]);               // models.py:20  mu.addcmul(std, epsilon)
    }
}

__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ lv,
                                                           const float* __restrict__ eps, float* __restrict__ dmu,
                                                           float* __restrict__ dlv, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = dz[i];
        dmu[i] = g;
        dlv[i] = g * eps[i] * (0.5f * expf(lv[i] * 0.5f));
    }
}

// partials: double[2 * gridDim.x] = {sum_b recon_b, sum_b kl_b} per block
__global__ __launch_bounds__(256) void elbo_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ r, int ldr,
                                                         const float* __restrict__ mu, const float* __restrict__ lv, float eps,
                                                         int64_t B, int F, int Z, float* __restrict__ kl_b,
                                                         double* __restrict__ partials) {
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nw = (int64_t)gridDim.x * 4;
    double s_rec = 0.0, s_kl = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += nw) {
        const float* xr = x + b * ldx;
        const float* rr = r + b * ldr;
        float acc = 0.f;
        for (int f = lane; f < F; f += 64) {
            const float xv = xr[f], rv = rr[f];
            acc += xv / rv - logf(xv + eps) + logf(rv) - 1.f;     // utils.py:74
        }
        float k = 0.f;
        for (int j = lane; j < Z; j += 64) {
            const float m = mu[b * Z + j], l = lv[b * Z + j];
            k += l - m * m - expf(l);                             // utils.py:75
        }
        acc = wave_sum(acc);
        k = -0.5f * wave_sum(k);
        if (lane == 0) {
            s_rec += (double)acc;
            s_kl += (double)k;
            if (kl_b) kl_b[b] = k;                                // models.py:165-167
        }
    }
    if (lane == 0) { red[wave][0] = s_rec; red[wave][1] = s_kl; }
    __syncthreads();
    if (threadIdx.x < 2) {
        partials[2 * blockIdx.x + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void elbo_final_kernel(const double* __restrict__ partials, int nblocks, int64_t B, float* __restrict__ out3) {
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0, k = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += partials[2 * i]; k += partials[2 * i + 1]; }
    a = wave_sum(a); k = wave_sum(k);
    if (lane == 0) { red[wave][0] = a; red[wave][1] = k; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float recon = (float)((red[0][0] + red[1][0] + red[2][0] + red[3][0]) / (double)B);
        const float kl = (float)((red[0][1] + red[1][1] + red[2][1] + red[3][1]) / (double)B);
        out3[0] = recon + kl; out3[1] = recon; out3[2] = kl;
    }
}

// upstream scalars: either g2 = {d/d recon, d/d KL} (ga = gb = null), or the three gradients of the outputs (loss, recon, KL)
// of elbo(), each a device scalar or null (= 0): d/d recon = g_loss + g_recon, d/d KL = g_loss + g_kl
__global__ __launch_bounds__(256) void elbo_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ r, int ldr,
                                                        const float* __restrict__ mu, const float* __restrict__ lv,
                                                        const float* __restrict__ g2, const float* __restrict__ ga, const float* __restrict__ gb,
                                                        int three, int64_t B, int F, int Z,
                                                        float* __restrict__ dr, int lddr, float* __restrict__ dmu, float* __restrict__ dlv) {
    const float invB = 1.f / (float)B;
    float s_r, s_k;
    if (three) {
        const float gl = g2 ? g2[0] : 0.f;
        s_r = gl + (ga ? ga[0] : 0.f); s_k = gl + (gb ? gb[0] : 0.f);
    } else { s_r = g2[0]; s_k = g2[1]; }
    const float gr = s_r * invB, gk = s_k * invB;
    const int64_t total = B * (int64_t)F, nz = B * (int64_t)Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const float xv = x[b * ldx + f], rv = r[b * ldr + f];
        if (dr) dr[b * lddr + f] = gr / rv - gr * ((xv / rv) / rv);    // d/dr [x/r + log r]
        if (i < nz) {
            if (dmu) dmu[i] = gk * mu[i];
            if (dlv) dlv[i] = -0.5f * gk * (1.f - expf(lv[i]));
        }
    }
}

// flat sum over B*Y elements; partials double[gridDim.x]
__global__ __launch_bounds__(256) void bce_sum_kernel(const float* __restrict__ r, const float* __restrict__ t, float eps,
                                                       int64_t n, int variant, double* __restrict__ partials) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float rv = r[i];
        const float tv = variant == 0 ? t[i] : (variant == 1 ? 0.5f : rv);
        s += (double)(tv * logf(rv + eps) + (1.f - tv) * logf(1.f - rv + eps));   // utils.py:55-63
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void bce_final_kernel(const double* __restrict__ partials, int nblocks, int64_t B, float* __restrict__ out1) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) a += partials[i];
    a = wave_sum(a);
    if (lane == 0) red[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) out1[0] = (float)(-(red[0] + red[1] + red[2] + red[3]) / (double)B);
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ r, const float* __restrict__ t, float eps,
                                                       const float* __restrict__ g, int64_t B, int64_t n, int variant,
                                                       float* __restrict__ dr, float* __restrict__ dt) {
    const float s = -g[0] / (float)B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float rv = r[i];
        const float la = logf(rv + eps), lb = logf(1.f - rv + eps);
        float d;
        if (variant == 0) {
            const float tv = t[i];
            d = tv / (rv + eps) - (1.f - tv) / (1.f - rv + eps);
            if (dt) dt[i] = s * (la - lb);
        } else if (variant == 1) {
            d = 0.5f / (rv + eps) - 0.5f / (1.f - rv + eps);
        } else {  // d/dr [r log(r+e) + (1-r) log(1-r+e)]
            d = la + rv / (rv + eps) - lb - (1.f - rv) / (1.f - rv + eps);
        }
        dr[i] = s * d;
    }
}

// torch.optim.Adam single-tensor update (no weight decay / amsgrad / maximize)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float one_minus_b1, float b2, float one_minus_b2,
                                                    float step_size, float bc2_sqrt, float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + one_minus_b1 * (gi - m[i]);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);       // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);                      // param.addcdiv_(exp_avg, denom, value=-step_size)
        m[i] = mi;
        v[i] = vi;
    }
}


// ---- loss zoo of packages/models/utils.py beyond elbo / BCE (reference :65-118): per-frame Itakura-Saito + KL rows (L_loss,
// ikatura_saito_divergence), the two-class BCE and the squared-error mask / signal / magnitude-spectrum-approximation losses.
// Same structure as above: one wave per frame or a flat grid-stride sum, double partials, one-block final pass, no atomics.
__global__ __launch_bounds__(256) void isrows_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ r, int ldr,
                                                          const float* __restrict__ mu, const float* __restrict__ lv, float eps,
                                                          int64_t B, int F, int Z, float* __restrict__ rec_b, float* __restrict__ kl_b) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += nw) {
        const float* xr = x + b * ldx;
        const float* rr = r + b * ldr;
        float acc = 0.f;
        for (int f = lane; f < F; f += 64) {
            const float xv = xr[f], rv = rr[f];
            acc += xv / rv - logf(xv + eps) + logf(rv) - 1.f;     // utils.py:71, 79
        }
        acc = wave_sum(acc);
        float k = 0.f;
        if (kl_b) {
            for (int j = lane; j < Z; j += 64) {
                const float m = mu[b * Z + j], l = lv[b * Z + j];
                k += l - m * m - expf(l);                         // utils.py:80
            }
            k = -0.5f * wave_sum(k);
        }
        if (lane == 0) { rec_b[b] = acc; if (kl_b) kl_b[b] = k; }
    }
}

__global__ __launch_bounds__(256) void isrows_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ r, int ldr,
                                                          const float* __restrict__ mu, const float* __restrict__ lv,
                                                          const float* __restrict__ g_rec, const float* __restrict__ g_kl,
                                                          int64_t B, int F, int Z, float* __restrict__ dr, int lddr,
                                                          float* __restrict__ dmu, float* __restrict__ dlv) {
    const int64_t total = B * (int64_t)F, nz = B * (int64_t)Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        if (dr) {
            const float gr = g_rec ? g_rec[b] : 0.f;
            const float xv = x[b * ldx + f], rv = r[b * ldr + f];
            dr[b * lddr + f] = gr / rv - gr * ((xv / rv) / rv);    // d/dr [x/r + log r]
        }
        if (i < nz && (dmu || dlv)) {
            const float gk = g_kl ? g_kl[i / Z] : 0.f;
            if (dmu) dmu[i] = gk * mu[i];
            if (dlv) dlv[i] = -0.5f * gk * (1.f - expf(lv[i]));
        }
    }
}

// two-class BCE (utils.py:65-66): sum of t log(r1 + eps) + (1 - t) log(r2 + eps)
__global__ __launch_bounds__(256) void bce2_sum_kernel(const float* __restrict__ r1, const float* __restrict__ r2, const float* __restrict__ t,
                                                        float eps, int64_t n, double* __restrict__ partials) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (double)(t[i] * logf(r1[i] + eps) + (1.f - t[i]) * logf(r2[i] + eps));
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void bce2_bwd_kernel(const float* __restrict__ r1, const float* __restrict__ r2, const float* __restrict__ t,
                                                        float eps, const float* __restrict__ g, int64_t B, int64_t n,
                                                        float* __restrict__ dr1, float* __restrict__ dr2, float* __restrict__ dt) {
    const float s = -g[0] / (float)B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float tv = t[i];
        if (dr1) dr1[i] = s * tv / (r1[i] + eps);
        if (dr2) dr2[i] = s * (1.f - tv) / (r2[i] + eps);
        if (dt) dt[i] = s * (logf(r1[i] + eps) - logf(r2[i] + eps));
    }
}

// squared-error family (utils.py:107-118), mean over frames of the row sums of |d|^2:
//   mode 0 (mean_square_error_signal): d = (y - yhat) * x       mode 1 (mean_square_error_mask): d = y - yhat
//   mode 2 (magnitude_spectrum_approxiamation_loss): d = s - yhat * x with complex64 s, x (float2) and a real mask yhat
__global__ __launch_bounds__(256) void sqerr_sum_kernel(int mode, const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ yhat, int64_t n, double* __restrict__ partials) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (mode == 2) {
            const float2 xc = reinterpret_cast<const float2*>(x)[i], sc = reinterpret_cast<const float2*>(y)[i];
            const float m = yhat[i];
            const float dr_ = sc.x - m * xc.x, di = sc.y - m * xc.y;
            s += (double)(dr_ * dr_ + di * di);
        } else {
            float d = y[i] - yhat[i];
            if (mode == 0) d *= x[i];
            s += (double)(d * d);
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mean_final_kernel(const double* __restrict__ partials, int nblocks, int64_t B, float sign, float* __restrict__ out1) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) a += partials[i];
    a = wave_sum(a);
    if (lane == 0) red[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) out1[0] = (float)((double)sign * (red[0] + red[1] + red[2] + red[3]) / (double)B);
}

__global__ __launch_bounds__(256) void sqerr_bwd_kernel(int mode, const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ yhat, const float* __restrict__ g, int64_t B, int64_t n,
                                                         float* __restrict__ dyhat, float* __restrict__ dy, float* __restrict__ dx) {
    const float s = 2.f * g[0] / (float)B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (mode == 2) {
            const float2 xc = reinterpret_cast<const float2*>(x)[i], sc = reinterpret_cast<const float2*>(y)[i];
            const float m = yhat[i];
            const float dr_ = sc.x - m * xc.x, di = sc.y - m * xc.y;
            if (dyhat) dyhat[i] = -s * (dr_ * xc.x + di * xc.y);        // d |s - m x|^2 / d m = -2 Re(d conj(x))
        } else {
            const float e = y[i] - yhat[i];
            const float xv = mode == 0 ? x[i] : 1.f;
            const float ge = s * e * xv * xv;
            if (dyhat) dyhat[i] = -ge;
            if (dy) dy[i] = ge;
            if (dx && mode == 0) dx[i] = s * e * e * xv;
        }
    }
}

static inline int ew_blocks(int64_t n) {
    int64_t b = cdiv(n, 256);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace dvae

using namespace dvae;

extern "C" int dvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream) {
    DVAE_CHECK_ARG(mu && logvar && eps && z && n >= 0, "reparam_fwd: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, z, n);
    DVAE_LAUNCH_OK("reparam_fwd");
    return 0;
}

extern "C" int dvae_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu, float* dlogvar, int64_t n, void* stream) {
    DVAE_CHECK_ARG(dz && logvar && eps && dmu && dlogvar && n >= 0, "reparam_bwd: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dz, logvar, eps, dmu, dlogvar, n);
    DVAE_LAUNCH_OK("reparam_bwd");
    return 0;
}

extern "C" size_t dvae_elbo_workspace_bytes(int64_t B) {
    (void)B;
    return (size_t)kMaxPartials * 2 * sizeof(double);
}

extern "C" int dvae_elbo_fwd(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar,
                             float eps, int64_t B, int F, int Z, float* out3, float* kl_b, void* ws, void* stream) {
    DVAE_CHECK_ARG(x && r && mu && logvar && out3 && ws && B > 0 && F > 0 && Z > 0 && ldx >= F && ldr >= F, "elbo_fwd: bad argument");
    int nb = (int)(cdiv(B, 4) < kMaxPartials ? cdiv(B, 4) : kMaxPartials);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(elbo_rows_kernel, dim3(nb), dim3(256), 0, s, x, ldx, r, ldr, mu, logvar, eps, B, F, Z, kl_b, (double*)ws);
    DVAE_LAUNCH_OK("elbo_rows");
    hipLaunchKernelGGL(elbo_final_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nb, B, out3);
    DVAE_LAUNCH_OK("elbo_final");
    return 0;
}

extern "C" int dvae_elbo_bwd(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar,
                             const float* g2, int64_t B, int F, int Z, float* dr, int lddr, float* dmu, float* dlogvar, void* stream) {
    DVAE_CHECK_ARG(x && r && mu && logvar && g2 && B > 0 && F > 0 && Z > 0 && ldx >= F && ldr >= F && (!dr || lddr >= F), "elbo_bwd: bad argument");
    DVAE_CHECK_ARG(Z <= F, "elbo_bwd: latent dim larger than feature dim");
    hipLaunchKernelGGL(elbo_bwd_kernel, dim3(ew_blocks(B * (int64_t)F)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, r, ldr, mu, logvar, g2, (const float*)nullptr, (const float*)nullptr, 0, B, F, Z, dr, lddr, dmu, dlogvar);
    DVAE_LAUNCH_OK("elbo_bwd");
    return 0;
}

extern "C" int dvae_elbo_bwd3(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar,
                              const float* g_loss, const float* g_recon, const float* g_kl, int64_t B, int F, int Z,
                              float* dr, int lddr, float* dmu, float* dlogvar, void* stream) {
    DVAE_CHECK_ARG(x && r && mu && logvar && B > 0 && F > 0 && Z > 0 && ldx >= F && ldr >= F && (!dr || lddr >= F), "elbo_bwd3: bad argument");
    DVAE_CHECK_ARG(Z <= F, "elbo_bwd3: latent dim larger than feature dim");
    hipLaunchKernelGGL(elbo_bwd_kernel, dim3(ew_blocks(B * (int64_t)F)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, r, ldr, mu, logvar, g_loss, g_recon, g_kl, 1, B, F, Z, dr, lddr, dmu, dlogvar);
    DVAE_LAUNCH_OK("elbo_bwd3");
    return 0;
}

extern "C" int dvae_bce_fwd(const float* r, const float* t, float eps, int64_t B, int Y, int variant, float* out1, void* ws, void* stream) {
    DVAE_CHECK_ARG(r && out1 && ws && B > 0 && Y > 0 && variant >= 0 && variant <= 2 && (variant != 0 || t), "bce_fwd: bad argument");
    const int64_t n = B * (int64_t)Y;
    int nb = (int)(cdiv(n, 256) < kMaxPartials ? cdiv(n, 256) : kMaxPartials);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_sum_kernel, dim3(nb), dim3(256), 0, s, r, t, eps, n, variant, (double*)ws);
    DVAE_LAUNCH_OK("bce_sum");
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nb, B, out1);
    DVAE_LAUNCH_OK("bce_final");
    return 0;
}

extern "C" int dvae_bce_bwd(const float* r, const float* t, float eps, const float* g, int64_t B, int Y, int variant,
                            float* dr, float* dt, void* stream) {
    DVAE_CHECK_ARG(r && g && dr && B > 0 && Y > 0 && variant >= 0 && variant <= 2 && (variant != 0 || t), "bce_bwd: bad argument");
    const int64_t n = B * (int64_t)Y;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, r, t, eps, g, B, n, variant, dr, dt);
    DVAE_LAUNCH_OK("bce_bwd");
    return 0;
}


/* ---- remaining losses of packages/models/utils.py (reference :65-118) ---- */
extern "C" int dvae_isrows_fwd(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar, float eps,
                               int64_t B, int F, int Z, float* recon_rows, float* kl_rows, void* stream) {
    DVAE_CHECK_ARG(x && r && recon_rows && B > 0 && F > 0 && ldx >= F && ldr >= F && (!kl_rows || (mu && logvar && Z > 0)), "isrows_fwd: bad argument");
    const int nb = (int)(cdiv(B, 4) < 2048 ? cdiv(B, 4) : 2048);
    hipLaunchKernelGGL(isrows_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, ldx, r, ldr, mu, logvar, eps, B, F, Z, recon_rows, kl_rows);
    DVAE_LAUNCH_OK("isrows_fwd");
    return 0;
}

extern "C" int dvae_isrows_bwd(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar,
                               const float* g_recon_rows, const float* g_kl_rows, int64_t B, int F, int Z,
                               float* dr, int lddr, float* dmu, float* dlogvar, void* stream) {
    DVAE_CHECK_ARG(x && r && B > 0 && F > 0 && ldx >= F && ldr >= F && (!dr || lddr >= F) && ((!dmu && !dlogvar) || (mu && logvar && Z > 0 && Z <= F)),
                   "isrows_bwd: bad argument");
    hipLaunchKernelGGL(isrows_bwd_kernel, dim3(ew_blocks(B * (int64_t)F)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, r, ldr, mu, logvar, g_recon_rows, g_kl_rows, B, F, Z, dr, lddr, dmu, dlogvar);
    DVAE_LAUNCH_OK("isrows_bwd");
    return 0;
}

extern "C" int dvae_bce2_fwd(const float* r1, const float* r2, const float* t, float eps, int64_t B, int Y, float* out1, void* ws, void* stream) {
    DVAE_CHECK_ARG(r1 && r2 && t && out1 && ws && B > 0 && Y > 0, "bce2_fwd: bad argument");
    const int64_t n = B * (int64_t)Y;
    const int nb = (int)(cdiv(n, 256) < kMaxPartials ? cdiv(n, 256) : kMaxPartials);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bce2_sum_kernel, dim3(nb), dim3(256), 0, s, r1, r2, t, eps, n, (double*)ws);
    DVAE_LAUNCH_OK("bce2_sum");
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nb, B, -1.f, out1);
    DVAE_LAUNCH_OK("bce2_final");
    return 0;
}

extern "C" int dvae_bce2_bwd(const float* r1, const float* r2, const float* t, float eps, const float* g, int64_t B, int Y,
                             float* dr1, float* dr2, float* dt, void* stream) {
    DVAE_CHECK_ARG(r1 && r2 && t && g && B > 0 && Y > 0, "bce2_bwd: bad argument");
    const int64_t n = B * (int64_t)Y;
    hipLaunchKernelGGL(bce2_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, r1, r2, t, eps, g, B, n, dr1, dr2, dt);
    DVAE_LAUNCH_OK("bce2_bwd");
    return 0;
}

extern "C" int dvae_sqerr_fwd(int mode, const void* x, const void* y, const float* yhat, int64_t B, int F, float* out1, void* ws, void* stream) {
    DVAE_CHECK_ARG(mode >= 0 && mode <= 2 && y && yhat && out1 && ws && B > 0 && F > 0 && (mode == 1 || x), "sqerr_fwd: bad argument");
    const int64_t n = B * (int64_t)F;
    const int nb = (int)(cdiv(n, 256) < kMaxPartials ? cdiv(n, 256) : kMaxPartials);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sqerr_sum_kernel, dim3(nb), dim3(256), 0, s, mode, (const float*)x, (const float*)y, yhat, n, (double*)ws);
    DVAE_LAUNCH_OK("sqerr_sum");
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nb, B, 1.f, out1);
    DVAE_LAUNCH_OK("sqerr_final");
    return 0;
}

extern "C" int dvae_sqerr_bwd(int mode, const void* x, const void* y, const float* yhat, const float* g, int64_t B, int F,
                              float* dyhat, float* dy, float* dx, void* stream) {
    DVAE_CHECK_ARG(mode >= 0 && mode <= 2 && y && yhat && g && B > 0 && F > 0 && (mode == 1 || x) && (mode != 2 || (!dy && !dx)), "sqerr_bwd: bad argument");
    const int64_t n = B * (int64_t)F;
    hipLaunchKernelGGL(sqerr_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, mode, (const float*)x, (const float*)y, yhat, g, B, n, dyhat, dy, dx);
    DVAE_LAUNCH_OK("sqerr_bwd");
    return 0;
}

extern "C" int dvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                              double eps, int step, double grad_scale, void* stream) {
    DVAE_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "adam_step: bad argument");
    if (n == 0) return 0;
    // bias corrections in double on the host, as torch's python scalars (torch/optim/adam.py)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps, (float)grad_scale);
    DVAE_LAUNCH_OK("adam");
    return 0;
}
