/* dvae_mcem.h -- C ABI of the MCEM speech-enhancement loop (libdvae_hip.so).
 *
 * Reference: packages/models/mcem.py (classes EM, MCEM_M1, MCEM_M2, MCEM_M2v2, MCEM_M2v3), driven by
 * scripts/evaluate_ntcd_M1.py / _M2.py / _M2_info_vad.py (`mcem.init_parameters(...)`, `mcem.run()`).
 * For the geometry those scripts use (F = 513 bins, latent 16, decoder 128-128-513, y_dim 0 / 1 / 513).
 *
 * Array shapes are the reference's own (row-major, fp32): X2, Vb, WFs, WFn (F, N); Vs (R, F, N);
 * Z (16, N); y (y_dim, N); g (N); W (F, K); H (K, N); sampled latents (N, R, 16).
 * All pointers are caller-owned device memory; calls enqueue on `stream` and do not synchronise.
 * Every random number is an argument: the caller draws `noise` and `logu` (the reference draws them with
 * torch.randn / torch.rand inside the loop, mcem.py:244,257).
 */
#ifndef DVAE_MCEM_H
#define DVAE_MCEM_H

#include <stddef.h>
#include <stdint.h>
#include "dvae_train.h"      /* DVAE_PREC_* */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t y_dim;            /* label rows fed to the decoder next to z: 0 (MCEM_M1), 1..16 or 513 */
    int32_t precision;        /* DVAE_PREC_F32 (exact fp32 products), DVAE_PREC_BF16X3 (split-bf16 operands: 16 mantissa bits, three MFMAs
                                 per product -- parity grade at a fraction of the fp32 matrix cost) or DVAE_PREC_BF16 (one bf16 per operand) */
    int32_t x_dim, z_dim, h_dim;   /* echoed: 513, 16, 128 */
    int32_t reserved;
    int64_t weights_bytes;    /* size of the kernel-layout decoder copy filled by dvae_mcem_pack */
} dvae_mcem_plan_t;

int dvae_mcem_plan(int y_dim, int precision, dvae_mcem_plan_t* plan);

/* Measurement only (no reference counterpart): while buf != NULL the weight-stationary chain kernel adds, per (workgroup, wave), the shader
 * clocks it spends in each of its 9 chain-step phases into buf[(workgroup * 4 + wave) * 16 + phase] (tools/stamp_mcem.py); NULL switches it off. */
int dvae_mcem_debug_stamps(void* buf);

/* Kernel-layout copy of the decoder (vae.decoder: hidden.0 [128][16+y_dim], hidden.1 [128][128], reconstruction
 * [513][128], nn.Linear layouts with row strides ld*; packages/models/models.py:108-122).  Call again after the
 * weights change. */
int dvae_mcem_pack(const dvae_mcem_plan_t* plan, const float* W3, int ld3, const float* b3, const float* W4, int ld4,
                   const float* b4, const float* W5, int ld5, const float* b5, void* weights, void* stream);

/* sample_posterior (mcem.py:207-277, 372-448): `nit` = nsamples + burnin Metropolis-Hastings iterations of the
 * random walk Z' = Z + sqrt(var_rw) * noise[m], accepted where logu[m] < log acceptance ratio.
 *   noise (nit, 16, N), logu (nit, N): the draws; Z0 (16, N) start; y (y_dim, N) or NULL when y_dim == 0.
 *   Zs (N, nit - burnin, 16): the kept samples (the reference's Z_sampled_t).
 *   Vs (nit - burnin, F, N) or NULL: decoder variances of the kept samples (compute_Vs, mcem.py:280-290).
 *   acc_logratio (nit, N), accepted (nit, N) bytes: optional diagnostics (NULL to skip). */
int dvae_mcem_sample(const dvae_mcem_plan_t* plan, const void* weights, const float* Z0, const float* y, const float* g,
                     const float* Vb, const float* X2, const float* noise, const float* logu, int nit, int burnin,
                     float var_rw, int64_t N, float* Zs, float* Vs, float* acc_logratio, unsigned char* accepted, void* stream);

/* compute_Vs alone (mcem.py:280-290, 451-461): Vs (R, F, N) = decoder([Zs[:, r, :] | y]). */
int dvae_mcem_decode(const dvae_mcem_plan_t* plan, const void* weights, const float* Zs, const float* y, int R, int64_t N,
                     float* Vs, void* stream);

/* EM.M_step (mcem.py:91-153) + compute_expected_neg_log_like (mcem.py:69-71).  In place: W, H (stored normalised),
 * g, and Vb (the product of the un-normalised factors, which is what the reference keeps for the next E-step).
 * cost: 1 float or NULL.  K <= 16. */
size_t dvae_mcem_m_step_workspace_bytes(int64_t N, int K, int U);
int dvae_mcem_m_step(const float* X2, const float* Vs, int R, int64_t N, int K, float* W, float* H, float* g, float* Vb,
                     float* cost, void* workspace, void* stream);

/* The same for U utterances laid side by side on the frame axis (the reference runs one utterance per process,
 * scripts/evaluate_ntcd_M2.py:317-327; one MI355X holds dozens of them in one launch).  Utterance u owns frames
 * seg_start[u] .. seg_start[u] + seg_count[u] - 1; seg_start[u] is a multiple of 32, N (the padded total) too;
 * tile_seg[t] = utterance of frames 32t .. 32t+31.  Padding frames must hold finite values (e.g. X2 = Vb = g = 1)
 * and are never written.  W is (U, F, K), cost (U); H, g, Vb, X2, Vs as above with N = padded total.
 * The three tables are device int32 arrays.  dvae_mcem_sample / _decode / _wiener act frame by frame and need
 * no batched form. */
int dvae_mcem_m_step_batch(const float* X2, const float* Vs, int R, int64_t N, int K, int U, const int* seg_start,
                           const int* seg_count, const int* tile_seg, float* W, float* H, float* g, float* Vb,
                           float* cost, void* workspace, void* stream);

/* One EM iteration, the body of EM.run's loop (mcem.py:156-160: E_step -- the chain of dvae_mcem_sample with the decoder variances of its
 * kept samples, mcem.py:207-218 / 372-383 -- then `self.Z = last kept sample`, M_step, cost) as ONE host call, so that the loop is bound by
 * its kernels and not by the interpreter between them.  Z (16, N) is read as the chain's start and overwritten with the last kept sample;
 * g, Vb, W, H are updated in place as by dvae_mcem_m_step_batch; Zs (N, nit - burnin, 16) and Vs (nit - burnin, F, N) are caller-owned
 * scratch that holds the iteration's samples / variances afterwards; cost (U) as dvae_mcem_m_step_batch.  U = 1 with NULL segment tables is
 * the single-utterance loop of scripts/evaluate_ntcd_M2.py:201-205. */
int dvae_mcem_em_iteration(const dvae_mcem_plan_t* plan, const void* weights, float* Z, const float* y, float* g, float* Vb,
                           const float* X2, const float* noise, const float* logu, int nit, int burnin, float var_rw, int64_t N,
                           int K, int U, const int* seg_start, const int* seg_count, const int* tile_seg, float* W, float* H,
                           float* Zs, float* Vs, float* cost, void* workspace, void* stream);

/* dvae_mcem_em_iteration with TWO launches of the M-step instead of three (EM.run's loop, mcem.py:156-160, for callers that read the cost
 * after the loop, as the drop-in classes' run() does): W is normalised inside the frames kernel, and the iteration's cost stays in the
 * workspace (the frames kernel's partial sums) until the NEXT lazy call writes it to cost_prev (U floats; NULL in the first call) from
 * its W update, or dvae_mcem_cost_flush after the last.  Same arithmetic and bits as dvae_mcem_em_iteration.  Register-resident M-step only
 * (at most 10 kept samples, rank 10): DVAE_E_UNSUPPORTED otherwise.  The workspace must stay untouched between the calls. */
int dvae_mcem_em_iteration_lazy(const dvae_mcem_plan_t* plan, const void* weights, float* Z, const float* y, float* g, float* Vb,
                                const float* X2, const float* noise, const float* logu, int nit, int burnin, float var_rw, int64_t N,
                                int K, int U, const int* seg_start, const int* seg_count, const int* tile_seg, float* W, float* H,
                                float* Zs, float* Vs, float* cost_prev, void* workspace, void* stream);
int dvae_mcem_cost_flush(int R, int64_t N, int K, int U, const int* seg_start, const int* seg_count, float* cost, void* workspace, void* stream);

/* compute_WF (mcem.py:321-327): WFs = mean_r(g Vs / Vx), WFn = mean_r(Vb / Vx), both (F, N). */
int dvae_mcem_wiener(const float* Vs, int R, int64_t N, const float* g, const float* Vb, float* WFs, float* WFn, void* stream);

#ifdef __cplusplus
}
#endif
#endif
