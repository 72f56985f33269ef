/* dvae.h -- C ABI of libdvae_hip.so: the MI355X (gfx950) hot path of
 * sp-uhh/disentangled-vae (VAE train step + STFT/ISTFT).
 *
 * The reference has no FFI of its own (pure Python on torch/librosa): the
 * drop-in boundary is its Python import surface.  Each entry
 * point below states which reference lines it replaces; the Python shells in
 * packages/ (same names/signatures as the reference) bind them with ctypes,
 * see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensor
 *     storage, passed as data_ptr()), fp32 row-major unless stated;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     kernels are enqueued on it and the call never synchronises;
 *   - the library allocates no user-visible memory; scratch comes from the
 *     caller (sizes from the *_workspace_bytes queries);
 *   - return 0 on success, non-zero (hipError_t or DVAE_E_*) on failure;
 *     dvae_last_error() returns a thread-local message; no exception crosses.
 */
#ifndef DVAE_H
#define DVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVAE_ABI_VERSION 1

enum { DVAE_E_BADARG = 1001, DVAE_E_WORKSPACE = 1002, DVAE_E_UNSUPPORTED = 1003 };

/* activations of the fused Linear+act kernels */
enum { DVAE_ACT_NONE = 0, DVAE_ACT_TANH = 1, DVAE_ACT_RELU = 2, DVAE_ACT_SIGMOID = 3, DVAE_ACT_EXP = 4 };

int         dvae_abi_version(void);
const char* dvae_last_error(void);
/* 1 for the diagnostic build (-DDVAE_DIAG: product kernels + the measured-slower alternates behind their A/B switches), 0 for the default one. */
int dvae_build_has_diag(void);
/* number of HIP devices visible to the library (0 when none): lets the host fail loudly */
int         dvae_device_count(void);

/* ---------------------------------------------------------------------------
 * Layer-level ops: what packages/models/models.py modules call per nn.Linear.
 * ------------------------------------------------------------------------- */

/* out[B,N] = act([x0 | x1] @ W^T + bias)            (x1 may be NULL, k1 = 0)
 * replaces `x = torch.tanh(layer(x))` / `torch.relu(layer(x))` / `torch.sigmoid(...)` /
 * `torch.exp(self.reconstruction(x))` and the `torch.cat([x, y], dim=1)` in front of it:
 * packages/models/models.py:57-63, 102-105, 119-122, 201-202.  W is nn.Linear layout [N, k0+k1]. */
int dvae_linear_act_fwd(const float* x0, int k0, int ld0, const float* x1, int k1, int ld1,
                        const float* W, int ldw, const float* bias,
                        float* out, int ldo, int64_t B, int N, int act, void* stream);

/* dpre[B,N] = dout * act'(out)  (act' expressed through the OUTPUT: tanh 1-o^2, relu o>0,
 * sigmoid o(1-o), exp o); autograd of the activations above. */
int dvae_act_bwd(const float* dout, int ldd, const float* out, int ldo, float* dpre, int ldp,
                 int64_t B, int N, int act, void* stream);

/* din[B,K] (+)= dpre[B,N] @ W[:, koff:koff+K]     (autograd of F.linear wrt its input);
 * accumulate != 0 adds into din (sum of the mu / log_var heads, models.py:34-36). */
int dvae_linear_bwd_data(const float* dpre, int ldp, const float* W, int ldw, int koff,
                         float* din, int ldi, int64_t B, int N, int K, int accumulate, void* stream);

/* dW[N, k0+k1] = dpre^T @ [x0 | x1],  db[N] = colsum(dpre)   (autograd of F.linear wrt W, b).
 * Reduction over the B frames is split over `ksplit` workgroup slices combined with fp32
 * atomics when ksplit > 1 (pass 0 to let the library choose).  db may be NULL. */
int dvae_linear_bwd_weight(const float* dpre, int ldp, const float* x0, int k0, int ld0,
                           const float* x1, int k1, int ld1, float* dW, int ldw, float* db,
                           int64_t B, int N, int ksplit, void* stream);
/* The same with a DETERMINISTIC combination of the slices (what the drop-in modules call: the reference's own GPU path -- cuBLAS -- returns
 * the same bits for the same inputs): every slice writes its partial matrix into `workspace` (dvae_linear_bwd_weight_workspace_bytes; may be
 * NULL when that is 0) and a second launch sums the slices in ascending order. */
size_t dvae_linear_bwd_weight_workspace_bytes(int64_t B, int N, int Kin, int ksplit);
int dvae_linear_bwd_weight_det(const float* dpre, int ldp, const float* x0, int k0, int ld0,
                               const float* x1, int k1, int ld1, float* dW, int ldw, float* db,
                               int64_t B, int N, int ksplit, void* workspace, void* stream);

/* z = mu + exp(0.5*logvar) * eps     packages/models/models.py:9-22 (Stochastic.reparametrize) */
int dvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z,
                     int64_t n, void* stream);
/* dmu = dz ; dlogvar = dz * eps * 0.5 * exp(0.5*logvar) */
int dvae_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dmu,
                     float* dlogvar, int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * Losses: packages/models/utils.py
 * ------------------------------------------------------------------------- */

/* elbo(x, r, mu, logvar, eps) -> out3 = {recon+KL, recon, KL}   utils.py:73-76.
 * recon = mean_b sum_f (x/r - log(x+eps) + log r - 1); KL = -0.5 mean_b sum_k (lv - mu^2 - e^lv).
 * Also writes kl_b[B] = per-frame KL (models.py:165-167 `_kld_v2`) when kl_b != NULL.
 * ws: dvae_elbo_workspace_bytes(B) bytes of scratch. */
size_t dvae_elbo_workspace_bytes(int64_t B);
int dvae_elbo_fwd(const float* x, int ldx, const float* r, int ldr, const float* mu,
                  const float* logvar, float eps, int64_t B, int F, int Z,
                  float* out3, float* kl_b, void* ws, void* stream);
/* gradients of g2[0]*recon + g2[1]*KL wrt r, mu, logvar; g2 = DEVICE pointer to
 * {g_loss + g_recon, g_loss + g_kl} (the upstream grads of the three returned scalars). */
int dvae_elbo_bwd(const float* x, int ldx, const float* r, int ldr, const float* mu,
                  const float* logvar, const float* g2, int64_t B, int F, int Z,
                  float* dr, int lddr, float* dmu, float* dlogvar, void* stream);
/* The same with the three upstream gradients of elbo()'s outputs (loss, recon, KL) as device scalars, each may be NULL (= 0):
 * d/d recon = *g_loss + *g_recon, d/d KL = *g_loss + *g_kl -- no host-side arithmetic on them (autograd hands them over separately). */
int dvae_elbo_bwd3(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar,
                   const float* g_loss, const float* g_recon, const float* g_kl, int64_t B, int F, int Z,
                   float* dr, int lddr, float* dmu, float* dlogvar, void* stream);

/* binary_cross_entropy family, utils.py:55-63: variant 0 = (r, x), 1 = _v2 (targets 0.5),
 * 2 = _v3 (targets r).  out1 = -mean_b sum_j [t log(r+eps) + (1-t) log(1-r+eps)]. */
int dvae_bce_fwd(const float* r, const float* t, float eps, int64_t B, int Y, int variant,
                 float* out1, void* ws, void* stream);
/* dr = g * d(bce)/dr ; dt (variant 0 only, may be NULL) */
int dvae_bce_bwd(const float* r, const float* t, float eps, const float* g, int64_t B, int Y,
                 int variant, float* dr, float* dt, void* stream);

/* ---------------------------------------------------------------------------
 * The rest of the loss zoo of packages/models/utils.py (reference :65-118; the earlier M2v3 / M2v4 experiments train on them).
 * Row layout [B, F] with leading dimensions like elbo; partial sums in double, no atomics; `ws` as for elbo.
 * ------------------------------------------------------------------------- */
/* per-frame Itakura-Saito rows recon_rows[b] = sum_f (x/r - log(x+eps) + log r - 1) (utils.py:68-71, 79) and, when kl_rows != NULL,
 * kl_rows[b] = -0.5 sum_k (logvar - mu^2 - exp(logvar)) (utils.py:80): L_loss / ikatura_saito_divergence */
int dvae_isrows_fwd(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar, float eps,
                    int64_t B, int F, int Z, float* recon_rows, float* kl_rows, void* stream);
/* gradients from per-frame upstream gradients g_recon_rows / g_kl_rows ([B], either may be NULL = zeros); outputs may be NULL */
int dvae_isrows_bwd(const float* x, int ldx, const float* r, int ldr, const float* mu, const float* logvar,
                    const float* g_recon_rows, const float* g_kl_rows, int64_t B, int F, int Z,
                    float* dr, int lddr, float* dmu, float* dlogvar, void* stream);
/* binary_cross_entropy_2classes (utils.py:65-66): -mean_b sum_j [t log(r1+eps) + (1-t) log(r2+eps)] */
int dvae_bce2_fwd(const float* r1, const float* r2, const float* t, float eps, int64_t B, int Y, float* out1, void* ws, void* stream);
int dvae_bce2_bwd(const float* r1, const float* r2, const float* t, float eps, const float* g, int64_t B, int Y,
                  float* dr1, float* dr2, float* dt, void* stream);
/* squared-error losses (utils.py:107-118), mean_b sum_f |d|^2.  mode 0 mean_square_error_signal: d = (y - yhat) x;
 * mode 1 mean_square_error_mask: d = y - yhat (x unused); mode 2 magnitude_spectrum_approxiamation_loss: d = s - yhat x with
 * x, y (= s) complex64 [B, F] and a real mask yhat.  Backward: dyhat always; dy, dx for the real modes (NULL = not wanted). */
int dvae_sqerr_fwd(int mode, const void* x, const void* y, const float* yhat, int64_t B, int F, float* out1, void* ws, void* stream);
int dvae_sqerr_bwd(int mode, const void* x, const void* y, const float* yhat, const float* g, int64_t B, int F,
                   float* dyhat, float* dy, float* dx, void* stream);

/* ---------------------------------------------------------------------------
 * Optimiser: torch.optim.Adam(lr, betas) as the scripts construct it
 * (scripts/training_M2.py:122); op order of torch's single-tensor Adam.
 * ------------------------------------------------------------------------- */
int dvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr,
                   double beta1, double beta2, double eps, int step, double grad_scale, void* stream);

/* ---------------------------------------------------------------------------
 * STFT / ISTFT: packages/processing/stft.py:13-60, 63-99 (librosa semantics, center=False
 * or pre-padded input), periodic Hann.  Indexing (pad rule, frame count) is decided on the
 * host in double precision by the Python shell; the kernels take the final frame count.
 * ------------------------------------------------------------------------- */

/* x: n samples (already end-padded / centre-padded by the caller), in_f64 selects double input.
 * out: layout 0 = [nfft/2+1, T] interleaved complex64 (column = frame, the librosa layout; the
 *      same bytes are the legacy torch.stft real view [nfft/2+1, T, 2] of stft_pytorch,
 *      packages/processing/stft.py:145-151);
 *      layout 1 = [T, nfft/2+1] float32 power |.|^2 (one training frame per row,
 *      scripts/create_train_set.py:152 / scripts/reconstruct_M2.py:153);
 *      layout 2 = [T, nfft/2+1] interleaved complex64 (row = frame): the values of layout 0 in the MEMORY order of
 *      librosa's result (librosa.stft fills a Fortran-ordered [nfft/2+1, T] array, packages/processing/stft.py:50-57);
 *      the host returns its transpose view, so the caller sees the reference's shape and strides.
 * window: nfft doubles (device).  Power-of-two nfft in [8, 2048] runs the LDS FFT; any other
 * even nfft <= 2048 (e.g. the wrapper's never-used 800-sample default) runs a plain DFT. */
int dvae_stft(const void* x, int in_f64, int64_t n, const double* window, int nfft, int hop,
              int64_t T, void* out, int layout, void* stream);

/* The transform of stft_pytorch (packages/processing/stft.py:123-152: torch.stft of a float32 tensor with torch.hann_window) in ITS
 * arithmetic: window product, FFT and result in float32 (dvae_stft computes in double whatever the input type -- the arithmetic of
 * stft(), where librosa multiplies by a float64 window).  nfft 1024 / hop 256 only (every caller); window: nfft floats (device);
 * layout 2 = [T, 513] interleaved complex64, row = frame -- the memory of the legacy torch.stft result, whose [513, T, 2] real view
 * is the transpose view of it; layout 1 = [T, 513] float32 re * re + im * im (packages/data_handling.py:136).  Signal and result
 * below 2 GB each.  Any other size / layout: DVAE_E_ARG (use dvae_stft). */
int dvae_stft_f32(const float* x, int64_t n, const float* window, int nfft, int hop, int64_t T, void* out, int layout, void* stream);

/* S: complex64 [nfft/2+1, ldT] of which the first T columns (frames) are used;
 * y[out_len] float32 = overlap-add of window * irfft(S[:, t]) (float32 accumulation in frame
 * order, as librosa), divided by the window sum-square where it exceeds FLT_MIN, read from
 * sample `start` on, zero padded / trimmed to out_len.  ws: dvae_istft_workspace_bytes_hop(T, nfft, hop) bytes
 * (nfft 1024 / hop 256 -- every caller of the reference -- runs inverse FFT and overlap-add in one kernel: 16 bytes below 1024
 * frames, T * 513 complex64 from there on (dvae_istft transposes long bin-major input into it and runs the frame-major walk);
 * otherwise T * nfft doubles, which dvae_istft_workspace_bytes(T, nfft) always returns). */
size_t dvae_istft_workspace_bytes(int64_t T, int nfft);
size_t dvae_istft_workspace_bytes_hop(int64_t T, int nfft, int hop);
int dvae_istft(const void* S, int64_t T, int64_t ldT, const double* window, int nfft, int hop,
               int64_t start, float* y, int64_t out_len, void* ws, void* stream);
/* The same transform of a FRAME-major spectrogram: S complex64 [T, ldF], row t = frame t (its first nfft/2+1 entries) -- the
 * memory order of a Fortran-ordered [nfft/2+1, T] array such as librosa.stft / dvae_stft layout 2 return.  Results are
 * bit-identical to dvae_istft on the transposed array (packages/processing/stft.py:63-99). */
int dvae_istft_frames(const void* S, int64_t T, int64_t ldF, const double* window, int nfft, int hop,
                      int64_t start, float* y, int64_t out_len, void* ws, void* stream);

/* The inverse transform of istft_pytorch (packages/processing/stft.py:154-190: torch.istft of a complex64 tensor with
 * torch.hann_window, center handled by `start` / `out_len` as above) in ITS arithmetic: inverse FFT, window product, overlap-add and
 * the division by the window envelope in float32 (dvae_istft computes in double -- the arithmetic of istft(), librosa's).  nfft 1024 /
 * hop 256 only; window: nfft floats (device).  frames = 0: S is [513][ld] (bin-major, ld >= T), transposed into ws (T * 513
 * complex64) first; frames = 1: S is [T][ld] (frame-major, ld >= 513), read in place, ws may be null.  Any other size: DVAE_E_ARG. */
int dvae_istft_f32(const void* S, int64_t T, int64_t ld, int frames, const float* window, int nfft, int hop,
                   int64_t start, float* y, int64_t out_len, void* ws, void* stream);

/* ---------------------------------------------------------------------------
 * Fused train step (the build's own harness; mirrors scripts/training_M1.py:134-139,
 * scripts/training_M2.py:142-147, scripts/training_M2_info_vad.py:159-198).
 * Declared in dvae_train.h.
 * ------------------------------------------------------------------------- */

/* ---- frame store (GPU-resident replacement of HDF5CleanSpectrogramLabeledFrames, packages/data_handling.py:19-67) ----
 * dvae_transpose: the on-disk (F, N) matrix (one frame per column, scripts/create_train_set.py:116) -> frames-major
 *   [N][F] rows, the layout the train step reads.  rows/cols describe `in`; rows <= 2M.
 * dvae_gather_rows: dst[i] = src[idx[i]] (epoch shuffle: what DataLoader(shuffle=True) does frame by frame,
 *   scripts/training_M2.py:84-86).  Indices outside [0, nsrc) are skipped and counted in *bad_count (device int,
 *   may be NULL). */
int dvae_transpose(const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo, void* stream);
int dvae_gather_rows(const float* src, int64_t ld, int64_t nsrc, const int64_t* idx, int64_t n, int cols, float* dst,
                     int64_t ldd, int* bad_count, void* stream);

/* ---- label makers of the training-set builders (packages/processing/target.py) ----
 * dvae_vad_labels: clean_speech_VAD (target.py:5-56, center=False): vad[t] = E[t] > 10^vad_threshold * min_t E[t],
 *   E[t] = sum of squares of frame t (double accumulation); y: n samples (float32, or float64 when in_f64), the zero
 *   end-pad of `hop` samples is implied (frames may reach n + hop); frames from the host-side pad rule.  vad: (frames).
 * dvae_ibm_labels: clean_speech_IBM (target.py:58-70): mask = 20 log10(|S| + eps) > max - ibm_threshold over the whole
 *   (rows, cols) complex64 matrix S; vad_gate (cols) or NULL multiplies each column (noise_robust_clean_speech_IBM,
 *   target.py:72-104).  mask: (rows, cols) float32. */
size_t dvae_vad_workspace_bytes(int64_t frames);
int dvae_vad_labels(const void* y, int in_f64, int64_t n, int nfft, int hop, int64_t frames, double vad_threshold,
                    float* vad, void* workspace, void* stream);
size_t dvae_ibm_workspace_bytes(void);
int dvae_ibm_labels(const void* S, int64_t rows, int64_t cols, float eps, float ibm_threshold, const float* vad_gate,
                    float* mask, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVAE_H */
