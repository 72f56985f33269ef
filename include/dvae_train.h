/* dvae_train.h -- C ABI of the fused train step (libdvae_hip.so).
 *
 * One step = the loop body of the reference's training scripts
 *   scripts/training_M1.py:134-139   r, mu, logvar = model(x); elbo; backward; Adam.step; zero_grad
 *   scripts/training_M2.py:142-147   the same with model(x, y)
 *   scripts/training_M2_info_vad.py:159-198   M2_info: + classifier(x), auxiliary(z), two BCE terms, two Adam
 *       groups incl. the reference's gradient-accumulation quirk (the auxiliary net sees (gamma - beta) * dBCE)
 * for the only geometry those scripts use (x_dim 513, z_dim 16, h_dim [128, 128]; y_dim 0, 1 or 513),
 * in three launches:
 *   rows kernel  : per 32-frame tile, the whole forward (encoder, reparametrisation, decoder),
 *                  the Itakura-Saito + KL sums and every backward-data product, on chip;
 *                  writes the operands of the weight gradients, transposed, to a stash;
 *   wgrad kernel : dW = dPre^T @ In for all layers (reduction over frames), k-split slabs;
 *   apply kernel : sums the slabs, torch.optim.Adam update of the fp32 master parameters,
 *                  refreshes the kernel-layout weight copies, finalises the loss scalars.
 * All state lives in caller-owned device buffers (PyTorch tensors): parameters / Adam
 * moments as flat fp32 buffers laid out per dvae_train_plan_t, scratch in `ws`.
 */
#ifndef DVAE_TRAIN_H
#define DVAE_TRAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DVAE_MODEL_M1 = 1, DVAE_MODEL_M2 = 2, DVAE_MODEL_M2_INFO = 3,
       /* encoder on x alone, decoder on [z | y]: the VAE body of DeepGenerativeModel_v3 / _v5 (packages/models/models.py:245-297, 437-444)
        * for the whole-model autograd path of the drop-in modules; y_dim 1, bf16 / bf16x3 operands (the 8-wave rows kernel) */
       DVAE_MODEL_M2_DEC = 4 };
/* Matrix-core operand policies (accumulation and master weights are always fp32):
 *   F32    exact fp32 MFMA (v_mfma_f32_32x32x2_f32)                       -- parity mode, 1/16 of the bf16 MFMA rate
 *   BF16   one bf16 per operand                                            -- fastest; weight gradients within ~4e-2 of their maximum
 *   BF16X3 split bf16: operand = hi + lo planes, product = hi*hi + lo*hi + hi*lo  -- parity-grade throughput mode
 *          (losses ~1e-7, gradients ~1e-4 of their maximum vs float64; tools/exp_precision.py), 3/16 of the F32 cost */
enum { DVAE_PREC_F32 = 0, DVAE_PREC_BF16 = 1, DVAE_PREC_BF16X3 = 2 };
#define DVAE_TRAIN_MAX_TENSORS 32

typedef struct {
    /* inputs (echoed) */
    int32_t model;            /* DVAE_MODEL_* */
    int32_t y_dim;            /* 0 (M1), 1 or 513 (M2), 1 (M2_info, M2_DEC) */
    int32_t precision;        /* DVAE_PREC_*: matrix-core operand type (accumulation is always fp32) */
    int32_t ksplit;           /* frame-axis slices of the weight-gradient reduction */
    int64_t B;                /* frames per step */
    /* flat fp32 parameter buffer: the reference's state_dict tensors, in state_dict order,
       each contiguous in nn.Linear layout [out, in] at a 256-byte aligned offset */
    int32_t n_tensors;
    int32_t reserved0;        /* schedule of the weight-gradient launch, set by dvae_train_plan: 0 = ksplit uniform frame slices; low 30 bits > 0 = class-sliced
                                 (workgroups of the launch; ksplit = the largest slice count of any block = slabs to sum); bit 30 = two launches
                                 (DVAE_EXCHANGE_GROUPS=2, dvae_train_grads_group).  Callers treat it as opaque. */
    int64_t n_params;                                   /* flat length in floats (with alignment gaps) */
    int64_t tensor_offset[DVAE_TRAIN_MAX_TENSORS];      /* in floats */
    int32_t tensor_rows[DVAE_TRAIN_MAX_TENSORS];
    int32_t tensor_cols[DVAE_TRAIN_MAX_TENSORS];        /* 1 for biases */
    /* scratch */
    int64_t workspace_bytes;
    int64_t grad_offset_bytes;   /* ws + this = ksplit slabs of n_params floats (slab 0 = flat gradient after reduce) */
    int64_t Bp;                  /* frames padded to the stash row length */
    int64_t rows_grid;           /* workgroups of the rows kernel */
    /* algorithmic work per step, for roofline accounting */
    double  flops_per_step;
    double  min_hbm_bytes_per_step;
    /* M2_info loss weights (scripts/training_M2_info_vad.py:53-55): enc_loss = ELBO + alpha*BCE(clf(x), y)
       - beta*BCE(aux(z), y), aux_loss = gamma*BCE(aux(z.detach()), y).  dvae_train_plan sets the script's
       defaults (0, 10, 1); the caller may overwrite them before dvae_train_init. */
    double  info_alpha, info_beta, info_gamma;
    /* Reparametrisation noise drawn inside the rows kernel when eps_noise == NULL: Philox4x32-10 keyed by rng_seed,
       counter (frame index, rng_step, draw); standard normals by Box-Muller.  The caller bumps rng_step every step
       (dvae_train_step uses its `step` argument instead).  dvae_train_noise reproduces the same numbers. */
    uint64_t rng_seed;
    uint64_t rng_step;
    /* Optional running sums for epoch logging (the scripts' `total_elbo += loss.item()`, training_M2.py:148-150, without a
       device-to-host sync per step): device pointer to 8 doubles, or 0.  dvae_train_apply / dvae_train_eval add the
       step's loss scalars (same order as losses3) to it. */
    uint64_t loss_accum;
    /* Optional gather (epoch shuffle without copying the data set): device pointer to B int64 row numbers, or 0.
       When set, x and y are the whole frame store (rows of ldx / ldy floats) and frame b of the step is row
       row_index[b].  row_count = rows of that store: an index outside [0, row_count) is never dereferenced -- the tile
       loader reads row 0 instead and adds one to the int32 device counter bad_row_counter (0 = none), which the
       caller inspects when it next synchronises. */
    uint64_t row_index;
    int64_t  row_count;
    uint64_t bad_row_counter;
    /* rows kernel generation chosen by dvae_train_plan: 2 = 8-wave chain + helper kernel (csrc/train_rows2.hip; M1 / M2 with
       bf16 or bf16x3 operands), 1 = 4-wave kernel (csrc/train_fused.hip; every model and policy).  Environment override
       DVAE_ROWS=1 at plan time. */
    int32_t  rows_kernel;
    int32_t  reserved1;
} dvae_train_plan_t;

/* Fill `plan` for (model, y_dim, precision, B).  ksplit_hint 0 = choose.  Returns DVAE_E_UNSUPPORTED
 * for geometries the fused kernels do not cover (the layer-level path of dvae.h covers those). */
int dvae_train_plan(int model, int y_dim, int precision, int64_t B, int ksplit_hint, dvae_train_plan_t* plan);

/* One-time setup of `ws` (zero fill, tile table, kernel-layout weight copies from `params`).
 * Synchronises the stream once (host-to-device copy of the tile table). */
int dvae_train_init(const dvae_train_plan_t* plan, const float* params, void* ws, void* stream);

/* rows kernel + wgrad kernel (+ slab reduction into slab 0 when reduce_slabs != 0).
 * x [B, 513] (ldx), y [B, y_dim] (ldy, may be NULL for M1), eps_noise [B, 16]: fp32 device tensors; eps_noise NULL = draw
 * the noise in the kernel (the reference draws torch.randn(mu.size()) on the host and copies it, models.py:10-13).
 * elbo_eps is the `eps` of elbo(x, r, mu, logvar, eps) (packages/models/utils.py:73). */
int dvae_train_grads(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                     const float* y, int ldy, const float* eps_noise, float elbo_eps, int reduce_slabs, void* stream);

/* The gradient pass of a step in TWO calls, so that the exchange of the first part of the flat gradient can run while the second part is
 * computed (no reference call site: scripts/training_M2.py:31-33 is single-device).  For a plan made while
 * DVAE_EXCHANGE_GROUPS=2 is set: group 0 = rows kernel + the weight-gradient launch of the decoder-side tensors (and M2_info's side nets) --
 * floats [tensor_offset[8], n_params) of the flat gradient; group 1 = the launch of the encoder's tensors, floats [0, tensor_offset[8]);
 * group 1 must follow a group 0 call on the same workspace.  reduce_slabs sums the slabs of the group's range into slab 0.
 * dvae_train_grads on such a plan runs both launches.  dvae_train_group_range: the range of a group (-1: everything), *ngroups = 2 or 1. */
int dvae_train_grads_group(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx, const float* y, int ldy,
                           const float* eps_noise, float elbo_eps, int group, int reduce_slabs, void* stream);
int dvae_train_group_range(const dvae_train_plan_t* plan, int group, int64_t* lo, int64_t* hi, int* ngroups);

/* apply kernel: g = grad_scale * sum of n_slabs slabs; Adam(lr, beta1, beta2, adam_eps) at `step` (1-based) on
 * params/m/v; refresh weight copies; losses3 = {recon + KL, recon, KL} (means over the B local frames);
 * for M2_info the buffer holds 8 floats: {ELBO, recon, KL, enc_loss, classif_loss, aux_loss, aux_enc_loss, 0}. */
int dvae_train_apply(const dvae_train_plan_t* plan, float* params, float* m, float* v, void* ws, int n_slabs,
                     int step, double lr, double beta1, double beta2, double adam_eps, double grad_scale,
                     float* losses3, void* stream);

/* dvae_train_grads + dvae_train_apply (single-GPU step). */
int dvae_train_step(const dvae_train_plan_t* plan, float* params, float* m, float* v, void* ws,
                    const float* x, int ldx, const float* y, int ldy, const float* eps_noise, float elbo_eps,
                    int step, double lr, double beta1, double beta2, double adam_eps, float* losses3, void* stream);

/* The same step in TWO launches (round 4): the optimizer update of step n is deferred into the opening of step n + 1's rows kernel, where
 * the GEMM waves would otherwise wait for the x tile (csrc/apply_common.hpp), and a step's loss scalars are finalised by its own rows
 * kernel.  After this call returns, losses3 (stream-ordered) holds step n's losses and the gradient slabs step n's gradient, but
 * params / m / v and the weight copies still hold the state BEFORE step n's update: it is PENDING on `ws` until the next
 * dvae_train_step_deferred on the same workspace or dvae_train_flush.  Every other entry point of this header flushes first by itself;
 * a caller that reads or writes params / m / v directly must call dvae_train_flush before (dvae_train_repack refuses while an update is
 * pending).  Results are bit-identical to dvae_train_step (same slab sums, same element arithmetic).  Falls back to dvae_train_step when
 * the plan cannot defer (M2_info, the 4-wave rows kernel, grids smaller than the update's task list or larger than the CUs) and unless
 * DVAE_DEFER_APPLY=1 is set: measured on the MI355X the two-launch step is bit-identical and NOT faster, so it is opt-in.  The same params / m / v pointers and hyper-parameter semantics as dvae_train_step. */
int dvae_train_step_deferred(const dvae_train_plan_t* plan, float* params, float* m, float* v, void* ws,
                             const float* x, int ldx, const float* y, int ldy, const float* eps_noise, float elbo_eps,
                             int step, double lr, double beta1, double beta2, double adam_eps, float* losses3, void* stream);
/* Apply the pending update of `ws`, if any (one apply launch on `stream`). */
int dvae_train_flush(const dvae_train_plan_t* plan, void* ws, void* stream);
/* 1 when an update is pending on `ws`. */
int dvae_train_pending(const void* ws);
/* 1 when dvae_train_step_deferred would defer on this plan / workspace / device / environment (otherwise it is dvae_train_step). */
int dvae_train_can_defer(const dvae_train_plan_t* plan, const void* ws);

/* Validation pass of the scripts (scripts/training_M2.py:176-193: forward + elbo, no backward, no update):
 * rows kernel + loss finalisation only.  losses3 as for dvae_train_apply. */
int dvae_train_eval(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                    const float* y, int ldy, const float* eps_noise, float elbo_eps, float* losses3, void* stream);

/* Whole-model autograd path of the drop-in modules (packages/models/models.py: VariationalAutoencoder.forward,
 * DeepGenerativeModel.forward -- reference models.py:172-179, 200-203): ONE launch for the model forward instead of one per
 * nn.Linear, and the backward pass as rows kernel + weight-gradient kernel + slab sum, under stock torch autograd / torch.optim.Adam
 * (scripts/training_M2.py:142-147).  M1 / M2 with bf16 / bf16x3 operands (plan->rows_kernel == 2), no gather table.
 *   dvae_module_forward : (repack != 0: rebuild the kernel-layout weight copies from `params` first) r = model(x, y) [B, 513] (ld_r),
 *                         mu, log_var, z [B, 16] (out_z may be NULL); eps_noise [B, 16] is the reparametrisation noise.
 *   dvae_module_backward: recomputes the forward, then the backward from the upstream gradients g_r [B, 513] (ld_gr), g_mu, g_lv,
 *                         g_z [B, 16] (each may be NULL = zero); grad_flat[n_params] (plan layout) = (accumulate ? grad_flat : 0) + dL/dparams. */
int dvae_module_forward(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                        const float* y, int ldy, const float* eps_noise, float* out_r, int ld_r, float* out_mu,
                        float* out_lv, float* out_z, int repack, void* stream);
int dvae_module_backward(const dvae_train_plan_t* plan, const float* params, void* ws, const float* x, int ldx,
                         const float* y, int ldy, const float* eps_noise, const float* g_r, int ld_gr,
                         const float* g_mu, const float* g_lv, const float* g_z, float* grad_flat, int accumulate, void* stream);

/* The [B, 16] standard normals the rows kernel draws for (plan->rng_seed, step) when eps_noise == NULL. */
int dvae_train_noise(const dvae_train_plan_t* plan, uint64_t step, float* eps_out, void* stream);

/* Rebuild the kernel-layout weight copies after `params` was written from outside (load_state_dict). */
int dvae_train_repack(const dvae_train_plan_t* plan, const float* params, void* ws, void* stream);

/* Per-kernel device time of the last `dvae_train_*` calls made with profiling enabled:
 * enable != 0 brackets each launch with hipEvents on its stream (a few us of host cost each).
 * dvae_train_profile_read synchronises and returns accumulated ms and launch counts
 * for {rows, wgrad, reduce, apply}, then clears them. */
int dvae_train_profile(int enable);
/* Diagnostic builds only: buf = device array of rows_grid * 32 uint64, filled with 100 MHz wall-clock stamps at the
 * phase boundaries of the rows kernel (NULL switches it off).  Never enabled by bench.py's timed region. */
int dvae_train_debug_stamps(void* buf);
int dvae_train_profile_read(double ms[4], int64_t calls[4]);

/* ---- direct gradient exchange of the data-parallel step (no reference call site: scripts/training_M2.py:31-33 is
 * single-device).  One launch per rank on the step's stream: slab sum -> reduce-scatter by pull -> all-gather by push over peer pointers
 * (hipIpc-mapped exchange buffers; xGMI on a multi-GPU node), see csrc/allreduce.hip.  Every wait for a peer is bounded by wall time
 * (dvae_comm_set_timeout_ms; default 20 s, env DVAE_COMM_TIMEOUT_MS): a missing or late peer ends the launch on EVERY rank with `out`
 * filled with NaN and dvae_comm_status() != 0 from then on, never with a hang and never with a stale sum.  The exchange buffer is
 * fine-grained device memory; where the platform has none, dvae_comm_create fails (no coarse-grained fallback).
 *   dvae_comm_create   allocates this rank's exchange buffer for n_floats and returns its IPC handle (DVAE_IPC_HANDLE_BYTES bytes);
 *   the caller gathers the handles of all ranks in rank order (e.g. torch.distributed.all_gather_object) and passes them to
 *   dvae_comm_connect; every rank must call dvae_allreduce_flat the same number of times.
 *   dvae_allreduce_flat: out[0 .. n) = sum over ranks of (sum over k < n_slabs of slabs[k * slab_stride + i]); `out` may alias slab 0. */
#define DVAE_IPC_HANDLE_BYTES 128   /* hipIpc handle + owning process + PCI address of the device */
typedef struct dvae_comm dvae_comm_t;
int dvae_comm_create(int rank, int world, int64_t n_floats, dvae_comm_t** out, unsigned char handle[DVAE_IPC_HANDLE_BYTES]);
int dvae_comm_connect(dvae_comm_t* c, const unsigned char* handles /* world x DVAE_IPC_HANDLE_BYTES, rank order */);
int dvae_allreduce_flat(dvae_comm_t* c, const float* slabs, int n_slabs, int64_t slab_stride, float* out, void* stream);
int dvae_comm_set_timeout_ms(dvae_comm_t* c, int64_t ms);   /* bound of every in-kernel wait of the following launches (0: one poll) */
int dvae_comm_status(dvae_comm_t* c, int* failed);      /* synchronises; *failed = 1 when a bounded wait expired on ANY rank since creation */
int dvae_comm_destroy(dvae_comm_t* c);

#ifdef __cplusplus
}
#endif
#endif /* DVAE_TRAIN_H */
