// Plain argument structs of the optimizer step (apply_common.hpp has the device code): TensorDesc / ApplyArgs of apply_kernel, and the
// task table / arguments of the deferred step that the rows kernel carries in its RowsArgs.
#pragma once
#include <stdint.h>

namespace dvae {
namespace fused {

struct TensorDesc {
    int64_t off;          // float offset in the flat parameter buffer
    int32_t rows, cols;
    // kernel-layout ("fragment-major") copies: element (row, col) of a [rows][ns * KSTEP] matrix sits at
    //   off + (((col / KSTEP) * nt + row / 32) * 64 + ((col % KSTEP) / E) * 32 + row % 32) * E + col % E      (nt = row tiles)
    int64_t sf_off;       // forward copy (A operand of the layer), -1 = none
    int32_t sf_nt, sf_split, sf_gap, sf_roff;   // column c -> c (c < split) or c + gap; row r -> r + roff
    int32_t sf_ld, st_ld;                       // row strides of the row-major variant (WFRAG == false)
    int64_t st_off;       // transposed copy (A operand of the backward-data product), -1 = none
    int32_t st_nt, st_roff, st_cmax;            // element (r, c < cmax) -> row c, column r + roff
    int32_t sf_f16_cols;                        // columns c < this of the FORWARD copy hold split-fp16 planes of W * 2^6 (fused_tiles.hpp: struct X16); 0 = none
    // rows3 kernel (train_rows3.hip): fragment-major copies for v_mfma_f32_16x16x32 -- 16-row tiles, 32-deep k-steps: element (row, col) at
    //   off + (((col / 32) * nt + row / 16) * 64 + ((col % 32) / 8) * 16 + row % 16) * 8 + col % 8          (nt = 16-row tiles)
    // kind16 != 0 selects it for both copies of the tensor.  rowmap (forward copy) / trowmap (transposed copy): 1, 2 = the latent heads'
    // interleaved row order (row k of mu / log_var -> tile k / 8, row 4 ((k % 8) / 2) + (k % 2) [+ 2 for log_var]): a lane of the 16 x 16 C
    // tile then holds mu_k and log_var_k of the same k
    int32_t kind16, rowmap, trowmap, pad2;
};

struct ApplyArgs {
    float* p; float* m; float* v;
    const float* slabs; int64_t slab_stride; int nslabs;
    const TensorDesc* tensors; int ntensors;
    const unsigned char* chunk_tensor; int64_t n_params;
    void* wcopy; int64_t wpl;      // kernel-layout weight copies; elements between the hi and lo planes (NP == 2)
    float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, gscale;
    const double* partials; int npartials; int64_t B; float* losses3; double* accum;
    int info; float alpha, beta, gamma;
};

struct DeferTask { int32_t tensor, r0, cbeg, cend, kind, pad0, pad1, pad2; };   // kind 0: rows r0.., parameter columns cbeg.. (< cend) of `tensor`; kind 1: elements cbeg.. (< cend)
constexpr int DEFER_SHARDS = 32;
struct DeferArgs {
    int on;                       // 1: the deferred protocol (the loss scalars come from an extra workgroup of the weight-gradient launch); 0: off
    int have;                     // a pending update exists: run the tasks and the arrival protocol
    int diag;                     // timing diagnostics (wrong results): bit 0 skip the tasks, bit 1 skip the arrival wait, bit 2 skip the loss finalisation
    ApplyArgs a;
    const DeferTask* tasks; int ntasks;
    unsigned* shard;              // DEFER_SHARDS counters, 32 words (128 bytes) apart
    unsigned* done;               // (unused)
    unsigned* err;                // sticky error word (a bounded wait ran out)
    unsigned seq_arrive;          // number of this launch among the workspace's deferred launches that carry an update (1, 2, ...)
    unsigned seq_done;            // ... among all its deferred launches
    unsigned long long timeout_ticks;   // 100 MHz wall clock
};


}  // namespace fused
}  // namespace dvae
