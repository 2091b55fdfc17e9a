// Host-side helpers shared by the translation units of libnoiseflow_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// record a thread-local error message (returned by nf_last_error) and hand back `code`
int nf_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int nf_fail_hip(hipError_t e, const char *what);

// geometry / device of a handle (nf_hostfed.hip sizes its staging from them)
struct nf_handle;
int nf_handle_geometry(const nf_handle *h, int32_t *H, int32_t *W, int32_t *device);
// frees the handle's host-fed pipeline, if one was created (nf_destroy)
void nf_hostpipe_release(nf_handle *h);

// Evaluation under batch statistics at the coupling widths / patch sizes the fused kernels' statistics passes do not take
// (nf_train.hip: a trimmed trainer whose couplings run on the matrix-core GEMMs of nf_train_mm.h); owned by the handle's
// batch-statistics state, freed with nf_trainer_destroy
#include "../../include/noiseflow_hip.h"
struct nf_bs_wide_args {
    int direction;             // 0: NLL, 1: sampling
    const float *in;           // x (NLL) / epsilon (sampling; NULL: the in-kernel Philox draw keyed by seed, patch_base + b, pixel)
    const float *y;
    int64_t B;
    const nf_cond *cond;
    uint64_t seed;
    int64_t patch_base;
    float in_scale;            // sampling: the temperature; NLL: 1
    float *out;                // z_out (may be NULL) / x_out
    float *nll_out, *sd_out, *ld_out;
    double *sums;              // DEVICE double[3], accumulated into
    bool prior;
    float *moments_out;        // HOST [n_couplings][4][w] or NULL
    nf_allreduce_fn sync_fn;
    void *sync_user;
    double *sync_buf;
    int sync_world;
};
int nf_bs_wide_create(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params, int64_t max_batch,
                      nf_trainer **out);
int64_t nf_bs_wide_capacity(const nf_trainer *t);
int nf_bs_wide_run(nf_trainer *t, const nf_bs_wide_args &a, hipStream_t st);
