/*
 * noiseflow_hip.h — C ABI of the MI355X-native Noise Flow bijector stack.
 *
 * This is the drop-in boundary for the ONE hot path this repository implements:
 * the reference's bijector chain evaluated in the likelihood direction
 * (NLL + log|det J|) and in the sampling direction.  The reference has no native
 * boundary of its own (it is 100 % Python on the TensorFlow 1.12 runtime); the
 * calls below replace the `sess.run(...)` feeds of
 *     train_noise_flow.py:112-113   (loss, sd_z)           -> nf_nll
 *     train_noise_flow.py:167-168   (x_sample)             -> nf_sample
 *     borealisflows/NoiseFlowWrapper.py:81-87              -> nf_sample
 * and the graph construction + Saver.restore of
 *     borealisflows/noise_flow_model.py:54-235, NoiseFlowWrapper.py:46-79 -> nf_create
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types cross this boundary;
 *   - all tensor pointers passed to nf_nll / nf_sample / nf_synth_patches are
 *     DEVICE pointers (HBM) owned by the caller, fp32, NHWC, contiguous;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - every function returns 0 on success and a negative NF_E* code on failure;
 *     nf_last_error() returns a thread-local description of the last failure;
 *   - a handle is immutable after nf_create: nf_nll / nf_sample are re-entrant
 *     and may be called concurrently from many host threads (the reference's
 *     16-32 Python threads sharing one tf.Session, job_noise_flow.sh:36,
 *     train_dncnn_noiseflow.py:195); they never synchronise and — patches of
 *     up to 64x64 — never allocate; images beyond 64x64 use scratch the handle
 *     owns (nf_reserve_workspace; grown on demand otherwise).  The
 *     nf_*_batchstats variants are the documented exception.
 */
#ifndef NOISEFLOW_HIP_H
#define NOISEFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_ABI_VERSION 1

/* error codes */
#define NF_OK            0
#define NF_EINVAL       -1   /* bad argument / unsupported configuration      */
#define NF_EHIP         -2   /* a HIP runtime call failed                     */
#define NF_ECOND        -3   /* unknown camera id (cond_utils.py:216-217)     */
#define NF_ENOMEM       -4

/* layer types = the bijectors of the shipped / job-script architectures
 * (noise_flow_model.py:71-235).  An `unc` arch entry is a CONV1X1 followed by a
 * COUPLING (noise_flow_model.py:79-104). */
#define NF_LAYER_CONV1X1   1   /* layers.py:74-145  Conv2d1x1, decomp = LU, no bias        */
#define NF_LAYER_COUPLING  2   /* layers.py:251-375 AffineCoupling + real_nvp_conv_template */
#define NF_LAYER_SDN5      3   /* AffineCouplingSdnEx5.py:22-132 + cond_utils.py:205-239    */
#define NF_LAYER_GAIN4     4   /* AffineCouplingGainEx4.py:23-127 + cond_utils.py:432-440   */
/* secondary variants reachable from job_noise_flow.sh (arch "sdn4|gain4") and the
 * plain sdn / gain layers: the same elementwise kernels with other host scalars */
#define NF_LAYER_SDN4      5   /* AffineCouplingSdnEx4.py + cond_utils.py:178-202 (no camera parameters) */
#define NF_LAYER_SDN       6   /* AffineCouplingSdn.py    + cond_utils.py:41-52   scale = sqrt(sig(b1) y + sig(b2)) */
#define NF_LAYER_GAIN      7   /* AffineCouplingGain.py   + cond_utils.py:319-330 scale = sig(g1) iso + sig(g2);
                                  log|det| = -log(scale) ONCE per patch, exactly as the reference writes it
                                  (AffineCouplingGain.py:113-127 omits the H*W*C factor) */
/* the rest of noise_flow_arch's vocabulary (noise_flow_model.py:116-223): per-ISO gain tables and other
 * parameterisations of the same two elementwise forms, scale^2 = a*y + b  /  scale = a */
#define NF_LAYER_SDN1      8   /* AffineCouplingSdnEx1.py + cond_utils.py:55-98   sqrt(sig(b1) y / r_gain + sig(b2)), r_gain = exp(1e-2 rg[iso]) iso */
#define NF_LAYER_SDN2      9   /* AffineCouplingSdnEx2.py + cond_utils.py:101-138 sqrt(gain (sig(b1) y / gain + sig(b2))),  gain = exp(0.1 g[iso]) iso */
#define NF_LAYER_SDN3     10   /* AffineCouplingSdnEx3.py + cond_utils.py:141-175 gain sqrt(sig(b1) y / gain + sig(b2)) */
#define NF_LAYER_SDN6     11   /* AffineCouplingSdnEx6.py + cond_utils.py:242-276 SDN5 with ONE camera parameter (scales the gain only) */
#define NF_LAYER_GAIN1    12   /* AffineCouplingGainEx1.py + cond_utils.py:333-350 scale = exp(1e-5 g1) iso + exp(1e-5 g2); log|det| once per patch (as GAIN) */
#define NF_LAYER_GAIN2    13   /* AffineCouplingGainEx2.py + cond_utils.py:353-392 scale = exp(0.1 g[iso]) iso; log|det| = -H*W*C log(scale) */
#define NF_LAYER_GAIN3    14   /* AffineCouplingGainEx3.py + cond_utils.py:395-429 scale = exp(1e-5 g[iso]);  log|det| once per patch (as GAIN) */
/* the other settings of hps.flow_permutation / hps.decomp (noise_flow_model.py:80-92, matrix_param.py:191-193).  They fold to
 * the same 4x4 channel-mixing op as CONV1X1; flow_permutation values other than 0 / 1 simply emit no layer. */
#define NF_LAYER_CONV1X1_NONE 15 /* layers.py:74-145 Conv2d1x1, decomp = NONE (matrix_param.py:23-29): A is the variable,
                                    A^-1 and log|det A| by pivoted elimination in double */
#define NF_LAYER_CONV1X1_LU2  16 /* decomp = LU2 (matrix_param.py:143-188): full-matrix L / U variables masked to their
                                    strict triangles, float64 evaluation */
#define NF_LAYER_PERMUTE      17 /* flow_permutation = 0: tfb.Permute(channels reversed), log|det| = 0, no parameters */
/* per-ISO tables hold the entries of ISO 100, 400, 800, 1600, 3200 in that order; any other ISO uses the ISO-800
 * entry (the tf.cond chains' last branch) — unlike SDN5/SDN4/SDN6, whose empty one-hot selects 0 */

/* Raw (checkpoint-semantics, un-folded) parameter layout of one layer inside the
 * flat `params` array, starting at `param_offset` floats.  C = 4 channels.
 *
 *  CONV1X1   (36 floats)  P[4][4] row-major, sign_S[4], log_S[4], L_vec[6], U_vec[6]
 *                         (matrix_param.py:100-140; vectors in tfdist.fill_triangular order)
 *  CONV1X1_NONE (16)      A[4][4] row-major
 *  CONV1X1_LU2  (56)      P[4][4], L[4][4] (only j<i read), sign_S[4], log_S[4], U[4][4] (only j>i read)
 *  PERMUTE      (0)       —
 *  COUPLING  (width w)    l_1/W[3][3][2][w], l_1/b[w], bn1_mean[w], bn1_var[w],
 *                         l_2/W[w][w],       l_2/b[w], bn2_mean[w], bn2_var[w],
 *                         l_last/W[3][3][w+1][4], l_last/b[4], l_last/logs[4],
 *                         rescaling_scale[1]
 *                         = 27w + w*w + 36w + 36 + 4 + 4 + 1 ... see nf_layer_param_count()
 *  SDN5      (23 floats)  beta1, beta2, gain_params[5], cam_params[3][5], c_i
 *  GAIN4     (1 float)    gain_val
 *  SDN4      (7 floats)   beta1, beta2, gain_params[5]
 *  SDN       (2 floats)   b1, b2
 *  GAIN      (2 floats)   g1, g2
 *  SDN1      (7 floats)   b1, b2, r_gain_param[5]
 *  SDN2/SDN3 (7 floats)   b1, b2, gain_param[5]
 *  SDN6      (13 floats)  beta1, beta2, gain_params[5], cam_params[1][5], c_i
 *  GAIN1     (2 floats)   g1, g2
 *  GAIN2/3   (5 floats)   gain_param[5]
 */
typedef struct nf_layer_desc {
    int32_t type;          /* NF_LAYER_*                                  */
    int32_t width;         /* coupling CNN width (COUPLING only, else 0)  */
    int64_t param_offset;  /* offset in floats into `params`              */
} nf_layer_desc;

/* Patch sizes.  Up to 64x64 a patch is held whole by one workgroup (every width and mode).  Beyond that — the reference
 * leaves --patch_height free (sidd/ArgParser.py:72-73) — nf_nll / nf_sample and their host-fed variants evaluate the image
 * as overlapping 64-pixel tiles: a coupling reads a 5x5 neighbourhood, so a tile is exact 2 x (number of couplings) pixels
 * inside every tile border that is not an image border, and only that core is reported (nf_tile_plan, nf_tile_segments
 * below).  Every coupling width, fp32 or NF_CFG_FP16_CNN (width 4 in fp32 on the fused width-4 kernels, width 16 in fp32 on its
 * own, the other widths up to 32 on the width-32 matrix-core kernel, zero-padded, widths 33 .. 512 on the GEMM kernels); nf_*_batchstats beyond 64x64: coupling width 4 (every launch of its
 * schedule tiled with a halo of 3, statistics over the core windows). */
#define NF_MAX_IMAGE_SIDE 4096

typedef struct nf_config {
    int32_t height;        /* patch H (1 .. NF_MAX_IMAGE_SIDE)            */
    int32_t width;         /* patch W (1 .. NF_MAX_IMAGE_SIDE)            */
    int32_t channels;      /* must be 4 (packed Bayer raw)                */
    int32_t n_layers;      /* number of nf_layer_desc entries, NLL order  */
    int32_t device;        /* HIP device ordinal, -1 = current device     */
    int32_t flags;         /* NF_CFG_* bits (0 = everything in fp32)       */
} nf_config;

/* nf_config.flags */
#define NF_CFG_FP16_CNN 1   /* coupling CNN convs in fp16 (fp32 accumulate) on the matrix cores;
                              1x1 mixes, tanh/exp, log-det and prior stay fp32.  Width 4: its own kernel
                              for full 32x32 / 64x64 patches (BASELINE configs[4]), any other shape on
                              the width-32 kernel, zero-padded; widths 8 / 16 / 32: any patch up to 64x64
                              (v_mfma_f32_32x32x16_f16); widths 33 .. 512: any patch up to 64x64. */

/* Per-call conditioning: ONE value per call, not per patch — the reference
 * feeds length-1 lists (MiniBatchSampler.py:61-64, NoiseFlowWrapper.py:85-86).
 * cam in {0..4} = IP, GP, S6, N6, G4; an unknown ISO silently selects gain
 * parameter 0 (cond_utils.py:227-229); an unknown camera is an error. */
typedef struct nf_cond {
    float iso;
    float cam;
    float nlf0;            /* carried for interface parity; unused by sdn5 */
    float nlf1;
} nf_cond;

typedef struct nf_handle nf_handle;

/* flags for nf_nll */
#define NF_ACCUMULATE   1u   /* add into sums_out instead of overwriting it          */
#define NF_NO_PRIOR     2u   /* nll_out receives -sum(log-dets) only (no base logp)   */
#define NF_SUMS_WIDE    4u   /* sums_out is the slotted layout below, not double[3]   */

/* Slotted sums (NF_SUMS_WIDE): sums_out = double[NF_SUMS_SLOTS * NF_SUMS_STRIDE] on the device;
 * slot s (at sums_out + s*NF_SUMS_STRIDE) holds (sum nll, sum sd, count) of the workgroups with
 * index = s mod NF_SUMS_SLOTS.  Every workgroup adds its two sums atomically; with the plain
 * double[3] all of those atomics hit one cache line and serialise (~10 ns each, 5.6 us of a 55 us
 * launch at B = 1024), the slots are 128 B apart.  Accumulate over as many calls as wanted, then
 * fold once with nf_sums_reduce (or add the slots up yourself). */
#define NF_SUMS_SLOTS   64
#define NF_SUMS_STRIDE  16

int         nf_abi_version(void);
const char *nf_last_error(void);

/* Number of raw floats a layer of this type/width occupies in `params`; <0 on error. */
int64_t     nf_layer_param_count(int32_t type, int32_t width);

/* Build a model: fold the raw parameters (PLU -> A and A^-1, BN-eval and
 * exp(3*logs) folded into the conv weights, edge-indicator channel reduced to a
 * 16-entry border table, gain folded into the neighbouring 1x1 matrix) and upload
 * the two device-resident programs (NLL order / sampling order). */
int nf_create(const nf_config *cfg, const nf_layer_desc *layers,
              const float *params, size_t n_params, nf_handle **out);
int nf_destroy(nf_handle *h);

/* Likelihood direction.  Replaces NoiseFlow._loss / .inverse
 * (noise_flow_model.py:394-428, 458-484).
 *   x, y      [B,H,W,4] noise / clean image (y may be NULL if the model has no SDN5 layer)
 *   nll_out   [B]  per-patch NLL (or NULL)
 *   sd_out    [B]  per-patch sqrt(var_hwc z) (or NULL)
 *   logdet_out[B]  per-patch sum of log|det J| over all layers (or NULL)
 *   z_out     [B,H,W,4] latent (or NULL)
 *   sums_out  double[3] on the DEVICE: sum_b nll, sum_b sd, B (or NULL); with NF_SUMS_WIDE the
 *             slotted layout above.  Must be ordinary (coarse-grained) hipMalloc memory: the kernels add to
 *             it with hardware fp64 atomics (the library is built with -munsafe-fp-atomics), which fine-grained /
 *             host-coherent allocations do not support.
 * Every tensor pointer (x, y, z_out; y, eps, x_out of nf_sample) must be 16-byte aligned — a pixel is one float4
 * access — and belong to the handle's device; misaligned pointers are rejected with NF_EINVAL.
 */
int nf_nll(nf_handle *h, const float *x, const float *y, int64_t B, const nf_cond *cond,
           float *nll_out, float *sd_out, float *logdet_out, float *z_out,
           double *sums_out, uint32_t flags, void *stream);

/* out3[k] (+)= sum over the slots of a NF_SUMS_WIDE buffer (flags: NF_ACCUMULATE adds into out3).
 * Both pointers are DEVICE memory; one tiny kernel on `stream`. */
int nf_sums_reduce(const double *wide, double *out3, uint32_t flags, void *stream);

/* Sampling direction.  Replaces NoiseFlow.sample / .forward
 * (noise_flow_model.py:430-456): z = eps*temp, then the bijectors in reverse.
 *   eps       [B,H,W,4] caller-supplied N(0,1) draw, or NULL to generate it
 *             in-kernel with Philox4x32-10 keyed (seed, patch_index_base + b, pixel)
 *   x_out     [B,H,W,4] synthesised noise (unclipped, as the reference)
 */
int nf_sample(nf_handle *h, const float *y, const float *eps, uint64_t seed,
              int64_t patch_index_base, float temp, int64_t B, const nf_cond *cond,
              float *x_out, void *stream);

/* The N(0,1) draw nf_sample makes in-kernel when eps = NULL, written to eps_out [B,H,W,4] (device): Philox4x32-10 keyed
 * (seed, patch_index_base + b, pixel), bit-identical to the in-kernel values.  For callers that post-process the draw before
 * the flow — e.g. one temperature PER PATCH, which the reference's prior supports (eps_std reshaped to [-1,1,1,1],
 * noise_flow_model.py:499-504) — and hand it back as `eps`. */
int nf_sample_eps(uint64_t seed, int64_t patch_index_base, int64_t B, int32_t height, int32_t width,
                  float *eps_out, void *stream);

/* The tile plan of one image axis (see "Patch sizes"): `size` pixels, tiles of `tile` = min(size, 64) pixels, `halo` =
 * 2 x the number of coupling layers.  Returns the number of tiles n (or a negative NF_E* code) and, for i < min(n, cap),
 * the tile's first pixel origin[i] and the window [core0[i], core1[i]) it reports — a partition of [0, size) with
 * core0[i] >= origin[i] + halo and core1[i] <= origin[i] + tile - halo wherever the tile border is not the image border.
 * Pure host arithmetic (no device needed); the kernels use the same formulas (csrc/nf_device.h). */
int nf_tile_plan(int32_t size, int32_t tile, int32_t halo, int32_t *origin, int32_t *core0, int32_t *core1, int32_t cap);

/* How a model on patches beyond 64x64 is cut into tiled launches: deep stacks leave a small core per tile (8 couplings: 32 of
 * 64 pixels per axis), so the program is split after a coupling into segments, each its own tiled launch with a halo of
 * 2 x ITS couplings, the tensor between two segments resident in HBM; the number of segments minimises tiles x (couplings +
 * per-tile overhead).  Returns the number of segments (0: the patch is held whole; negative: NF_E*), and for i < cap
 * out5[5 i ..] = first op, one-past-last op (indices into nf_fold_params' op list), halo, tiles along H, tiles along W.
 * Host arithmetic only.  NF_TILE_SEGMENTS=<n> in the environment forces the count (A/B aid). */
int nf_tile_segments(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params,
                     int32_t direction, int32_t *out5, int32_t cap);

/* Host-fed variants: the call pattern of the reference's drivers — `sess.run(..., feed_dict={x: numpy, y: numpy})` with the
 * float64 minibatches of sidd/MiniBatchSampler.py:54-55 (train_noise_flow.py:112-113) and
 * NoiseFlowWrapper.sample_noise_nf(batch_x, ...) (NoiseFlowWrapper.py:81-87): every tensor pointer is HOST memory (pageable
 * or pinned, no alignment requirement), results are in the caller's host buffers when the call returns.
 *   x, y / y     [B,H,W,4] float32 (NF_HOST_F32) or float64 (NF_HOST_F64; narrowed to float32 exactly as the feed does)
 *   eps          [B,H,W,4] float32 or NULL (in-kernel Philox keyed by (seed, patch_index_base + b, pixel))
 *   nll_out, sd_out, logdet_out [B], z_out / x_out [B,H,W,4]: float32 host buffers (each optional except x_out)
 *   sums_out     HOST double[3] = (sum nll, sum sd, B), or NULL; NF_ACCUMULATE adds to it.  flags: NF_NO_PRIOR, NF_ACCUMULATE.
 * The call is a chunked pipeline (created on first use, freed by nf_destroy): a chunk is narrowed / copied into pinned staging
 * while the previous chunks cross PCIe and run on three internal streams; patches are independent, so every per-patch output
 * is bit-identical to nf_nll / nf_sample on the same data.  Evaluation mode only (running BN statistics).
 * Re-entrant like nf_nll: any number of host threads may call on one handle at once (the reference's 16 queue workers each
 * call sess.run on their own, train_noise_flow.py:30-47, job_noise_flow.sh:36) — up to 4 calls are in flight at a time
 * (NF_HOSTFED_PIPES), each on a pipeline of its own, so one caller's narrowing overlaps another's DMA and kernel, and further
 * callers wait their turn; a lone caller's narrowing is spread over a process-wide worker pool instead.  A tensor result is stored by the kernel straight into the caller's buffer when that is page-locked over its
 * whole length and 16-byte aligned, through pinned staging otherwise.  Not fork-safe while a call is in flight; a forked child
 * starts with a fresh worker pool.  Environment: NF_HOSTFED_THREADS (default: the CPUs this process may use, at most 32),
 * NF_HOSTFED_CHUNK (patches per chunk; default 8 MiB of one tensor). */
#define NF_HOST_F32 0
#define NF_HOST_F64 1
int nf_nll_host(nf_handle *h, const void *x, const void *y, int32_t dtype, int64_t B, const nf_cond *cond,
                float *nll_out, float *sd_out, float *logdet_out, float *z_out,
                double *sums_out, uint32_t flags);
int nf_sample_host(nf_handle *h, const void *y, int32_t y_dtype, const float *eps, uint64_t seed,
                   int64_t patch_index_base, float temp, int64_t B, const nf_cond *cond, float *x_out);

/* Batch-statistics variants = the reference's `is_training=True` graphs (layers.py:386-398; what
 * NoiseFlowWrapper.py:49 builds): every batch_norm of the coupling CNNs normalises with the moments
 * of the CURRENT call's B patches over (N,H,W) instead of the stored running statistics, so the
 * result of one patch depends on all patches of the call.  Same arguments as nf_nll / nf_sample, plus
 *   moments_out  HOST float [n_couplings][4][w] (or NULL): batch mean1, var1, mean2, var2 of each
 *                coupling, couplings in NLL layer order — what the reference's assign_sub EMA
 *                (layers.py:392-393, decay 0.1) consumes; applying it is the caller's business.
 * Unlike nf_nll / nf_sample these calls run 2 statistics passes per coupling before the fused pass,
 * SYNCHRONISE `stream`, use a per-handle scratch (allocated on first use; concurrent calls on one
 * handle serialise) and are fp32 only.  B must be >= 1.  Every coupling width 1 .. 512 and every patch size the handle was
 * created for: width 4 on the matrix-core schedule of the fused kernel (images beyond 64x64 as overlapping tiles); widths up to 32
 * on the scalar-weight kernel while its two LDS tiles hold the patch (1024 pixels at width 32, ~2270 at 16, ~4090 at 8; a width
 * between two kernel widths runs zero-padded on the next one); everything else — widths beyond 32 (sidd/ArgParser.py:43 defaults to
 * 512), 64x64 at the paper's width 32 — layer by layer on the trainer's matrix-core GEMM path (csrc/nf_train_mm.h) over one
 * resident tensor, batch sums in the GEMM epilogues.  moments_out rows keep the model's own w channels on every route.
 * NF_BS_WIDE=1 in the environment sends every call down the last route (A/B aid; read per call, 0 = off). */
int nf_nll_batchstats(nf_handle *h, const float *x, const float *y, int64_t B, const nf_cond *cond,
                      float *nll_out, float *sd_out, float *logdet_out, float *z_out,
                      double *sums_out, uint32_t flags, float *moments_out, void *stream);
int nf_sample_batchstats(nf_handle *h, const float *y, const float *eps, uint64_t seed,
                         int64_t patch_index_base, float temp, int64_t B, const nf_cond *cond,
                         float *x_out, float *moments_out, void *stream);

/* Cross-rank batch statistics for nf_nll_batchstats / nf_sample_batchstats: the reference's batch_norm takes its moments over
 * the WHOLE minibatch (layers.py:386-398), which under data parallelism is the union of the ranks' shards.  With a callback
 * installed the library calls fn(user, sync_buf, count, stream) after every statistics pass — 2 per coupling on the fused
 * kernels' routes (widths up to 32), ceil(2 w / 64) per normalisation on the GEMM route (widths beyond 32 and the sizes listed
 * above: the 2 w sums of a normalisation go in chunks of at most 64 doubles, i.e. 2 x ceil(2 w / 64) per coupling) —: sync_buf
 * (DEVICE, caller-owned, >= 64 doubles) holds this rank's `count` sums, written by work already enqueued on `stream`; the callback must
 * enqueue a SUM all-reduce over the ranks so that later work on `stream` sees the totals, and return 0 (the contract of
 * nf_trainer_set_sync, declared below with nf_allreduce_fn).  Every rank must call with the same B; the moments then use
 * world_size x the local pixel count, and every rank normalises with the same, global moments — N ranks x B patches evaluate
 * exactly like one rank on the N*B concatenated patches.  fn = NULL removes the hook. */
typedef int (*nf_allreduce_fn)(void *user, double *buf, int64_t count, void *stream);
int nf_set_sync(nf_handle *h, nf_allreduce_fn fn, void *user, double *sync_buf, int32_t world_size);

/* ---- Training step ------------------------------------------------------
 * Replaces `sess.run([train_op, loss, sd_z], {..., is_training: True})` (train_noise_flow.py:64-66)
 * with `train_op = AdamOptimizer(lr, 0.9, 0.999, 1e-8).minimize(loss)` or
 * `MomentumOptimizer(lr, 0.9).minimize(loss)` (train_noise_flow.py:187-198): forward in the NLL
 * direction with batch-statistics BN, loss = mean_b nll_b, its gradient w.r.t. every trainable
 * variable, the BN running-statistics EMA (layers.py:392-393) and the optimizer update.  The
 * trainer owns a device-resident copy of the RAW parameters (layout of nf_create), the optimizer
 * slots and an activation workspace sized for `max_batch` patches; a step only enqueues kernels on
 * `stream` — and on an internal side stream forked from and joined back into it with events — with no
 * allocation and no host synchronisation.  One stream at a time per trainer.
 * Layers: every NF_LAYER_* above (COUPLING at any width 1..512: 4/8/16/32 on stage kernels of their own, every other width on the
 * matrix-core GEMMs of csrc/nf_train_mm.h — no library GEMM, nothing loaded at run time) — the whole
 * vocabulary of noise_flow_arch under every
 * setting of hps.flow_permutation / hps.decomp; fp32 (nf_config.flags must be 0).
 * Trainable = everything except P / sign_S of CONV1X1 / CONV1X1_LU2, the BN statistics and c_i of SDN5 / SDN6.
 * Kernel selection (read from the environment by nf_trainer_create; the defaults are the fast paths, the others exist for
 * A/B tests and profiling — every combination computes the same step up to fp32 summation order):
 *   NF_TRAIN_TILED      bit 0 / 1: per-patch tiled backward / forward stages (widths 4 and 8, patches <= 1024 pixels,
 *                       <= 384 patches); 0 = one kernel per layer stage.  Default 3.
 *   NF_TRAIN_WIDE_MFMA  widths 16 and 32: bit 0 filter gradients, 1 l_2 forward, 2 l_2 backward, 4 l_last forward, 5 l_last transposed,
 *                       6 l_1 transposed (8: l_1 forward) on v_mfma_f32_32x32x2_f32; 3 one-pass statistics finalisers (widths >= 16);
 *                       7 filter gradients inside the stage kernels (>= 400k pixels per step), 8 l_1 forward,
 *                       11 affine / tanh backward inside the transposed l_last kernel.  Default 4095.
 *   NF_TRAIN_PR         width 32 on 32x32 patches: the patch-resident stages of csrc/nf_train_pr.h (the coupling CNN recomputed from z
 *                       between the batch-statistics barriers; no [pixel][32] tensor is allocated).  1 (default): 8 wavefronts per
 *                       patch, 2: 4 wavefronts, 0: the stage kernels NF_TRAIN_WIDE_MFMA selects; + 4: the first stage of the coupling
 *                       above in a launch of its own, + 8: likewise the first backward stage of the coupling below (by default they
 *                       ride in the neighbouring coupling's last launch), + 16: d l_last/W inside its stage at every minibatch size
 *                       (by default a launch of its own on the side stream while a quarter .. three quarters of the CUs hold a patch;
 *                       up to half of them, d l_1/W of the coupling above rides in the same side launch: + 32 keeps that one inside).
 *                       NF_TRAIN_PR_GRID=<n>: at most n workgroups walk the
 *                       patches (default: one per CU).
 *   NF_MM_MODE          the 128-column GEMMs of the widths beyond 64: 2 (default) fp32-accurate products on the bf16 matrix pipe,
 *                       8 wavefronts on 256 x 128; 3 the same on 128 x 128; 0 / 1 fp32 products on 128 x 128 / 256 x 128.
 *   NF_TRAIN_BAND       pixels (rows x patch width, halo included; 96..320, default 320) a band kernel keeps in LDS.
 *   NF_TRAIN_GEMM=1     widths 4/8/16/32 on the library-GEMM path of the other widths as well.
 *   NF_TRAIN_GEMM_C1=0  library-GEMM path: l_1 forward as sgemm + statistics pass instead of the fused kernel.
 *   NF_TRAIN_SERIAL=1   no side stream: every kernel on the caller's stream (kernel traces without overlap). */
typedef struct nf_trainer nf_trainer;
#define NF_OPT_ADAM     0
#define NF_OPT_MOMENTUM 1

int nf_trainer_create(const nf_config *cfg, const nf_layer_desc *layers, const float *params,
                      size_t n_params, int64_t max_batch, int32_t optimizer, nf_trainer **out);
int nf_trainer_destroy(nf_trainer *t);

/* Forward + backward of one minibatch (1 <= B <= max_batch); moves the BN running statistics.
 *   grads_out  DEVICE float[n_params] in the raw layout, zeros at non-trainable positions, or NULL
 *              to keep the gradient in the trainer (then nf_trainer_apply(t, NULL, ...) uses it).
 *              A data-parallel caller all-reduces this buffer between the two calls.
 *              A variable the reference shares between layers (its AUTO_REUSE scope 'sdn_gain': two GAIN4 layers have ONE
 *              gain_val) has one slot per layer here; the caller ties the slots by giving each the SUM of their gradients
 *              before nf_trainer_apply (noise_flow_amd/train.py does).
 *   loss_out   DEVICE float[2] = (mean_b nll_b, sd_z) or NULL  */
int nf_trainer_forward_backward(nf_trainer *t, const float *x, const float *y, int64_t B,
                                const nf_cond *cond, float *grads_out, float *loss_out, void *stream);
/* Forward half only: (loss, sd_z) of a minibatch under batch-statistics BN, running statistics moved,
 * no gradient — the `sidd_cond == 'condSDN'` branch of train_thread (train_noise_flow.py:61-63). */
int nf_trainer_forward(nf_trainer *t, const float *x, const float *y, int64_t B, const nf_cond *cond,
                       float *loss_out, void *stream);
/* Cross-rank batch normalisation (the reference's batch_norm takes its moments over the WHOLE minibatch,
 * layers.py:386-398; under data parallelism that is the union of the ranks' shards).  With a callback installed,
 * nf_trainer_forward_backward / _forward / _step call
 *     fn(user, buf, count, stream)
 * at every point where batch sums are formed (2 per coupling in the forward pass, 2 in the backward pass): `buf`
 * (= sync_buf, DEVICE, caller-owned, >= 64 doubles) holds this rank's `count` sums, written by work already enqueued
 * on `stream`; the callback must enqueue a SUM all-reduce of buf[0..count) over the ranks so that work enqueued on
 * `stream` afterwards sees the totals (RCCL on that stream, or any blocking implementation), and return 0.  Every rank
 * must call with the same batch size; moments then use world_size x the local pixel count.  The gradient all-reduce
 * between forward_backward and apply stays the caller's.  fn = NULL removes the hook. */
int nf_trainer_set_sync(nf_trainer *t, nf_allreduce_fn fn, void *user, double *sync_buf, int32_t world_size);
/* One optimizer update from `grads` (DEVICE float[n_params]; NULL = the trainer's own buffer). */
int nf_trainer_apply(nf_trainer *t, const float *grads, float lr, void *stream);
/* = nf_trainer_forward_backward(..., NULL, loss_out) + nf_trainer_apply(t, NULL, lr). */
int nf_trainer_step(nf_trainer *t, const float *x, const float *y, int64_t B, const nf_cond *cond,
                    float lr, float *loss_out, void *stream);
/* Copy the current raw parameters to / from HOST memory (synchronises `stream`). */
int nf_trainer_get_params(nf_trainer *t, float *params_out, size_t n_params, void *stream);
int nf_trainer_set_params(nf_trainer *t, const float *params, size_t n_params, void *stream);
int64_t nf_trainer_steps(const nf_trainer *t);   /* optimizer updates applied so far */

/* Counter-based synthetic SIDD-like patches, identical for any sharding:
 *   y_k ~ U[0,1)^(HxWx4),  x_k = eps * sqrt(beta1*y_k + beta2),  eps ~ N(0,1),
 * keyed by (seed, global patch index k = patch_index_base + b, pixel). */
int nf_synth_patches(uint64_t seed, int64_t patch_index_base, int64_t B,
                     int32_t height, int32_t width, float beta1, float beta2,
                     float *y_out, float *x_out, void *stream);

/* Host-only view of the folded program (no GPU needed; used by the CPU test
 * tier to check the C++ folding against the numpy oracle).
 *   direction 0 = NLL order, 1 = sampling order
 *   ops_out   int32 pairs (type, float offset) — at most ops_cap pairs
 *   folded    folded parameter block — at most folded_cap floats
 *   ld_const  sum of the constant log-dets (H*W*sum log_S, -H*W*C*log gain_val)
 */
int nf_fold_params(const nf_config *cfg, const nf_layer_desc *layers,
                   const float *params, size_t n_params, int32_t direction,
                   int32_t *ops_out, int32_t ops_cap, int32_t *n_ops,
                   float *folded, size_t folded_cap, size_t *n_folded,
                   double *ld_const);

/* Host-only: the parameter block of ONE kernel family's layout (`path` = NF_PATH_*; csrc/nf_device.h documents each), i.e.
 * what nf_create uploads for that family — so that a test can un-permute a layout and hold it to the folded model without a
 * GPU.  `layout_width` = the coupling width the layout is padded to.  Fails with NF_EINVAL when the model has no block
 * for that family (e.g. NF_PATH_GEMM at widths <= 32).  No reference counterpart (diagnostic). */
int nf_fold_layout(const nf_config *cfg, const nf_layer_desc *layers,
                   const float *params, size_t n_params, int32_t direction, int32_t path,
                   int32_t *ops_out, int32_t ops_cap, int32_t *n_ops, int32_t *layout_width,
                   float *folded, size_t folded_cap, size_t *n_folded);

/* Which kernel family nf_nll (direction 0) / nf_sample (direction 1) of this handle launch:
 *   NF_PATH_SCALAR  scalar-weight VALU kernel (any width / shape; also NF_KERNEL=valu)
 *   NF_PATH_MFMA4   width 4 on v_mfma_f32_4x4x1            NF_PATH_FP16 width 4, fp16 CNN (NF_CFG_FP16_CNN)
 *   NF_PATH_WIDE32  width 32 on v_mfma_f32_32x32x2_f32     NF_PATH_WIDE16 width 16 on v_mfma_f32_16x16x4_f32
 * or a negative status.  No reference counterpart (diagnostic; the parity tests use it to make sure the
 * kernel they mean to check is the one that ran). */
#define NF_PATH_SCALAR 0
#define NF_PATH_MFMA4 1
#define NF_PATH_FP16 2
#define NF_PATH_WIDE32 3
#define NF_PATH_WIDE16 4
#define NF_PATH_WIDE32_FP16 5   /* NF_CFG_FP16_CNN at width 8 / 16 / 32 (and width 4 off the full shapes): v_mfma_f32_32x32x16_f16 */
#define NF_PATH_GEMM 6          /* widths 33 .. 512: LDS-staged GEMM on v_mfma_f32_32x32x2_f32 (csrc/nf_gemm.hip) */
#define NF_PATH_GEMM_FP16 7     /* NF_CFG_FP16_CNN at widths 33 .. 512: the same on v_mfma_f32_32x32x16_f16 (csrc/nf_gemm16.hip) */
/* Both GEMM families have two variants, picked by nf_create from the padded width: weights resident in LDS with one pixel tile per
 * wavefront (<= 128) or bands of pixels with the weights streamed from L2 (256 / 512); the environment variables NF_GEMM=a /
 * NF_GEMM16=a (read at nf_create) force the band variant everywhere — an A/B aid, like NF_KERNEL=valu. */
int nf_kernel_path(const nf_handle *h, int32_t direction);

/* Device scratch of calls on images beyond 64x64 (evaluated as overlapping tiles: the per-tile sums and the tensors between two
 * segments of the program).  The handle owns a small set of buffers, one per call in flight; nf_nll / nf_sample allocate only
 * when a call needs more than any earlier call did or when more calls are in flight than ever before.  nf_workspace_bytes says
 * what one call of B images needs (0 for patches of up to 64x64: they need none); nf_reserve_workspace allocates that for
 * `calls_in_flight` concurrent calls up front (synchronous; at nf_create time, so to speak), after which no call of up to B
 * images allocates anything.  No reference counterpart (TF sizes its arena inside sess.run). */
int64_t nf_workspace_bytes(const nf_handle *h, int32_t direction, int64_t B);
int nf_reserve_workspace(nf_handle *h, int64_t B, int32_t calls_in_flight);

/* Host-only: the SDN5 scalars the kernels receive for a given (iso, cam):
 * out[0] = beta1/gain, out[1] = beta2 (cond_utils.py:205-239). */
int nf_sdn5_scalars(const float *sdn_params /*23 floats*/, const nf_cond *cond, double out[2]);

#ifdef __cplusplus
}
#endif
#endif /* NOISEFLOW_HIP_H */
