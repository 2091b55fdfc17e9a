// Training step of the Noise Flow stack on gfx950: forward in the NLL direction with
// batch-statistics BN, backward, BN running-statistics EMA and the optimizer update — everything
// device-resident and stream-ordered (no host synchronisation inside a step).
//
// Reference call sites replaced (paths relative to /root/reference):
//   train_noise_flow.py:64-66     sess.run([train_op, loss, sd_z], {..., is_training: True})
//   train_noise_flow.py:187-198   tf.train.AdamOptimizer(lr, 0.9, 0.999, 1e-8) / MomentumOptimizer(lr, 0.9)
//   noise_flow_model.py:394-428, 458-484   inverse + loss (the differentiated function)
//   layers.py:355-375, 463-497, 378-401    AffineCoupling, its CNN, batch_norm(training=True) + EMA
//   layers.py:117-130, matrix_param.py:100-140   Conv2d1x1 through its PLU parameters
//   AffineCouplingSdnEx5.py:118-132 + cond_utils.py:205-239, AffineCouplingGainEx4.py:114-127
//
// Structure.  Batch-statistics BN couples all patches of the minibatch at 16 points of the forward
// pass and 16 of the backward pass, so a step is a sequence of layer kernels over HBM-resident
// activations ([B,H,W,C] fp32, NHWC, one thread per pixel) rather than one fused per-patch kernel;
// at the reference's minibatch sizes (138 patches = 2.3 MB per tensor) every tensor lives in the
// L2 / Infinity Cache and a step is bound by kernel-launch latency, not by HBM.  Per-channel and
// per-parameter sums are reduced in registers -> wavefront shuffles -> fp64 atomics; the BN
// finalisers, the PLU / sdn5 chain rule and the optimizer run as tiny device kernels so the host
// only enqueues.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/noiseflow_hip.h"
#include "nf_internal.h"
#include "nf_dev_util.h"   // philox_normal4 (the evaluator draws the fused kernels' epsilon)

namespace {

constexpr int TB = 256;                 // threads per block of the pixel kernels
constexpr float kBnEps = 1e-4f;         // layers.py:378
constexpr float kBnDecay = 0.1f;        // layers.py:378
constexpr float kLogscale = 3.0f;       // layers.py:653
constexpr int kMaxLayers = 64;

struct Geo {
    int B, H, W, HW;
    int64_t npix;    // B*H*W
    int64_t nloop;   // npix rounded up to a multiple of 64: whole wavefronts iterate together
    int nslot;       // partial-sum slots the reducers of this step read (= grid.x of the pixel kernels)
};

struct TLayer {
    int type;    // NF_LAYER_* of the KERNELS that run it: every 1x1-mix kind is CONV1X1, every sdn kind SDN5, every gain kind GAIN4
    int kind;    // NF_LAYER_* as given: which parameterisation k_prep / k_finish evaluate
    int width;
    int off;     // offset of the layer's raw parameters (floats)
    int aux;     // index among the layers of the same type
};

struct TLayers {
    int n;
    TLayer l[kMaxLayers];
};

// Wavefront sums, the total in every lane.  Four DPP adds finish the 16-lane rows (xor 1 and xor 2 inside the quads, then
// the half-row and row mirrors), four v_readlane + three adds join the rows: ~11 VALU/SALU issues.  The __shfl_xor
// butterfly this replaces is six DEPENDENT ds_bpermute round trips (~60 cycles each): with 8 .. 70 sums at the end of
// every kernel of the step that was 1.5 - 3.5 us per launch (measured: 0.89 -> 0.82 ms per step at B = 138).
#define NF_DPP_I(x, CTRL) __builtin_amdgcn_update_dpp(0, (x), (CTRL), 0xf, 0xf, false)
#define NF_DPP_QX1 0xB1    // quad_perm [1,0,3,2]
#define NF_DPP_QX2 0x4E    // quad_perm [2,3,0,1]
#define NF_DPP_HMIR 0x141  // row_half_mirror
#define NF_DPP_MIR 0x140   // row_mirror
__device__ __forceinline__ float wsum(float v)
{
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_QX1));
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_QX2));
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_HMIR));
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_MIR));   // every lane of a 16-lane row now holds the row's sum
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// the first half of wsum: every lane of a 16-lane DPP row ends up with its row's sum (4 issues).  Reductions of MANY values stop
// here and let one lane per row park the row sums in LDS — the 7 readlane / add issues per value that join the four rows of a
// wavefront are then paid once, by the few threads that add the partials up, not by every wavefront for every value
__device__ __forceinline__ float row_sum16(float v)
{
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_QX1));
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_QX2));
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_HMIR));
    v += __int_as_float(NF_DPP_I(__float_as_int(v), NF_DPP_MIR));
    return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_get(double v)
{
    const int lo = NF_DPP_I(__double2loint(v), CTRL), hi = NF_DPP_I(__double2hiint(v), CTRL);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum(double v)
{
    v += dpp_get<NF_DPP_QX1>(v);
    v += dpp_get<NF_DPP_QX2>(v);
    v += dpp_get<NF_DPP_HMIR>(v);
    v += dpp_get<NF_DPP_MIR>(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k), __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// Reductions over the minibatch (BN moments, parameter gradients) avoid atomics: device-scope
// atomics on one cache line serialise at ~10 ns each on this part (measured: 8k of them made a 5 us
// kernel take 75 us).  Instead every workgroup reduces its share (registers -> wavefront shuffles ->
// LDS) and stores ONE partial per value into its own slot; a later kernel adds the slots up in
// fp64.  Deterministic for a given grid, so a training run is reproducible bit for bit.
constexpr int NSLOT = 1024;  // >= the largest grid.x of a reducing kernel

struct Acc {      // value k of this accumulator group lives at p[k*NSLOT + slot]
    float *p;
};
__host__ __device__ __forceinline__ Acc operator+(Acc a, int k) { return Acc{a.p + (size_t)k * NSLOT}; }

// every thread of the workgroup must call this (it contains barriers)
__device__ __forceinline__ void acc_add(Acc dst, float v, int nslot)
{
    __shared__ float part[TB / 64];
    const float s = wsum(v);
    __syncthreads();                       // the previous use of `part` is over
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < TB / 64; ++i) tot += part[i];
        dst.p[blockIdx.x] = tot;
        // kernels launched with fewer workgroups than `nslot` clear the slots nobody owns
        for (int k = blockIdx.x + gridDim.x; k < nslot; k += gridDim.x) dst.p[k] = 0.0f;
    }
}

// N consecutive values with ONE barrier pair (the per-value form above costs two barriers each)
template <int N>
__device__ __forceinline__ void acc_add_n(Acc dst, const float (&v)[N], int nslot)
{
    __shared__ float part[TB / 16][N];     // one partial per 16-lane row (row_sum16)
    const int row = threadIdx.x >> 4;
    __syncthreads();                       // the previous use of `part` is over
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float s = row_sum16(v[k]);
        if ((threadIdx.x & 15) == 0) part[row][k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += TB) {
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < TB / 64; ++i)   // wavefront by wavefront, its four rows joined as wsum joins them
            tot += (part[4 * i][k] + part[4 * i + 1][k]) + (part[4 * i + 2][k] + part[4 * i + 3][k]);
        float *d = dst.p + (size_t)k * NSLOT;
        d[blockIdx.x] = tot;
        for (int q = blockIdx.x + gridDim.x; q < nslot; q += gridDim.x) d[q] = 0.0f;
    }
}

// sum of the first `nslot` slots of one value, computed by one whole wavefront; the loads are
// issued together (the slots were written by other XCDs, so each one is a long-latency miss)
__device__ __forceinline__ double acc_total(Acc a, int nslot)
{
    float x[NSLOT / 64];
#pragma unroll
    for (int k = 0; k < NSLOT / 64; ++k) {
        const int i = (threadIdx.x & 63) + 64 * k;
        x[k] = i < nslot ? a.p[i] : 0.0f;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT / 64; ++k) s += (double)x[k];
    return wsum(s);
}

// two values at once: all 2*NSLOT/64 loads are in flight together (each is a remote-L2 / memory miss)
__device__ __forceinline__ void acc_total2(Acc a, Acc b, int nslot, double &sa, double &sb)
{
    float x[NSLOT / 64], y[NSLOT / 64];
#pragma unroll
    for (int k = 0; k < NSLOT / 64; ++k) {
        const int i = (threadIdx.x & 63) + 64 * k;
        x[k] = i < nslot ? a.p[i] : 0.0f;
        y[k] = i < nslot ? b.p[i] : 0.0f;
    }
    sa = 0.0;
    sb = 0.0;
#pragma unroll
    for (int k = 0; k < NSLOT / 64; ++k) {
        sa += (double)x[k];
        sb += (double)y[k];
    }
    sa = wsum(sa);
    sb = wsum(sb);
}

// Cross-rank batch normalisation (nf_trainer_set_sync): between the kernel that produces a group of slotted sums
// and the one that consumes it, the per-rank totals go through the caller's all-reduce.  compact: one wavefront per
// value adds its slots in fp64; scatter: the global total comes back as slot 0 + slot 1 (hi + lo floats, so that the
// consumers' fp64 slot sum reconstructs it) with every other slot cleared.
__global__ __launch_bounds__(64) void k_slots_compact(Acc a, int nslot, double *__restrict__ out)
{
    const double s = acc_total(a + (int)blockIdx.x, nslot);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void k_slots_scatter(Acc a, int nslot, const double *__restrict__ in)
{
    float *d = (a + (int)blockIdx.x).p;
    const double tot = in[blockIdx.x];
    const float hi = (float)tot, lo = (float)(tot - (double)hi);
    for (int k = threadIdx.x; k < nslot; k += 64) d[k] = k == 0 ? hi : k == 1 ? lo : 0.0f;
}

// per-patch accumulator: wavefronts that sit inside one patch reduce first
__device__ __forceinline__ void patch_add(float *arr, int b, bool valid, float v)
{
    const int b0 = __shfl(b, 0);
    if (__all(!valid || b == b0)) {
        const float s = wsum(valid ? v : 0.0f);
        if ((threadIdx.x & 63) == 0 && b0 >= 0) atomicAdd(&arr[b0], s);
    } else if (valid) {
        atomicAdd(&arr[b], v);
    }
}

#define NF_PIXEL_LOOP(g, p) \
    for (int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x; p < (g).nloop; p += (int64_t)gridDim.x * TB)

// ---------------------------------------------------------------------------------------------
// per-step scalar preparation: A = P L U, sdn5 (a, b), constant log-det
// ---------------------------------------------------------------------------------------------
// vector element -> matrix entry of tfdist.fill_triangular padded to a strict triangle
// (matrix_param.py:31-56); the same table as fold_conv1x1 in nf_host.hip
__device__ const int kLr[6] = {3, 3, 3, 1, 2, 2}, kLc[6] = {2, 1, 0, 0, 1, 0};
__device__ const int kUr[6] = {0, 0, 0, 2, 1, 1}, kUc[6] = {1, 2, 3, 3, 2, 3};

__device__ void plu_matrices(const float *p, double Pm[4][4], double L[4][4], double U[4][4])
{
    const float *sign_s = p + 16, *log_s = p + 20, *lv = p + 24, *uv = p + 30;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            Pm[i][j] = p[i * 4 + j];
            L[i][j] = i == j ? 1.0 : 0.0;
            U[i][j] = 0.0;
        }
    for (int k = 0; k < 6; ++k) {
        L[kLr[k]][kLc[k]] = lv[k];
        U[kUr[k]][kUc[k]] = uv[k];
    }
    for (int i = 0; i < 4; ++i) U[i][i] = (double)sign_s[i] * exp((double)log_s[i]);
}

struct CondIdx {
    float iso;
    int iso_idx;   // index into gain_params or -1 (unknown ISO -> 0, cond_utils.py:227-229)
    int cam_idx;   // 0..4
};

__device__ void sdn5_eval(const float *sp, CondIdx ci, double &a, double &b)
{
    const double c_i = sp[22];
    double cp[3];
    for (int r = 0; r < 3; ++r) cp[r] = exp(c_i * (double)sp[7 + r * 5 + ci.cam_idx]);
    const double g = ci.iso_idx >= 0 ? (double)sp[2 + ci.iso_idx] : 0.0;
    const double gain = exp(c_i * g * cp[2]) * (double)ci.iso;
    a = exp(c_i * (double)sp[0] * cp[0]) / gain;
    b = exp(c_i * (double)sp[1] * cp[1]);
}

// 4x4 inverse and log|det| by Gauss-Jordan with partial pivoting (decomp = NONE: tf.matrix_inverse / tf.linalg.slogdet)
__device__ void inv4(const double M[4][4], double inv[4][4], double &lad)
{
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = M[i][j];
            a[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    lad = 0.0;
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        for (int j = 0; j < 8; ++j) {
            const double t = a[piv][j];
            a[piv][j] = a[c][j];
            a[c][j] = t;
        }
        const double d = a[c][c];
        lad += log(fabs(d));
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i][j] = a[i][4 + j];
}

// P, L, U of decomp = LU2 (matrix_param.py:143-188): full-matrix variables masked to their strict triangles
__device__ void lu2_matrices(const float *p, double Pm[4][4], double L[4][4], double U[4][4])
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            Pm[i][j] = p[i * 4 + j];
            L[i][j] = j < i ? (double)p[16 + i * 4 + j] : (i == j ? 1.0 : 0.0);
            U[i][j] = j > i ? (double)p[40 + i * 4 + j] : (i == j ? (double)p[32 + i] * exp((double)p[36 + i]) : 0.0);
        }
}

__device__ __forceinline__ double sigm(double v) { return 1.0 / (1.0 + exp(-v)); }

// the per-ISO tables of the Ex1-Ex3 layers: any ISO outside 100..3200 takes the ISO-800 entry (cond_utils.py:69-88)
__device__ __forceinline__ int table_idx(CondIdx ci) { return ci.iso_idx >= 0 ? ci.iso_idx : 2; }

// scale^2 = a y + b of every sdn kind (cond_utils.py:41-276)
__device__ void sdn_ab(int kind, const float *p, CondIdx ci, double &a, double &b)
{
    const double iso = ci.iso;
    if (kind == NF_LAYER_SDN5) {
        sdn5_eval(p, ci, a, b);
    } else if (kind == NF_LAYER_SDN4) {                                    // c = 1, no camera
        const double gpar = ci.iso_idx >= 0 ? (double)p[2 + ci.iso_idx] : 0.0;
        a = exp((double)p[0]) / (exp(gpar) * iso);
        b = exp((double)p[1]);
    } else if (kind == NF_LAYER_SDN) {
        a = sigm(p[0]);
        b = sigm(p[1]);
    } else if (kind == NF_LAYER_SDN6) {                                    // one camera parameter, on the gain exponent
        const double c = p[12], cp = exp(c * (double)p[7 + ci.cam_idx]);
        const double g = ci.iso_idx >= 0 ? (double)p[2 + ci.iso_idx] : 0.0;
        a = exp(c * (double)p[0]) / (exp(c * g * cp) * iso);
        b = exp(c * (double)p[1]);
    } else {                                                               // SDN1 / SDN2 / SDN3
        const double gain = exp((kind == NF_LAYER_SDN1 ? 1e-2 : 1e-1) * (double)p[2 + table_idx(ci)]) * iso;
        const double A0 = sigm(p[0]), B0 = sigm(p[1]);
        if (kind == NF_LAYER_SDN1) { a = A0 / gain; b = B0; }
        else if (kind == NF_LAYER_SDN2) { a = A0; b = gain * B0; }
        else { a = gain * A0; b = gain * gain * B0; }
    }
}

// scale of every gain kind and the number of log(scale) terms its log-det holds (cond_utils.py:319-440 and the layers:
// GainEx4 / GainEx2 write the full H*W*C sum, Gain / GainEx1 / GainEx3 -log(scale) once per patch)
__device__ void gain_s(int kind, const float *p, CondIdx ci, int HW, double &sv, double &K)
{
    const double iso = ci.iso;
    K = 1.0;
    if (kind == NF_LAYER_GAIN4) { sv = p[0]; K = 4.0 * HW; }
    else if (kind == NF_LAYER_GAIN) sv = sigm(p[0]) * iso + sigm(p[1]);
    else if (kind == NF_LAYER_GAIN1) sv = exp(1e-5 * (double)p[0]) * iso + exp(1e-5 * (double)p[1]);
    else if (kind == NF_LAYER_GAIN2) { sv = exp(1e-1 * (double)p[table_idx(ci)]) * iso; K = 4.0 * HW; }
    else sv = exp(1e-5 * (double)p[table_idx(ci)]);
}

__device__ void k_prep_layer(const TLayer L, const float *__restrict__ P, CondIdx ci, int HW, float *__restrict__ Abuf,
                             float *__restrict__ abbuf, float *__restrict__ sbuf, double *ldc)
{
    const float *p = P + L.off;
    if (L.kind == NF_LAYER_PERMUTE) {                                      // tfb.Permute(channels reversed), log|det| = 0
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Abuf[L.aux * 16 + i * 4 + j] = (i + j == 3) ? 1.0f : 0.0f;
    } else if (L.kind == NF_LAYER_CONV1X1_NONE) {
        double M[4][4], Mi[4][4], lad;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                M[i][j] = p[i * 4 + j];
                Abuf[L.aux * 16 + i * 4 + j] = p[i * 4 + j];
            }
        inv4(M, Mi, lad);
        *ldc += (double)HW * lad;
    } else if (L.kind == NF_LAYER_CONV1X1_LU2) {
        double Pm[4][4], Lm[4][4], Um[4][4];
        lu2_matrices(p, Pm, Lm, Um);
        double lad = 0.0;
        for (int i = 0; i < 4; ++i) {
            lad += (double)p[36 + i];
            for (int j = 0; j < 4; ++j) {
                double sacc = 0.0;
                for (int k = 0; k < 4; ++k)
                    for (int m = 0; m < 4; ++m) sacc += Pm[i][k] * Lm[k][m] * Um[m][j];
                Abuf[L.aux * 16 + i * 4 + j] = (float)sacc;
            }
        }
        *ldc += (double)HW * lad;
    } else if (L.type == NF_LAYER_CONV1X1) {
        double Pm[4][4], Lm[4][4], Um[4][4], LU[4][4];
        plu_matrices(p, Pm, Lm, Um);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Lm[i][k] * Um[k][j];
                LU[i][j] = s;
            }
        double lad = 0.0;
        for (int i = 0; i < 4; ++i) {
            lad += (double)p[20 + i];
            for (int j = 0; j < 4; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Pm[i][k] * LU[k][j];
                Abuf[L.aux * 16 + i * 4 + j] = (float)s;
            }
        }
        *ldc += (double)HW * lad;                                  // layers.py:129-130
    } else if (L.type == NF_LAYER_SDN5) {
        double a, b;
        sdn_ab(L.kind, p, ci, a, b);
        abbuf[L.aux * 2] = (float)a;
        abbuf[L.aux * 2 + 1] = (float)b;
    } else if (L.type == NF_LAYER_GAIN4) {
        double sv, K;
        gain_s(L.kind, p, ci, HW, sv, K);
        sbuf[L.aux] = (float)sv;
        *ldc += -K * log(sv);                                      // AffineCouplingGainEx4.py:114-127 and siblings
    }
}

// One workgroup of 64 threads, one layer per thread.  Also clears the per-patch accumulators of k_prior and adds the
// layers' constant log-det terms up serially (no memset launches in front of the step, and a deterministic sum).
__global__ __launch_bounds__(64) void k_prep(TLayers ls, const float *__restrict__ P, CondIdx ci, int HW, float *__restrict__ Abuf,
                                             float *__restrict__ abbuf, float *__restrict__ sbuf, double *__restrict__ ldc,
                                             float *__restrict__ patch_acc, int n_patch_acc)
{
    __shared__ double ld_part[kMaxLayers];
    for (int i = threadIdx.x; i < n_patch_acc; i += 64) patch_acc[i] = 0.0f;
    const int l = threadIdx.x;
    ld_part[l] = 0.0;
    if (l < ls.n) k_prep_layer(ls.l[l], P, ci, HW, Abuf, abbuf, sbuf, &ld_part[l]);
    __syncthreads();
    if (l == 0) {
        double tot = 0.0;
        for (int i = 0; i < ls.n; ++i) tot += ld_part[i];
        *ldc = tot;
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
__global__ void k_sdn_fwd(Geo g, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ ab,
                          float *__restrict__ z, Acc ldacc)
{
    // the loss needs only the batch total of this layer's log-det: one slot-reduced value, no atomics
    const float a = ab[0], b = ab[1];
    float l = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 xv = reinterpret_cast<const float4 *>(x)[p], yv = reinterpret_cast<const float4 *>(y)[p];
            const float s0 = sqrtf(fmaf(a, yv.x, b)), s1 = sqrtf(fmaf(a, yv.y, b)), s2 = sqrtf(fmaf(a, yv.z, b)),
                        s3 = sqrtf(fmaf(a, yv.w, b));
            reinterpret_cast<float4 *>(z)[p] = make_float4(xv.x / s0, xv.y / s1, xv.z / s2, xv.w / s3);
            l -= logf(s0) + logf(s1) + logf(s2) + logf(s3);
        }
    }
    const float lv[1] = {l};
    acc_add_n<1>(ldacc, lv, g.nslot);
}

__global__ void k_scale_fwd(Geo g, const float *__restrict__ zin, const float *__restrict__ gain, float *__restrict__ zout)
{
    const float inv = 1.0f / gain[0];
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 v = reinterpret_cast<const float4 *>(zin)[p];
            reinterpret_cast<float4 *>(zout)[p] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
        }
    }
}

__global__ void k_mix_fwd(Geo g, const float *__restrict__ zin, const float *__restrict__ A, float *__restrict__ zout)
{
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = A[i];
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 v = reinterpret_cast<const float4 *>(zin)[p];
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = v.x * m[j] + v.y * m[4 + j] + v.z * m[8 + j] + v.w * m[12 + j];
            reinterpret_cast<float4 *>(zout)[p] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// l_1: 3x3 SAME conv of the pass-through half + bias; per-channel sum / sum of squares.
// MIX = the preceding Conv2d1x1 is folded in: `zin` is then the tensor BEFORE the 1x1 mix, the two
// pass-through channels of every neighbour are recomputed on the fly (8 FMAs) and the mixed pixel is
// stored to `zmixed` for the rest of the step — one launch and one pass over z less per `unc`.
template <int W, bool MIX>
__global__ void k_c1_fwd(Geo g, const float *__restrict__ zin, const float *__restrict__ A, float *__restrict__ zmixed,
                         const float *__restrict__ P, int off, float *__restrict__ h1, Acc stats)
{
    const float *W1 = P + off, *b1 = W1 + 18 * W;
    float m[16];
    if (MIX) {
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = A[i];
    }
    float s[W], q[W];
#pragma unroll
    for (int j = 0; j < W; ++j) s[j] = q[j] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            float h[W];
#pragma unroll
            for (int j = 0; j < W; ++j) h[j] = b1[j];
            for (int di = 0; di < 3; ++di) {
                const int rr = r + di - 1;
                if (rr < 0 || rr >= g.H) continue;
                for (int dj = 0; dj < 3; ++dj) {
                    const int cc = c + dj - 1;
                    if (cc < 0 || cc >= g.W) continue;
                    const int64_t qi = (int64_t)b * g.HW + rr * g.W + cc;
                    float2 v;
                    if (MIX) {
                        const float4 u = reinterpret_cast<const float4 *>(zin)[qi];
                        v.x = u.x * m[0] + u.y * m[4] + u.z * m[8] + u.w * m[12];
                        v.y = u.x * m[1] + u.y * m[5] + u.z * m[9] + u.w * m[13];
                        if (di == 1 && dj == 1)
                            reinterpret_cast<float4 *>(zmixed)[p] =
                                make_float4(v.x, v.y, u.x * m[2] + u.y * m[6] + u.z * m[10] + u.w * m[14],
                                            u.x * m[3] + u.y * m[7] + u.z * m[11] + u.w * m[15]);
                    } else {
                        v = *reinterpret_cast<const float2 *>(zin + qi * 4);
                    }
                    const float *w = W1 + (di * 3 + dj) * 2 * W;
#pragma unroll
                    for (int j = 0; j < W; ++j) h[j] = fmaf(v.x, w[j], fmaf(v.y, w[W + j], h[j]));
                }
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                h1[p * W + j] = h[j];
                s[j] += h[j];
                q[j] = fmaf(h[j], h[j], q[j]);
            }
        }
    }
    float sq[2 * W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        sq[j] = s[j];
        sq[W + j] = q[j];
    }
    acc_add_n<2 * W>(stats, sq, g.nslot);
}

// BN finalisers, fused into the kernel that consumes them: EVERY workgroup adds up the slot sums
// (one wavefront per channel, fp64) and keeps (mean, 1/sqrt(var+eps)) in LDS; workgroup 0 also
// publishes them for the backward pass and moves the running statistics (layers.py:388-393).
// fin: a finaliser kernel (k_bn_fin, wide couplings) already did all of this; the moments are read from bn_out.
template <int W>
__device__ __forceinline__ void bn_from_slots(Acc stats, int nslot, double n, float *sh, float *__restrict__ P,
                                              int off_mean, int off_var, float *__restrict__ bn_out, bool fin = false)
{
    if (fin) {
        if (threadIdx.x < 2 * W) sh[threadIdx.x] = bn_out[threadIdx.x];
        __syncthreads();
        return;
    }
    for (int j = threadIdx.x >> 6; j < W; j += TB / 64) {
        double sm, sq;
        acc_total2(stats + j, stats + W + j, nslot, sm, sq);
        const double m = sm / n;
        double v = sq / n - m * m;
        if (v < 0.0) v = 0.0;
        if ((threadIdx.x & 63) == 0) {
            const float mf = (float)m, rf = (float)(1.0 / sqrt(v + (double)kBnEps));
            sh[j] = mf;
            sh[W + j] = rf;
            if (blockIdx.x == 0 && blockIdx.y == 0) {
                bn_out[j] = mf;
                bn_out[W + j] = rf;
                P[off_mean + j] -= kBnDecay * (P[off_mean + j] - mf);
                P[off_var + j] -= kBnDecay * (P[off_var + j] - (float)v);
            }
        }
    }
    __syncthreads();
}

// the two batch means of the BN backward formula, same scheme
template <int W>
__device__ __forceinline__ void bnb_from_slots(Acc bstats, int nslot, double n, float *sh, const float *__restrict__ pre = nullptr)
{
    if (pre) {   // k_bnb_fin already added the slots up
        if (threadIdx.x < 2 * W) sh[threadIdx.x] = pre[threadIdx.x];
        __syncthreads();
        return;
    }
    for (int j = threadIdx.x >> 6; j < W; j += TB / 64) {
        double a, b;
        acc_total2(bstats + j, bstats + W + j, nslot, a, b);
        if ((threadIdx.x & 63) == 0) {
            sh[j] = (float)(a / n);
            sh[W + j] = (float)(b / n);
        }
    }
    __syncthreads();
}

// Wide couplings: with W channels and up to 1 024 slots, every workgroup adding up the slots itself reads 8 W KB from L2 —
// 256 KB per workgroup at width 32, as much in total as the activation tensor (measured: 120 of the 240 us of a stage
// kernel).  There a one-wavefront-per-channel kernel does it once; the extra launch is noise next to 100-us stages.
__global__ __launch_bounds__(64) void k_bn_fin(Acc stats, int W, int nslot, double n, float *__restrict__ P, int off_mean, int off_var,
                                               float *__restrict__ bn_out)
{
    const int j = blockIdx.x;
    double sm, sq;
    acc_total2(stats + j, stats + W + j, nslot, sm, sq);
    const double m = sm / n;
    double v = sq / n - m * m;
    if (v < 0.0) v = 0.0;
    if (threadIdx.x == 0) {
        const float mf = (float)m;
        bn_out[j] = mf;
        bn_out[W + j] = (float)(1.0 / sqrt(v + (double)kBnEps));
        P[off_mean + j] -= kBnDecay * (P[off_mean + j] - mf);
        P[off_var + j] -= kBnDecay * (P[off_var + j] - (float)v);
    }
}

__global__ __launch_bounds__(64) void k_bnb_fin(Acc bstats, int W, int nslot, double n, float *__restrict__ out)
{
    const int j = blockIdx.x;
    double a, b;
    acc_total2(bstats + j, bstats + W + j, nslot, a, b);
    if (threadIdx.x == 0) {
        out[j] = (float)(a / n);
        out[W + j] = (float)(b / n);
    }
}

// BN1 + ReLU + l_2 (1x1) + bias; statistics of the result
template <int W>
__global__ void k_c2_fwd(Geo g, const float *__restrict__ h1, Acc stats1, double n, float *__restrict__ P, int off_m1,
                         float *__restrict__ bn1_out, int off_w2, float *__restrict__ h2, Acc stats, const float *__restrict__ Pw,
                         bool fin)
{   // Pw = P, read-only view for the filters (see k_tiled_fwd)
    __shared__ float bn1[2 * W];
    // the pixel's row is requested before the statistics are added up, and the next iteration's before this one's arithmetic: at
    // 138 patches a thread has two pixels, and two exposed memory latencies were a third of the kernel (width 4: 7.5 us)
    constexpr bool kAhead = W <= 8;
    const int64_t pstride = (int64_t)gridDim.x * TB;
    float hn[W];
#pragma unroll
    for (int j = 0; j < W; ++j) hn[j] = 0.0f;
    if (kAhead) {
        const int64_t p0 = (int64_t)blockIdx.x * TB + threadIdx.x;
        if (p0 < g.npix) {
#pragma unroll
            for (int j = 0; j < W; ++j) hn[j] = h1[p0 * W + j];
        }
    }
    bn_from_slots<W>(stats1, g.nslot, n, bn1, P, off_m1, off_m1 + W, bn1_out, fin);
    const float *W2 = Pw + off_w2, *b2 = W2 + W * W;
    float s[W], q[W];
#pragma unroll
    for (int j = 0; j < W; ++j) s[j] = q[j] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        float hc[W];
#pragma unroll
        for (int j = 0; j < W; ++j) hc[j] = hn[j];
        if (kAhead && p + pstride < g.npix) {
#pragma unroll
            for (int j = 0; j < W; ++j) hn[j] = h1[(p + pstride) * W + j];
        }
        if (p < g.npix) {
            float h[W];
#pragma unroll
            for (int j = 0; j < W; ++j) h[j] = b2[j];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float a = fmaxf(((kAhead ? hc[i] : h1[p * W + i]) - bn1[i]) * bn1[W + i], 0.0f);
#pragma unroll
                for (int j = 0; j < W; ++j) h[j] = fmaf(a, W2[i * W + j], h[j]);
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                h2[p * W + j] = h[j];
                s[j] += h[j];
                q[j] = fmaf(h[j], h[j], q[j]);
            }
        }
    }
    float sq[2 * W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        sq[j] = s[j];
        sq[W + j] = q[j];
    }
    acc_add_n<2 * W>(stats, sq, g.nslot);
}

// the l_last pre-activation u = conv3x3_VALID(pad(relu(bn2(h2))) ++ edge) + b  (layers.py:491, 555-583, 651-670)
template <int W>
__device__ __forceinline__ void l_last_u(const Geo &g, int b, int r, int c, const float *__restrict__ h2,
                                         const float *__restrict__ bn2, const float *__restrict__ W3,
                                         const float *__restrict__ b3, float u[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = b3[k];
    for (int di = 0; di < 3; ++di) {
        const int rr = r + di - 1;
        for (int dj = 0; dj < 3; ++dj) {
            const int cc = c + dj - 1;
            const float *w = W3 + (di * 3 + dj) * (W + 1) * 4;
            if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {   // on the padding ring: zeros + indicator 1
#pragma unroll
                for (int k = 0; k < 4; ++k) u[k] += w[W * 4 + k];
            } else {
                const float *hp = h2 + ((int64_t)b * g.HW + rr * g.W + cc) * W;
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const float a = fmaxf((hp[i] - bn2[i]) * bn2[W + i], 0.0f);
#pragma unroll
                    for (int k = 0; k < 4; ++k) u[k] = fmaf(a, w[i * 4 + k], u[k]);
                }
            }
        }
    }
}

// BN2 + ReLU + l_last + affine transform of the second half (layers.py:355-375)
template <int W>
__global__ void k_c3_fwd(Geo g, const float *__restrict__ zin, const float *__restrict__ h2, Acc stats2, double n,
                         float *__restrict__ P, int off_m2, float *__restrict__ bn2_out, int off_w3,
                         float *__restrict__ zout, Acc ldacc, const float *__restrict__ Pw, bool fin, float *__restrict__ u_out)
{   // Pw = P, read-only view for the filters (see k_tiled_fwd);  u_out: keep l_last's output for the backward pass (wide couplings)
    __shared__ float bn2[2 * W];
    bn_from_slots<W>(stats2, g.nslot, n, bn2, P, off_m2, off_m2 + W, bn2_out, fin);
    const float *W3 = Pw + off_w3, *b3 = W3 + 36 * (W + 1), *logs = b3 + 4;
    const float sc = logs[4];
    const float e30 = expf(kLogscale * logs[0]), e31 = expf(kLogscale * logs[1]), e32 = expf(kLogscale * logs[2]),
                e33 = expf(kLogscale * logs[3]);   // uniform: once per thread, not once per pixel
    float l = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW);
            const int rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            float u[4];
            l_last_u<W>(g, b, r, c, h2, bn2, W3, b3, u);
            if (u_out) reinterpret_cast<float4 *>(u_out)[p] = make_float4(u[0], u[1], u[2], u[3]);
            const float4 zi = reinterpret_cast<const float4 *>(zin)[p];
            const float sh0 = u[0] * e30, sh1 = u[1] * e31;
            const float ls0 = sc * tanhf(u[2] * e32), ls1 = sc * tanhf(u[3] * e33);
            reinterpret_cast<float4 *>(zout)[p] = make_float4(zi.x, zi.y, fmaf(zi.z, expf(ls0), sh0), fmaf(zi.w, expf(ls1), sh1));
            l += ls0 + ls1;
        }
    }
    const float lv[1] = {l};
    acc_add_n<1>(ldacc, lv, g.nslot);
}

// Per-patch sum and sum of squares of the latent.  Patches of a multiple of 64 pixels (32x32, 64x64: every wavefront of the
// pixel loop sits inside one patch): one partial per wavefront, wp[2 w] / wp[2 w + 1] for wavefront w = p / 64 of the batch,
// added up in a fixed order by k_loss — the reported loss / sd_z are then reproducible bit for bit like the gradients.
// Other shapes: float atomics into s1 / s2 (cleared by k_prep).
__global__ void k_prior(Geo g, const float *__restrict__ z, float *__restrict__ s1, float *__restrict__ s2, float *__restrict__ wp)
{
    const bool per_wave = (g.HW & 63) == 0;
    NF_PIXEL_LOOP(g, p) {
        const bool valid = p < g.npix;
        const int b = valid ? (int)(p / g.HW) : -1;
        float a = 0.0f, q = 0.0f;
        if (valid) {
            const float4 v = reinterpret_cast<const float4 *>(z)[p];
            a = v.x + v.y + v.z + v.w;
            q = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        if (per_wave) {
            const float sa = wsum(a), sq = wsum(q);
            if ((threadIdx.x & 63) == 0 && valid) {
                wp[2 * (p >> 6)] = sa;
                wp[2 * (p >> 6) + 1] = sq;
            }
        } else {
            patch_add(s1, b, valid, a);
            patch_add(s2, b, valid, q);
        }
    }
}

// loss = mean_b nll_b, sd_z = mean_b sqrt(var_hwc z_b)   (noise_flow_model.py:458-484)
__global__ void k_loss(int B, double n, Acc ld0, int n_layers, int nslot, const float *__restrict__ s1,
                       const float *__restrict__ s2, const float *__restrict__ wp, int waves_per_patch,
                       const double *__restrict__ ldc, float *__restrict__ out)
{
    __shared__ double sh[2][TB];
    __shared__ double shl[TB / 64];
    // sum over layers and patches of the data-dependent log-dets: the layers are dealt to the wavefronts
    double lpart = 0.0;
    for (int l = threadIdx.x >> 6; l < n_layers; l += TB / 64) lpart += acc_total(ld0 + l, nslot);
    if ((threadIdx.x & 63) == 0) shl[threadIdx.x >> 6] = lpart;
    __syncthreads();
    double ldsum = 0.0;
#pragma unroll
    for (int i = 0; i < TB / 64; ++i) ldsum += shl[i];
    double a = 0.0, d = 0.0;
    for (int b = threadIdx.x; b < B; b += TB) {
        double sum1, sum2;
        if (waves_per_patch > 0) {          // the wavefront partials of k_prior, in wavefront order
            sum1 = sum2 = 0.0;
            for (int w = 0; w < waves_per_patch; ++w) {
                sum1 += (double)wp[2 * ((size_t)b * waves_per_patch + w)];
                sum2 += (double)wp[2 * ((size_t)b * waves_per_patch + w) + 1];
            }
        } else {
            sum1 = s1[b];
            sum2 = s2[b];
        }
        a += 0.5 * n * 1.8378770664093453 + 0.5 * sum2 - ldc[0];
        const double m = sum1 / n;
        double v = sum2 / n - m * m;
        d += sqrt(v > 0.0 ? v : 0.0);
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = d;
    __syncthreads();
    for (int o = TB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)((sh[0][0] - ldsum) / B);
        out[1] = (float)(sh[1][0] / B);
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// d loss / d z of the prior term: z / B
__global__ void k_dz_init(Geo g, const float *__restrict__ z, float invB, float *__restrict__ dz)
{
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 v = reinterpret_cast<const float4 *>(z)[p];
            reinterpret_cast<float4 *>(dz)[p] = make_float4(v.x * invB, v.y * invB, v.z * invB, v.w * invB);
        }
    }
}

// gain4: z_out = z_in / g
__global__ void k_scale_bwd(Geo g, const float *__restrict__ zout, const float *__restrict__ gain, float *__restrict__ dz,
                            Acc dgain)
{
    const float inv = 1.0f / gain[0];
    float acc = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 d = reinterpret_cast<const float4 *>(dz)[p], z = reinterpret_cast<const float4 *>(zout)[p];
            acc -= (d.x * z.x + d.y * z.y + d.z * z.z + d.w * z.w) * inv;
            reinterpret_cast<float4 *>(dz)[p] = make_float4(d.x * inv, d.y * inv, d.z * inv, d.w * inv);
        }
    }
    acc_add(dgain, acc, g.nslot);
}

// sdn5: z = x / sqrt(a y + b), loss += (1/B) sum log scale
__global__ void k_sdn_bwd(Geo g, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ ab,
                          float invB, float *__restrict__ dz, Acc dab)
{
    const float a = ab[0], b = ab[1];
    float ga = 0.0f, gb = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 xv = reinterpret_cast<const float4 *>(x)[p], yv = reinterpret_cast<const float4 *>(y)[p];
            float4 d = reinterpret_cast<const float4 *>(dz)[p];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
            float ds[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float s2 = fmaf(a, ys[k], b), is = rsqrtf(s2);
                // d loss / d scale = -dz x / scale^2 + invB / scale ;  d scale / d(a,b) = (y, 1) / (2 scale)
                const float gs = (-ds[k] * xs[k] / s2 + invB * is) * 0.5f * is;
                ga = fmaf(gs, ys[k], ga);
                gb += gs;
                ds[k] *= is;
            }
            reinterpret_cast<float4 *>(dz)[p] = make_float4(ds[0], ds[1], ds[2], ds[3]);
        }
    }
    const float gab[2] = {ga, gb};
    acc_add_n<2>(dab, gab, g.nslot);
}

// 1x1 mix: z_out = z_in A
__global__ void k_mix_bwd(Geo g, const float *__restrict__ zin, const float *__restrict__ A, float *__restrict__ dz,
                          Acc dA)
{
    float m[16], acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        m[i] = A[i];
        acc[i] = 0.0f;
    }
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 zv = reinterpret_cast<const float4 *>(zin)[p], dv = reinterpret_cast<const float4 *>(dz)[p];
            const float zi[4] = {zv.x, zv.y, zv.z, zv.w}, d[4] = {dv.x, dv.y, dv.z, dv.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = m[i * 4] * d[0] + m[i * 4 + 1] * d[1] + m[i * 4 + 2] * d[2] + m[i * 4 + 3] * d[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = fmaf(zi[i], d[j], acc[i * 4 + j]);
            }
            reinterpret_cast<float4 *>(dz)[p] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    acc_add_n<16>(dA, acc, g.nslot);
}

// coupling, stage 1: through the affine transform, tanh, exp(3 logs); leaves d loss / d u in `gu`,
// d loss / d z1 in dz[2:4]; accumulates d rescaling_scale, d logs, d l_last/b
template <int W>
__global__ void k_c3_bwd(Geo g, const float *__restrict__ zin, const float *__restrict__ h2, const float *__restrict__ bn2,
                         const float *__restrict__ P, int off_w3, float invB, float *__restrict__ dz,
                         float *__restrict__ gu, Acc G, const float *__restrict__ zlat, const float *__restrict__ u_in)
{   // zlat != null: this is the first stage of the backward pass; d loss / d latent = latent / B is formed here (k_dz_init)
    // u_in != null: l_last's output was kept by the forward pass (wide couplings: recomputing it is 1 188 MAC per pixel)
    const float *W3 = P + off_w3, *b3 = W3 + 36 * (W + 1), *logs = b3 + 4;
    const float sc = logs[4];
    const int off_b3 = off_w3 + 36 * (W + 1);
    float e3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e3[k] = expf(kLogscale * logs[k]);
    float g_s = 0.0f, g_logs[4] = {0.f, 0.f, 0.f, 0.f}, g_b3[4] = {0.f, 0.f, 0.f, 0.f};
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            float u[4];
            if (u_in) {
                const float4 uv = reinterpret_cast<const float4 *>(u_in)[p];
                u[0] = uv.x; u[1] = uv.y; u[2] = uv.z; u[3] = uv.w;
            } else {
                l_last_u<W>(g, b, r, c, h2, bn2, W3, b3, u);
            }
            const float4 zi = reinterpret_cast<const float4 *>(zin)[p];
            float4 d;
            if (zlat) {
                const float4 zl = reinterpret_cast<const float4 *>(zlat)[p];
                d = make_float4(zl.x * invB, zl.y * invB, zl.z * invB, zl.w * invB);
            } else {
                d = reinterpret_cast<const float4 *>(dz)[p];
            }
            const float z1[2] = {zi.z, zi.w}, gx1[2] = {d.z, d.w};
            float go[4], o[4], gz1[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = u[k] * e3[k];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float t = tanhf(o[2 + k]), E = expf(sc * t);
                gz1[k] = gx1[k] * E;
                const float gls = gx1[k] * z1[k] * E - invB;   // loss = mean(-(sum ls + ...))
                g_s = fmaf(gls, t, g_s);
                go[k] = gx1[k];                              // shift
                go[2 + k] = gls * sc * (1.0f - t * t);       // raw
            }
            float guv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g_logs[k] = fmaf(kLogscale * go[k], o[k], g_logs[k]);
                guv[k] = go[k] * e3[k];
                g_b3[k] += guv[k];
            }
            reinterpret_cast<float4 *>(gu)[p] = make_float4(guv[0], guv[1], guv[2], guv[3]);
            d.z = gz1[0];
            d.w = gz1[1];
            reinterpret_cast<float4 *>(dz)[p] = d;
        }
    }
    const float tail[9] = {g_b3[0], g_b3[1], g_b3[2], g_b3[3], g_logs[0], g_logs[1], g_logs[2], g_logs[3], g_s};
    acc_add_n<9>(G + off_b3, tail, g.nslot);   // l_last/b, l_last/logs, rescaling_scale are adjacent
}

// d l_last/W: one filter tap per blockIdx.y
template <int W>
__global__ void k_w3_grad(Geo g, const float *__restrict__ h2, const float *__restrict__ bn2, const float *__restrict__ gu,
                          int off_w3, Acc G)
{
    const int tap = blockIdx.y, di = tap / 3, dj = tap - di * 3;
    float acc[W + 1][4];
#pragma unroll
    for (int i = 0; i <= W; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            const int rr = r + di - 1, cc = c + dj - 1;
            const float4 gv = reinterpret_cast<const float4 *>(gu)[p];
            const float gk[4] = {gv.x, gv.y, gv.z, gv.w};
            if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[W][k] += gk[k];
            } else {
                const float *hp = h2 + ((int64_t)b * g.HW + rr * g.W + cc) * W;
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    const float a = fmaxf((hp[i] - bn2[i]) * bn2[W + i], 0.0f);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][k] = fmaf(a, gk[k], acc[i][k]);
                }
            }
        }
    }
    float flat[(W + 1) * 4];
#pragma unroll
    for (int i = 0; i <= W; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) flat[i * 4 + k] = acc[i][k];
    acc_add_n<(W + 1) * 4>(G + off_w3 + tap * (W + 1) * 4, flat, g.nslot);
}

// coupling, stage 2: transposed l_last + ReLU mask -> d loss / d xhat2 (into t1) and the two
// batch sums the BN backward needs
template <int W>
__global__ void k_c3_dh(Geo g, const float *__restrict__ h2, const float *__restrict__ bn2, const float *__restrict__ P,
                        int off_w3, const float *__restrict__ gu, float *__restrict__ t1, Acc bstats)
{
    const float *W3 = P + off_w3;
    float s[W], q[W];
#pragma unroll
    for (int j = 0; j < W; ++j) s[j] = q[j] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            float gh[W];
#pragma unroll
            for (int i = 0; i < W; ++i) gh[i] = 0.0f;
            for (int di = 0; di < 3; ++di) {
                const int qr = r - (di - 1);
                if (qr < 0 || qr >= g.H) continue;
                for (int dj = 0; dj < 3; ++dj) {
                    const int qc = c - (dj - 1);
                    if (qc < 0 || qc >= g.W) continue;
                    const float4 gv = reinterpret_cast<const float4 *>(gu)[(int64_t)b * g.HW + qr * g.W + qc];
                    const float *w = W3 + (di * 3 + dj) * (W + 1) * 4;
#pragma unroll
                    for (int i = 0; i < W; ++i)
                        gh[i] += w[i * 4] * gv.x + w[i * 4 + 1] * gv.y + w[i * 4 + 2] * gv.z + w[i * 4 + 3] * gv.w;
                }
            }
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float xh = (h2[p * W + i] - bn2[i]) * bn2[W + i];
                const float gx = xh > 0.0f ? gh[i] : 0.0f;
                t1[p * W + i] = gx;
                s[i] += gx;
                q[i] = fmaf(gx, xh, q[i]);
            }
        }
    }
    float sq[2 * W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        sq[j] = s[j];
        sq[W + j] = q[j];
    }
    acc_add_n<2 * W>(bstats, sq, g.nslot);
}

// G[i] = sum of the partials of value i, one wavefront per value
__global__ void k_reduce(int n, const float *__restrict__ part, int nslot, double *__restrict__ G)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = threadIdx.x; k < nslot; k += 64) s += (double)part[(size_t)i * NSLOT + k];
    s = wsum(s);
    if (threadIdx.x == 0) G[i] = s;
}

// coupling, stage 3: BN2 backward -> g_h2 (t1, in place), d l_2/b; transposed l_2 + ReLU mask ->
// d loss / d xhat1 (t2) and its two batch sums
template <int W>
__global__ void k_c2_bwd(Geo g, const float *__restrict__ h1, const float *__restrict__ bn1, const float *__restrict__ h2,
                         const float *__restrict__ bn2, Acc bstats2, double n, const float *__restrict__ P,
                         int off_w2, float *__restrict__ t1, float *__restrict__ t2, Acc bstats, Acc G, const float *__restrict__ pre)
{
    __shared__ float bb2[2 * W];
    // (rows requested ahead of their use, as in k_c2_fwd)
    constexpr bool kAhead = W <= 8;
    const int64_t pstride = (int64_t)gridDim.x * TB;
    float n2[W], n1[W], nt[W];
#pragma unroll
    for (int j = 0; j < W; ++j) n2[j] = n1[j] = nt[j] = 0.0f;
    if (kAhead) {
        const int64_t p0 = (int64_t)blockIdx.x * TB + threadIdx.x;
        if (p0 < g.npix) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                n2[j] = h2[p0 * W + j];
                nt[j] = t1[p0 * W + j];
                n1[j] = h1[p0 * W + j];
            }
        }
    }
    bnb_from_slots<W>(bstats2, g.nslot, n, bb2, pre);
    const float *W2 = P + off_w2;
    float s[W], q[W], gb[W];
#pragma unroll
    for (int j = 0; j < W; ++j) s[j] = q[j] = gb[j] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        float c2[W], c1[W], ct[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            c2[j] = n2[j];
            c1[j] = n1[j];
            ct[j] = nt[j];
        }
        if (kAhead && p + pstride < g.npix) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                n2[j] = h2[(p + pstride) * W + j];
                nt[j] = t1[(p + pstride) * W + j];
                n1[j] = h1[(p + pstride) * W + j];
            }
        }
        if (p < g.npix) {
            float gh2[W];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const float xh = ((kAhead ? c2[j] : h2[p * W + j]) - bn2[j]) * bn2[W + j];
                gh2[j] = bn2[W + j] * ((kAhead ? ct[j] : t1[p * W + j]) - bb2[j] - xh * bb2[W + j]);
                t1[p * W + j] = gh2[j];
                gb[j] += gh2[j];
            }
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const float xh = ((kAhead ? c1[i] : h1[p * W + i]) - bn1[i]) * bn1[W + i];
                float gh = 0.0f;
#pragma unroll
                for (int j = 0; j < W; ++j) gh = fmaf(W2[i * W + j], gh2[j], gh);
                const float gx = xh > 0.0f ? gh : 0.0f;
                t2[p * W + i] = gx;
                s[i] += gx;
                q[i] = fmaf(gx, xh, q[i]);
            }
        }
    }
    float sq[2 * W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        sq[j] = s[j];
        sq[W + j] = q[j];
    }
    acc_add_n<2 * W>(bstats, sq, g.nslot);
    acc_add_n<W>(G + off_w2 + W * W, gb, g.nslot);
}

// d l_2/W: one input channel per blockIdx.y
template <int W>
__global__ void k_w2_grad(Geo g, const float *__restrict__ h1, const float *__restrict__ bn1, const float *__restrict__ t1,
                          int off_w2, Acc G)
{
    const int i = blockIdx.y;
    const float m = bn1[i], rs = bn1[W + i];
    float acc[W];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float a = fmaxf((h1[p * W + i] - m) * rs, 0.0f);
#pragma unroll
            for (int j = 0; j < W; ++j) acc[j] = fmaf(a, t1[p * W + j], acc[j]);
        }
    }
    acc_add_n<W>(G + off_w2 + i * W, acc, g.nslot);
}

// coupling, stage 4: BN1 backward -> g_h1 (t2, in place), d l_1/b
template <int W>
__global__ void k_c1_bwd(Geo g, const float *__restrict__ h1, const float *__restrict__ bn1, Acc bstats1, double n,
                         int off_b1, float *__restrict__ t2, Acc G, const float *__restrict__ pre)
{
    __shared__ float bb1[2 * W];
    bnb_from_slots<W>(bstats1, g.nslot, n, bb1, pre);
    float gb[W];
#pragma unroll
    for (int j = 0; j < W; ++j) gb[j] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const float xh = (h1[p * W + j] - bn1[j]) * bn1[W + j];
                const float gh = bn1[W + j] * (t2[p * W + j] - bb1[j] - xh * bb1[W + j]);
                t2[p * W + j] = gh;
                gb[j] += gh;
            }
        }
    }
    acc_add_n<W>(G + off_b1, gb, g.nslot);
}

// k_c1_bwd for wide couplings: one float4 of the [pixel][W] tensors per thread and step, so that a wavefront touches 1 KB of
// consecutive memory per load (the per-pixel form walks 128-byte rows with a 128-byte lane stride); the grid stride is a
// multiple of W / 4, so a thread keeps its 4 channels.
template <int W>
__global__ __launch_bounds__(256) void k_c1_bwd_flat(Geo g, const float *__restrict__ h1, const float *__restrict__ bn1,
                                                     const float *__restrict__ bb1, int off_b1, float *__restrict__ t2, Acc G)
{
    __shared__ float red[256 * 4];
    constexpr int Q = W / 4;
    const int t = threadIdx.x, cg = t % Q;
    float m[4], rs[4], ba[4], bq[4], gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = bn1[4 * cg + k];
        rs[k] = bn1[W + 4 * cg + k];
        ba[k] = bb1[4 * cg + k];
        bq[k] = bb1[W + 4 * cg + k];
    }
    const int64_t total = g.npix * Q;
    for (int64_t e = (int64_t)blockIdx.x * 256 + t; e < total; e += (int64_t)gridDim.x * 256) {
        const float4 h = reinterpret_cast<const float4 *>(h1)[e], gx = reinterpret_cast<const float4 *>(t2)[e];
        const float hv[4] = {h.x, h.y, h.z, h.w}, gv[4] = {gx.x, gx.y, gx.z, gx.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (hv[k] - m[k]) * rs[k];
            o[k] = rs[k] * (gv[k] - ba[k] - xh * bq[k]);
            gb[k] += o[k];
        }
        reinterpret_cast<float4 *>(t2)[e] = make_float4(o[0], o[1], o[2], o[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[t * 4 + k] = gb[k];
    __syncthreads();
    if (t < W) {   // channel t: threads with t' % Q == t / 4, component t % 4
        float tot = 0.0f;
        for (int u = t >> 2; u < 256; u += Q) tot += red[u * 4 + (t & 3)];
        float *d = (G + off_b1 + t).p;
        d[blockIdx.x] = tot;
        for (int q = blockIdx.x + gridDim.x; q < g.nslot; q += gridDim.x) d[q] = 0.0f;
    }
}

// d l_1/W: one filter tap per blockIdx.y
template <int W>
__global__ void k_w1_grad(Geo g, const float *__restrict__ zin, const float *__restrict__ t2, int off_w1, Acc G)
{
    const int tap = blockIdx.y, di = tap / 3, dj = tap - di * 3;
    float acc[2][W];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[0][j] = acc[1][j] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            const int rr = r + di - 1, cc = c + dj - 1;
            if (rr >= 0 && rr < g.H && cc >= 0 && cc < g.W) {
                const float2 v = *reinterpret_cast<const float2 *>(zin + ((int64_t)b * g.HW + rr * g.W + cc) * 4);
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const float gh = t2[p * W + j];
                    acc[0][j] = fmaf(v.x, gh, acc[0][j]);
                    acc[1][j] = fmaf(v.y, gh, acc[1][j]);
                }
            }
        }
    }
    float flat[2 * W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        flat[j] = acc[0][j];
        flat[W + j] = acc[1][j];
    }
    acc_add_n<2 * W>(G + off_w1 + tap * 2 * W, flat, g.nslot);
}

// coupling, stage 5: transposed l_1 adds the CNN path into d loss / d z0.
// MIX = the backward of the preceding Conv2d1x1 is folded in (per pixel: dA += z_in^T d, d <- d A^T).
template <int W, bool MIX>
__global__ void k_c1_dz(Geo g, const float *__restrict__ t2, const float *__restrict__ P, int off_w1, float *__restrict__ dz,
                        const float *__restrict__ zmix_in, const float *__restrict__ A, Acc dA, const float *__restrict__ dz_in)
{   // dz_in != null: d loss / d z is read from there (k_c3_dh_mfma's second buffer) and written to dz
    const float *W1 = P + off_w1;
    float m[16], acc[16];
    if (MIX) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            m[i] = A[i];
            acc[i] = 0.0f;
        }
    }
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            float a0 = 0.0f, a1 = 0.0f;
            for (int di = 0; di < 3; ++di) {
                const int qr = r - (di - 1);
                if (qr < 0 || qr >= g.H) continue;
                for (int dj = 0; dj < 3; ++dj) {
                    const int qc = c - (dj - 1);
                    if (qc < 0 || qc >= g.W) continue;
                    const float *gh = t2 + ((int64_t)b * g.HW + qr * g.W + qc) * W;
                    const float *w = W1 + (di * 3 + dj) * 2 * W;
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        a0 = fmaf(w[j], gh[j], a0);
                        a1 = fmaf(w[W + j], gh[j], a1);
                    }
                }
            }
            if (MIX) {
                const float4 dv = reinterpret_cast<const float4 *>(dz_in ? dz_in : dz)[p], zv = reinterpret_cast<const float4 *>(zmix_in)[p];
                const float d[4] = {dv.x + a0, dv.y + a1, dv.z, dv.w}, zi[4] = {zv.x, zv.y, zv.z, zv.w};
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = m[i * 4] * d[0] + m[i * 4 + 1] * d[1] + m[i * 4 + 2] * d[2] + m[i * 4 + 3] * d[3];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i * 4 + j] = fmaf(zi[i], d[j], acc[i * 4 + j]);
                }
                reinterpret_cast<float4 *>(dz)[p] = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                if (dz_in) {
                    const float4 v = reinterpret_cast<const float4 *>(dz_in)[p];
                    reinterpret_cast<float4 *>(dz)[p] = make_float4(v.x + a0, v.y + a1, v.z, v.w);
                } else {
                    float2 *d = reinterpret_cast<float2 *>(dz + p * 4);
                    const float2 v = *d;
                    *d = make_float2(v.x + a0, v.y + a1);
                }
            }
        }
    }
    if (MIX) acc_add_n<16>(dA, acc, g.nslot);
}

#include "nf_train_tiled.h"   // k_tiled_fwd, k_tiled_CA: one workgroup per patch, widths 4 / 8
#include "nf_train_wide.h"    // widths 16 / 32 on the matrix cores
#include "nf_train_pr.h"      // width 32 on 32x32 patches: patch-resident stages (no [pixel][32] tensor in HBM)

// chain rule of the scalar parameterisations: dA -> PLU factors, d(a,b) -> sdn5 variables, gain_val
__global__ void k_finish(TLayers ls, const float *__restrict__ P, CondIdx ci, int HW, const double *__restrict__ dAbuf,
                         const double *__restrict__ dabbuf, const double *__restrict__ dgbuf, double *__restrict__ G)
{
    const int l = threadIdx.x;
    if (l >= ls.n) return;
    const TLayer L = ls.l[l];
    const float *p = P + L.off;
    double *gp = G + L.off;
    if (L.kind == NF_LAYER_PERMUTE) {
        // no parameters
    } else if (L.kind == NF_LAYER_CONV1X1_NONE) {                          // A is the variable: dA - H W A^-T
        double M[4][4], Mi[4][4], lad;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) M[i][j] = p[i * 4 + j];
        inv4(M, Mi, lad);
        const double *dA = dAbuf + L.aux * 16;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) gp[i * 4 + j] = dA[i * 4 + j] - (double)HW * Mi[j][i];
    } else if (L.kind == NF_LAYER_CONV1X1_LU2) {
        double Pm[4][4], Lm[4][4], Um[4][4], dM[4][4];
        lu2_matrices(p, Pm, Lm, Um);
        const double *dA = dAbuf + L.aux * 16;
        for (int i = 0; i < 4; ++i)          // dM = P^T dA   (A = P M, M = L U)
            for (int j = 0; j < 4; ++j) {
                double sacc = 0.0;
                for (int k = 0; k < 4; ++k) sacc += Pm[k][i] * dA[k * 4 + j];
                dM[i][j] = sacc;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double dl = 0.0, du = 0.0;
                for (int k = 0; k < 4; ++k) {
                    dl += dM[i][k] * Um[j][k];   // dL = dM U^T
                    du += Lm[k][i] * dM[k][j];   // dU = L^T dM
                }
                gp[16 + i * 4 + j] = j < i ? dl : 0.0;                     // the mask of matrix_param.py:171-173
                gp[40 + i * 4 + j] = j > i ? du : 0.0;
                if (i == j) gp[36 + i] = du * Um[i][i] - (double)HW;       // log_S: diagonal + log-det term
            }
    } else if (L.type == NF_LAYER_CONV1X1) {
        double Pm[4][4], Lm[4][4], Um[4][4], dM[4][4], dL[4][4], dU[4][4];
        plu_matrices(p, Pm, Lm, Um);
        const double *dA = dAbuf + L.aux * 16;
        for (int i = 0; i < 4; ++i)          // dM = P^T dA   (A = P M, M = L U)
            for (int j = 0; j < 4; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Pm[k][i] * dA[k * 4 + j];
                dM[i][j] = s;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0.0, t = 0.0;
                for (int k = 0; k < 4; ++k) {
                    s += dM[i][k] * Um[j][k];   // dL = dM U^T
                    t += Lm[k][i] * dM[k][j];   // dU = L^T dM
                }
                dL[i][j] = s;
                dU[i][j] = t;
            }
        for (int i = 0; i < 4; ++i) gp[20 + i] = dU[i][i] * Um[i][i] - (double)HW;   // log_S: diagonal + log-det term
        for (int k = 0; k < 6; ++k) {
            gp[24 + k] = dL[kLr[k]][kLc[k]];
            gp[30 + k] = dU[kUr[k]][kUc[k]];
        }
    } else if (L.kind == NF_LAYER_SDN || L.kind == NF_LAYER_SDN1 || L.kind == NF_LAYER_SDN2 || L.kind == NF_LAYER_SDN3 ||
               L.kind == NF_LAYER_SDN6) {
        double a, b;
        sdn_ab(L.kind, p, ci, a, b);
        const double ga = dabbuf[L.aux * 2], gb = dabbuf[L.aux * 2 + 1];
        if (L.kind == NF_LAYER_SDN6) {
            const double c = p[12], cp = exp(c * (double)p[7 + ci.cam_idx]);
            const double g = ci.iso_idx >= 0 ? (double)p[2 + ci.iso_idx] : 0.0;
            gp[0] = ga * a * c;
            gp[1] = gb * b * c;
            if (ci.iso_idx >= 0) gp[2 + ci.iso_idx] = -ga * a * c * cp;
            gp[7 + ci.cam_idx] = -ga * a * c * c * g * cp;
        } else {
            const double A0 = sigm(p[0]), B0 = sigm(p[1]), dA0 = A0 * (1.0 - A0), dB0 = B0 * (1.0 - B0);
            const int k = 2 + table_idx(ci);
            if (L.kind == NF_LAYER_SDN) {
                gp[0] = ga * dA0;
                gp[1] = gb * dB0;
            } else if (L.kind == NF_LAYER_SDN1) {                          // a = A0 / gain, b = B0, gain = exp(1e-2 t) iso
                gp[0] = ga * a * (1.0 - A0);
                gp[1] = gb * dB0;
                gp[k] = -ga * a * 1e-2;
            } else if (L.kind == NF_LAYER_SDN2) {                          // a = A0, b = gain B0, gain = exp(1e-1 t) iso
                gp[0] = ga * dA0;
                gp[1] = gb * b * (1.0 - B0);
                gp[k] = gb * b * 1e-1;
            } else {                                                       // SDN3: a = gain A0, b = gain^2 B0
                gp[0] = ga * a * (1.0 - A0);
                gp[1] = gb * b * (1.0 - B0);
                gp[k] = (ga * a + 2.0 * gb * b) * 1e-1;
            }
        }
    } else if (L.kind == NF_LAYER_GAIN || L.kind == NF_LAYER_GAIN1 || L.kind == NF_LAYER_GAIN2 || L.kind == NF_LAYER_GAIN3) {
        double sv, K;
        gain_s(L.kind, p, ci, HW, sv, K);
        const double gs = dgbuf[L.aux] + K / sv;                            // data path + the log-det term
        const double iso = ci.iso;
        if (L.kind == NF_LAYER_GAIN) {
            const double A0 = sigm(p[0]), B0 = sigm(p[1]);
            gp[0] = gs * iso * A0 * (1.0 - A0);
            gp[1] = gs * B0 * (1.0 - B0);
        } else if (L.kind == NF_LAYER_GAIN1) {
            gp[0] = gs * iso * exp(1e-5 * (double)p[0]) * 1e-5;
            gp[1] = gs * exp(1e-5 * (double)p[1]) * 1e-5;
        } else {
            gp[table_idx(ci)] = gs * sv * (L.kind == NF_LAYER_GAIN2 ? 1e-1 : 1e-5);
        }
    } else if (L.kind == NF_LAYER_SDN5) {
        double a, b;
        sdn5_eval(p, ci, a, b);
        const double ga = dabbuf[L.aux * 2], gb = dabbuf[L.aux * 2 + 1];
        const double c_i = p[22];
        double cp[3];
        for (int r = 0; r < 3; ++r) cp[r] = exp(c_i * (double)p[7 + r * 5 + ci.cam_idx]);
        const double beta1 = p[0], beta2 = p[1];
        const double gpar = ci.iso_idx >= 0 ? (double)p[2 + ci.iso_idx] : 0.0;
        gp[0] = ga * a * c_i * cp[0];
        gp[1] = gb * b * c_i * cp[1];
        if (ci.iso_idx >= 0) gp[2 + ci.iso_idx] = -ga * a * c_i * cp[2];
        gp[7 + 0 * 5 + ci.cam_idx] = ga * a * c_i * c_i * beta1 * cp[0];
        gp[7 + 1 * 5 + ci.cam_idx] = gb * b * c_i * c_i * beta2 * cp[1];
        gp[7 + 2 * 5 + ci.cam_idx] = -ga * a * c_i * c_i * gpar * cp[2];
    } else if (L.kind == NF_LAYER_SDN4) {
        const double gpar = ci.iso_idx >= 0 ? (double)p[2 + ci.iso_idx] : 0.0;
        const double a = exp((double)p[0]) / (exp(gpar) * (double)ci.iso), b = exp((double)p[1]);
        const double ga = dabbuf[L.aux * 2], gb = dabbuf[L.aux * 2 + 1];
        gp[0] = ga * a;
        gp[1] = gb * b;
        if (ci.iso_idx >= 0) gp[2 + ci.iso_idx] = -ga * a;
    } else if (L.kind == NF_LAYER_GAIN4) {
        gp[0] = dgbuf[L.aux] + (double)HW * 4.0 / (double)p[0];
    }
}

__global__ void k_grads_out(int n, const double *__restrict__ G, const uint8_t *__restrict__ mask, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mask[i] ? (float)G[i] : 0.0f;
}

// tf.train.AdamOptimizer._apply_dense (lr_t computed by the host from the step count)
__global__ void k_adam(int n, float *__restrict__ P, const float *__restrict__ grads, float *__restrict__ m,
                       float *__restrict__ v, const uint8_t *__restrict__ mask, float lr_t, float b1, float b2, float eps)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i]) {
        const float gr = grads[i];
        const float mi = m[i] + (gr - m[i]) * (1.0f - b1);
        const float vi = v[i] + (gr * gr - v[i]) * (1.0f - b2);
        m[i] = mi;
        v[i] = vi;
        P[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// tf.train.MomentumOptimizer(lr, 0.9): accum = 0.9 accum + g; theta -= lr accum
__global__ void k_momentum(int n, float *__restrict__ P, const float *__restrict__ grads, float *__restrict__ acc,
                           const uint8_t *__restrict__ mask, float lr, float mom)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i]) {
        const float a = mom * acc[i] + grads[i];
        acc[i] = a;
        P[i] -= lr * a;
    }
}

struct Cpl {      // per-coupling workspace
    float *h1 = nullptr, *h2 = nullptr;
    float *u = nullptr;                   // l_last's output, kept for the backward pass (wide couplings only)
    int f_bn1, f_bn2, f_bb1, f_bb2;       // offsets into the float scalar buffer
    int d_st1, d_st2, d_bs1, d_bs2;       // offsets into the double buffer
};

}  // namespace

struct nf_trainer {
    nf_config cfg;
    int device = 0;
    int64_t max_batch = 0;
    int optimizer = 0;
    int64_t step = 0;
    int n_params = 0;
    int width = 0;
    std::vector<nf_layer_desc> layers;
    TLayers tl;
    std::vector<Cpl> cpl;           // indexed by TLayer::aux of coupling layers
    float *d_params = nullptr, *d_m = nullptr, *d_v = nullptr, *d_gradf = nullptr;
    uint8_t *d_mask = nullptr;
    double *d_dbl = nullptr;        // [0,n_params) gradients, then dA / dab / dgain / BN sums, last: ldc
    size_t n_dbl = 0;
    float *d_part = nullptr;        // per-workgroup partials of every reducible value: [n_dbl - 1 - hole rows][NSLOT]
    // Values that never take slotted sums have no rows: the l_2/W bodies of couplings on the GEMM path (their gradients come whole
    // from the GEMMs; at width 512 they would be 9 of the 10 GB).  `holes` lists them (first value, count), ascending.
    std::vector<std::pair<int, int>> holes;
    size_t hole_rows = 0;
    Acc acc(int idx) const
    {
        size_t row = (size_t)idx;
        for (const auto &h : holes) {
            if (h.first >= idx) break;
            row -= (size_t)h.second;
        }
        return Acc{d_part + row * NSLOT};
    }
    size_t part_floats() const { return (n_dbl - 1 - hole_rows) * NSLOT; }
    int d_dA = 0, d_dab = 0, d_dg = 0, d_ld0 = 0, d_ldc = 0;
    float *d_flt = nullptr;         // A matrices, sdn5 (a,b), BN scalars
    size_t n_flt = 0;
    int f_A = 0, f_ab = 0, f_s = 0;
    float *d_patch = nullptr;       // s1[B], s2[B]
    float *d_wpart = nullptr;       // per-wavefront (sum z, sum z^2) partials of k_prior (patches of a multiple of 64 pixels)
    std::vector<float *> zs;        // zs[l] = input of layer l (zs[0] is the caller's x), zs[n] = latent
    // backward temporaries, double-buffered by coupling parity: the filter-gradient kernels of one
    // coupling run on `side` while the main stream is already in the next coupling
    float *t1[3] = {nullptr, nullptr, nullptr}, *t2[3] = {nullptr, nullptr, nullptr}, *gu[3] = {nullptr, nullptr, nullptr}, *dz = nullptr,
          *dz2 = nullptr;   // second d loss / d z buffer (wide couplings: k_c3_dh_mfma writes there what every workgroup of a patch still reads from dz)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork[3] = {nullptr, nullptr, nullptr}, ev_done[3] = {nullptr, nullptr, nullptr};
    bool done_pending[3] = {false, false, false};
    int band_cap = 320;        // pixels (rows x width, halo included) a band kernel keeps in LDS
    bool serial = false;   // NF_TRAIN_SERIAL=1: everything on the caller's stream (kernel durations without overlap, for profiling)
    int wide_mfma = 4095;   // NF_TRAIN_WIDE_MFMA: width-32 stages on the matrix cores (bit 0 filter gradients, 1 l_2 forward, 2 l_2 backward, 3 statistics finalisers, 4 l_last forward, 5 l_last transposed, 6 l_1 transposed, 7 filter gradients inside the stage that holds their operands, 8 l_1 forward, 11 affine/tanh backward inside the transposed l_last; 0: layer kernels only)
    int pr = 1;            // NF_TRAIN_PR: width 32 on 32x32 patches on the patch-resident stages of nf_train_pr.h (1: 8 wavefronts per patch, 2: 4; + 4 / + 8: no forward / backward launch fusion, + 16: no side-stream d l_last/W; 0: the stage kernels of nf_train_wide.h)
    float *pr_img = nullptr;   // [couplings][PR_SIZE] packed weights of this step (k_pr_pack)
    // the coupling above the one a patch-resident forward is launched for, when only a Conv2d1x1 lies between them: its stage 0 rides in
    // this coupling's last launch (set by the layer loop; pr_f0_done: the next call's stage 0 already ran)
    const TLayer *pr_next = nullptr;
    const float *pr_next_A = nullptr;
    float *pr_next_zin = nullptr;
    bool pr_f0_done = false;
    // ... and in the backward pass: stage A of the coupling BELOW rides in this coupling's last launch
    const TLayer *pr_below = nullptr;
    const float *pr_below_zin = nullptr;
    bool pr_a_done = false;
    int pr_side_full = 0;      // NF_TRAIN_PR_SIDE_FULL=1 (A/B aid): one side workgroup per patch even when fewer CUs are idle
    int pr_both_max = 0;       // NF_TRAIN_PR_BOTH_MAX=<patches> (A/B aid): up to how many patches both products take the side stream (default CUs / 2)
    int tiled = 3;   // NF_TRAIN_TILED: bit 0 = tiled backward stages, bit 1 = tiled forward stages (0: layer kernels only)
    std::vector<void *> owned;
    bool has_sdn = false;
    bool needs_cam = false;         // an SDN5 / SDN6 layer is present: the camera id must be one of 0..4
    bool needs_cond = false;        // a gain layer that reads the ISO is present: cond must be given
    // cross-rank batch normalisation (nf_trainer_set_sync)
    nf_allreduce_fn sync_fn = nullptr;
    void *sync_user = nullptr;
    double *sync_buf = nullptr;     // caller-owned device buffer, >= 64 doubles
    int sync_world = 1;
    int sync_rc = 0;                // first non-zero status a callback returned during the current step
    // coupling widths without stage kernels of their own (nf_train_gemm.h): the operands of the matrix-core GEMMs of nf_train_mm.h
    int n_cu = 256;
    float *gz18 = nullptr, *gp36 = nullptr, *gq18 = nullptr;   // [pixels][20] windows (18 + 2 zero columns), [pixels][36] taps, [pixels][18]
    int nb_floor = 0;               // GEMM path: at least this many slots / workgroups of the pixel kernels (blocks_for)
    size_t gz18_stride = 0;         // trainer: every coupling keeps its windows for the backward pass (floats between two couplings')
    float *gpack = nullptr;         // every coupling's weights in the GEMMs' packed layouts (gemm_pack_floats(w) each; written by the forward pass)
    float *gdw = nullptr;           // filter gradients of every coupling as the pixel-K GEMMs leave them: 3 per coupling x gemm_part_floats(w)
    int gnp[3 * kMaxLayers] = {};   // how many partial products each of them holds (this step)
    bool gdual[kMaxLayers] = {};    // d l_last/W was left as [chunk][relu | mask][w][36] (k_mm_kpix <APRO 3>)
    // evaluation under batch statistics at the widths / patch sizes the fused kernels' statistics passes do not take (nf_bs_wide_*,
    // called by nf_*_batchstats): a trimmed trainer — every coupling on the GEMM path, ONE pair of hidden tensors, no backward state
    bool all_gemm = false, eval_only = false;
    float *ebuf = nullptr;          // [max_batch][H][W][4] working tensor (when the caller gives no output tensor)
    double *eldp = nullptr;         // [max_batch] per-patch data-dependent log-det
    float *emom = nullptr;          // [n_cpl][4][w] batch mean1, var1, mean2, var2 of the current call
    float *eAinv = nullptr;         // [n_mix][16] inverse Conv2d1x1 matrices (sampling direction)
    int n_mix = 0;
    bool gemm_c1_fused = false;     // NF_TRAIN_GEMM_C1=1: l_1 forward on the VALU kernel k_g_c1_fwd (widths that are a multiple of 4; A/B aid)
};

namespace {

int dev_alloc(nf_trainer *t, void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) return nf_fail_hip(e, "hipMalloc(trainer workspace)");
    t->owned.push_back(*p);
    return NF_OK;
}

inline unsigned blocks_for(int64_t npix)
{
    // grid-stride loops; at most NSLOT workgroups (= 2 per CU), one partial-sum slot each
    // ~2 pixels per thread for small minibatches (fewer slots to add up, fewer workgroups to
    // launch), every SIMD 4 waves deep for large ones
    int64_t b = (npix + 2 * TB - 1) / (2 * TB);
    return (unsigned)std::min<int64_t>(std::max<int64_t>(b, 1), NSLOT);
}
// The GEMM path wants more slots at small minibatches: a GEMM with batch sums runs one workgroup per slot and pixel-tile share
// (138 patches: 276 slots = ONE workgroup per CU at widths <= 128, where the channel axis is a single tile)
inline unsigned blocks_for(const nf_trainer *t, int64_t npix)
{
    int64_t b = blocks_for(npix);
    if (t->nb_floor) b = std::max<int64_t>(b, std::min<int64_t>(t->nb_floor, (npix + 127) / 128));
    return (unsigned)std::min<int64_t>(b, NSLOT);
}

struct Guard {
    int prev = -1;
    bool changed = false;
    int enter(int dev)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return nf_fail_hip(e, "hipGetDevice");
        if (prev != dev) {
            if ((e = hipSetDevice(dev)) != hipSuccess) return nf_fail_hip(e, "hipSetDevice");
            changed = true;
        }
        return NF_OK;
    }
    ~Guard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

// all-reduce `count` slotted sums (a, a+1, ...) over the ranks; no-op without nf_trainer_set_sync
void sync_slots(nf_trainer *t, Acc a, int count, int nslot, hipStream_t st)
{
    if (!t->sync_fn || t->sync_world < 2) return;
    for (int c0 = 0; c0 < count; c0 += 64) {   // the caller's buffer holds 64 doubles (wide couplings have up to 1 024 sums per group)
        const int cn = count - c0 < 64 ? count - c0 : 64;
        hipLaunchKernelGGL(k_slots_compact, dim3((unsigned)cn), dim3(64), 0, st, a + c0, nslot, t->sync_buf);
        const int rc = t->sync_fn(t->sync_user, t->sync_buf, (int64_t)cn, (void *)st);
        if (rc != 0 && t->sync_rc == 0) t->sync_rc = rc;
        hipLaunchKernelGGL(k_slots_scatter, dim3((unsigned)cn), dim3(64), 0, st, a + c0, nslot, (const double *)t->sync_buf);
    }
}

// per-patch kernels: how many workgroups share one patch's tiles — small minibatches leave CUs idle otherwise (138 patches on
// 256 CUs); bounded by the slots (one per workgroup)
inline int patch_split(const Geo &g)
{
    const int64_t npatch = g.npix / g.HW;
    return (int)std::max<int64_t>(1, std::min<int64_t>(4, std::min<int64_t>(g.nslot / npatch, 512 / npatch)));
}

}  // namespace
#include "nf_train_gemm.h"    // widths without stage kernels of their own: matrix-core GEMMs (nf_train_mm.h) + pixel kernels of run-time width
namespace {

// ---- width 32 on 32x32 patches: the patch-resident stages (nf_train_pr.h) ----
template <int NW>
int pr_set_attributes_nw()
{
    const size_t ca = std::max(pr_bwd_lds(2, NW), pr_bwd_lds(0, NW));
    struct { const void *fn; size_t lds; } ks[] = {
        {reinterpret_cast<const void *>(&k_pr_fwd<2, false, NW>), pr_fwd_lds(2, NW)},
        {reinterpret_cast<const void *>(&k_pr_fwd<2, false, NW, true>), pr_fwd_lds(2, NW)},
        {reinterpret_cast<const void *>(&k_pr_fwd<2, false, NW, false, 1>), pr_fwd_lds(2, NW)},
        {reinterpret_cast<const void *>(&k_pr_fwd<2, false, NW, false, 2>), pr_fwd_lds(2, NW)},
        {reinterpret_cast<const void *>(&k_pr_bwd<0, false, NW, 0>), pr_bwd_lds(0, NW)},
        {reinterpret_cast<const void *>(&k_pr_bwd<0, false, NW, 1>), pr_bwd_lds(0, NW)},
        {reinterpret_cast<const void *>(&k_pr_bwd<0, false, NW, 2>), pr_bwd_lds(0, NW)},
        {reinterpret_cast<const void *>(&k_pr_bwd<1, false, NW, 2>), pr_bwd_lds(1, NW)},
        {reinterpret_cast<const void *>(&k_pr_bwd<2, false, NW, 2>), pr_bwd_lds(2, NW)},
        {reinterpret_cast<const void *>(&k_pr_bwd<2, true, NW, 2>), pr_bwd_lds(2, NW)},
        {reinterpret_cast<const void *>(&k_pr_bwd_CA<true, NW, 0, 0>), ca},
        {reinterpret_cast<const void *>(&k_pr_bwd_CA<true, NW, 0, 2>), ca},
        {reinterpret_cast<const void *>(&k_pr_bwd_CA<true, NW, 2, 2>), ca},
        {reinterpret_cast<const void *>(&k_pr_bwd_CA<false, NW, 0, 0>), ca},
        {reinterpret_cast<const void *>(&k_pr_bwd_CA<false, NW, 0, 2>), ca},
        {reinterpret_cast<const void *>(&k_pr_bwd_CA<false, NW, 2, 2>), ca},
        {reinterpret_cast<const void *>(&k_pr_bwd_grads<NW>), ca},
    };
    for (const auto &k : ks) {
        if (k.lds <= 64 * 1024) continue;
        const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds);
        if (e != hipSuccess) return nf_fail_hip(e, "hipFuncSetAttribute(patch-resident training stage)");
    }
    return NF_OK;
}
int pr_set_attributes(int mode) { return (mode & 3) == 2 ? pr_set_attributes_nw<4>() : pr_set_attributes_nw<8>(); }

void pr_pack_step(nf_trainer *t, hipStream_t st)
{
    PrOffs offs{};
    int n = 0;
    for (int l = 0; l < t->tl.n; ++l)
        if (t->tl.l[l].type == NF_LAYER_COUPLING) {
            offs.off[t->tl.l[l].aux] = t->tl.l[l].off;
            n = std::max(n, t->tl.l[l].aux + 1);
        }
    if (n) hipLaunchKernelGGL(k_pr_pack, dim3((unsigned)n, 13), dim3(256), 0, st, (const float *)t->d_params, offs, t->pr_img);
}

// One workgroup per patch up to one per CU (the stages keep ~100 KB of LDS: one resident workgroup per CU); beyond that every
// workgroup walks its share of the patches, so that set-up, moment finalisation and the slotted sums are paid once per CU.
inline unsigned pr_grid(const nf_trainer *t, const Geo &g)
{
    const int64_t npatch = g.npix / g.HW, per = (npatch + t->n_cu - 1) / t->n_cu;
    return (unsigned)std::min<int64_t>((npatch + per - 1) / per, g.nslot);
}
// the slots a consumer adds up: the producer's grid — or the two slots the cross-rank totals come back in (sync_slots)
inline int pr_nred(const nf_trainer *t, unsigned grid) { return (t->sync_fn && t->sync_world > 1) ? std::max<int>((int)grid, 2) : (int)grid; }

template <int NW>
void pr_coupling_forward(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float *zout, Acc ldacc, const float *zpre,
                         const float *A, hipStream_t st)
{
    constexpr int w = 32;
    const Cpl &c = t->cpl[L.aux];
    const int off_m1 = L.off + 19 * w, off_m2 = L.off + 22 * w + w * w, off_w3 = L.off + 24 * w + w * w;
    const unsigned grid = pr_grid(t, g);
    PrFwdArgs a{};
    a.img = t->pr_img + (size_t)L.aux * PR_SIZE;
    a.bn1 = t->d_flt + c.f_bn1;
    a.bn2 = t->d_flt + c.f_bn2;
    a.n = (double)g.npix * t->sync_world;   // the moments are over the GLOBAL minibatch when the ranks are synchronised
    a.nred = pr_nred(t, grid);
    a.tail3 = t->d_params + off_w3 + 36 * (w + 1);
    a.stats = t->acc(c.d_st1);
    if (t->pr_f0_done) {
        // stage 0 ran in the last launch of the coupling below
    } else if (zpre) {
        a.zsrc = zpre;
        a.A = A;
        a.zmixed = const_cast<float *>(zin);
        hipLaunchKernelGGL((k_pr_fwd<0, true, NW>), dim3(grid), dim3(64 * NW), pr_fwd_lds(0, NW), st, g, a);
    } else {
        a.zsrc = zin;
        hipLaunchKernelGGL((k_pr_fwd<0, false, NW>), dim3(grid), dim3(64 * NW), pr_fwd_lds(0, NW), st, g, a);
    }
    t->pr_f0_done = false;
    sync_slots(t, t->acc(c.d_st1), 2 * w, g.nslot, st);
    a.zsrc = zin;
    a.A = nullptr;
    a.zmixed = nullptr;
    a.stats_in = t->acc(c.d_st1);
    a.run_mean = t->d_params + off_m1;
    a.run_var = t->d_params + off_m1 + w;
    a.stats = t->acc(c.d_st2);
    hipLaunchKernelGGL((k_pr_fwd<1, false, NW>), dim3(grid), dim3(64 * NW), pr_fwd_lds(1, NW), st, g, a);
    sync_slots(t, t->acc(c.d_st2), 2 * w, g.nslot, st);
    a.stats_in = t->acc(c.d_st2);
    a.run_mean = t->d_params + off_m2;
    a.run_var = t->d_params + off_m2 + w;
    a.zout = zout;
    a.u_out = c.u;
    a.ldacc = ldacc;
    if (t->pr_next && (t->pr & 4) == 0) {
        const Cpl &cn = t->cpl[t->pr_next->aux];
        a.next_A = t->pr_next_A;
        a.next_zmixed = t->pr_next_zin;
        a.next_img = t->pr_img + (size_t)t->pr_next->aux * PR_SIZE;
        a.next_stats = t->acc(cn.d_st1);
        hipLaunchKernelGGL((k_pr_fwd<2, false, NW, true>), dim3(grid), dim3(64 * NW), pr_fwd_lds(2, NW), st, g, a);
        t->pr_f0_done = true;
    } else {
        hipLaunchKernelGGL((k_pr_fwd<2, false, NW>), dim3(grid), dim3(64 * NW), pr_fwd_lds(2, NW), st, g, a);
    }
}

// the operands every backward stage of a coupling shares (stage A reads dz / zlat and writes dz2, gu; B and C read gu)
PrBwdArgs pr_bwd_args(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float invB, const float *zlat, unsigned grid)
{
    constexpr int w = 32;
    const Cpl &c = t->cpl[L.aux];
    PrBwdArgs a{};
    a.zin = zin;
    a.img = t->pr_img + (size_t)L.aux * PR_SIZE;
    a.bn1 = t->d_flt + c.f_bn1;
    a.bn2 = t->d_flt + c.f_bn2;
    a.bb2 = t->d_flt + c.f_bb2;
    a.bb1 = t->d_flt + c.f_bb1;
    a.n = (double)g.npix * t->sync_world;
    a.nred = pr_nred(t, grid);
    a.off_w1 = L.off;
    a.off_b1 = L.off + 18 * w;
    a.off_w2 = L.off + 21 * w;
    a.off_w3 = L.off + 24 * w + w * w;
    a.tail3 = t->d_params + a.off_w3 + 36 * (w + 1);
    a.u = c.u;
    a.zlat = zlat;
    a.dz = t->dz;
    a.dz2 = t->dz2;
    a.gu = t->gu[L.aux % 3];   // (by coupling index: the filter-gradient launches of a coupling may still read it on the side stream)
    a.invB = invB;
    a.G = t->acc(0);
    a.bstats = t->acc(c.d_bs2);
    return a;
}

// Small minibatches leave CUs idle (138 patches on 256): d l_last/W — 32 of the 75 matrix instructions per tile of stage A — then runs
// as a launch of its own on the side stream, in the idle CUs (GRAD 1: the CNN recomputed once more up to the product's operands; at
// most as many workgroups as there are idle CUs: a stage keeps ~100 KB of LDS, so two never share a CU), and stage A drops it
// (GRAD 0).  The products of stages B and C stay where they are: offloaded too (measured), the side stream needs the CUs the next
// main launch is waiting for, and every extra launch + event pair costs ~20 us of host time per coupling.
inline bool pr_split(const nf_trainer *t, const Geo &g)
{
    const int64_t npatch = g.npix / g.HW;
    // (its launch walks ceil(patches / idle CUs) patches per workgroup: beyond 3/4 of the CUs busy it would outlast the main launches —
    // 250 patches on 6 workgroups measured 6.7 ms per step; below a quarter the stages are too short to gain)
    return (t->pr & 16) == 0 && !t->serial && 4 * npatch <= 3 * (int64_t)t->n_cu && 4 * npatch >= t->n_cu;
}

template <int NW>
void pr_coupling_backward(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float invB, const float *zmix_in, const float *A,
                          Acc dA, hipStream_t st, const float *zlat)
{
    constexpr int w = 32;
    const Cpl &c = t->cpl[L.aux];
    const unsigned grid = pr_grid(t, g);
    const bool split = pr_split(t, g);
    const int par = L.aux % 3;
    hipStream_t sd = t->serial ? st : t->side;
    auto wait_side = [&](int p) {   // the side work of the coupling that used buffer set p last
        if (t->done_pending[p]) {
            (void)hipStreamWaitEvent(st, t->ev_done[p], 0);
            t->done_pending[p] = false;
        }
    };
    auto fork = [&] {               // the side stream picks up behind the main stream's last launch
        (void)hipEventRecord(t->ev_fork[0], st);
        (void)hipStreamWaitEvent(sd, t->ev_fork[0], 0);
    };
    PrBwdArgs a = pr_bwd_args(t, g, L, zin, invB, zlat, grid);
    if (!t->pr_a_done) {   // (else: stage A rode in the last launch of the coupling above)
        wait_side(par);
#ifdef NF_PR_TIMELINE
        a.dz_out = t->gu[(par + 1) % 3];   // the stamps of stage A (nf_train_pr.h, PR_TL)
#endif
        if (split)
            hipLaunchKernelGGL((k_pr_bwd<0, false, NW, 0>), dim3(grid), dim3(64 * NW), pr_bwd_lds(0, NW), st, g, a);
        else
            hipLaunchKernelGGL((k_pr_bwd<0, false, NW, 2>), dim3(grid), dim3(64 * NW), pr_bwd_lds(0, NW), st, g, a);
#ifdef NF_PR_TIMELINE
        {   // average phase lengths over the workgroups, printed for a few launches (100 MHz counter: 10 ns units)
            static int shown = 0;
            if (shown < 40 && (++shown % 8) == 0) {
                (void)hipStreamSynchronize(st);
                std::vector<long long> h((size_t)grid * 16);
                (void)hipMemcpy(h.data(), t->gu[(par + 1) % 3], h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                static const int ph[6] = {0, 1, 2, 3, 4, 7};   // the stamps the kernel takes
                double d[6] = {0};
                long long t0 = h[0], t1 = h[7];
                for (unsigned b = 0; b < grid; ++b) {
                    for (int i = 1; i < 6; ++i) d[i] += (double)(h[b * 16 + ph[i]] - h[b * 16 + ph[i - 1]]) / grid;
                    t0 = std::min(t0, h[b * 16]);
                    t1 = std::max(t1, h[b * 16 + 7]);
                }
                fprintf(stderr, "stage A timeline (us, mean over %u workgroups): set-up %.2f | gu + tiles %.2f | strip loop %.2f | partials to LDS %.2f | "
                        "sums + stores %.2f | first start to last end %.2f\n", grid, d[1] / 100, d[2] / 100, d[3] / 100, d[4] / 100, d[5] / 100,
                        (double)(t1 - t0) / 100);
            }
        }
#endif
    }
    const bool a_rode = t->pr_a_done;   // stage A ran in the launch of the coupling above: its product was launched behind that one
    t->pr_a_done = false;
    // both products (stage C's and the next stage A's) leave the main launches while every side workgroup has ONE patch — at most half
    // of the CUs busy; beyond (138 patches: 118 side workgroups, 20 of them with two patches) the side launch outlasts the main one
    // (measured: 1.25 against 1.19 ms) and only d l_last/W goes.  NF_TRAIN_PR + 32: d l_last/W only, at every size
    const bool both_ok = (t->pr & 32) == 0 && (t->pr_both_max ? g.npix / g.HW <= t->pr_both_max : 2 * (g.npix / g.HW) <= (int64_t)t->n_cu);
    const unsigned side_grid = !split ? 0u : t->pr_side_full ? grid : std::max(1u, std::min(grid, (unsigned)t->n_cu - grid));
    if (split && !(a_rode && both_ok)) {
        fork();
        hipLaunchKernelGGL((k_pr_bwd<0, false, NW, 1>), dim3(side_grid), dim3(64 * NW), pr_bwd_lds(0, NW), sd, g, a);
        (void)hipEventRecord(t->ev_done[par], sd);
        t->done_pending[par] = true;
    }
    sync_slots(t, t->acc(c.d_bs2), 2 * w, g.nslot, st);
    a.bstats_in = t->acc(c.d_bs2);
    a.bstats = t->acc(c.d_bs1);
    hipLaunchKernelGGL((k_pr_bwd<1, false, NW, 2>), dim3(grid), dim3(64 * NW), pr_bwd_lds(1, NW), st, g, a);
    sync_slots(t, t->acc(c.d_bs1), 2 * w, g.nslot, st);
    a.bstats_in = t->acc(c.d_bs1);
    a.dz_out = t->dz;
    a.dA = dA;
    a.zmix_in = zmix_in;
    a.A = A;
    bool fused = false;
#ifndef NF_PR_TIMELINE
    if (t->pr_below && (t->pr & 8) == 0) {
        // stage A of the coupling below in the same launch (nf_train_pr.h, k_pr_bwd_CA)
        wait_side(t->pr_below->aux % 3);
        const PrBwdArgs an = pr_bwd_args(t, g, *t->pr_below, t->pr_below_zin, invB, nullptr, grid);
        const size_t lds = std::max(pr_bwd_lds(2, NW), pr_bwd_lds(0, NW));
        // split: both stages without their products — d l_1/W of this coupling and d l_last/W of the one below follow in ONE
        // launch on the side stream (NF_TRAIN_PR + 32: stage C keeps its product, stage A's is launched with the coupling below)
        const bool both = split && both_ok;
        if (zmix_in && both) hipLaunchKernelGGL((k_pr_bwd_CA<true, NW, 0, 0>), dim3(grid), dim3(64 * NW), lds, st, g, a, an);
        else if (zmix_in && split) hipLaunchKernelGGL((k_pr_bwd_CA<true, NW, 0, 2>), dim3(grid), dim3(64 * NW), lds, st, g, a, an);
        else if (zmix_in) hipLaunchKernelGGL((k_pr_bwd_CA<true, NW, 2, 2>), dim3(grid), dim3(64 * NW), lds, st, g, a, an);
        else if (both) hipLaunchKernelGGL((k_pr_bwd_CA<false, NW, 0, 0>), dim3(grid), dim3(64 * NW), lds, st, g, a, an);
        else if (split) hipLaunchKernelGGL((k_pr_bwd_CA<false, NW, 0, 2>), dim3(grid), dim3(64 * NW), lds, st, g, a, an);
        else hipLaunchKernelGGL((k_pr_bwd_CA<false, NW, 2, 2>), dim3(grid), dim3(64 * NW), lds, st, g, a, an);
        if (both) {
            const int pb = t->pr_below->aux % 3;
            fork();
            hipLaunchKernelGGL((k_pr_bwd_grads<NW>), dim3(side_grid), dim3(64 * NW), lds, sd, g, a, an);
            (void)hipEventRecord(t->ev_done[par], sd);
            (void)hipEventRecord(t->ev_done[pb], sd);
            t->done_pending[par] = t->done_pending[pb] = true;
        }
        t->pr_a_done = true;
        fused = true;
    }
#endif
    if (!fused) {
        if (zmix_in) hipLaunchKernelGGL((k_pr_bwd<2, true, NW, 2>), dim3(grid), dim3(64 * NW), pr_bwd_lds(2, NW), st, g, a);
        else hipLaunchKernelGGL((k_pr_bwd<2, false, NW, 2>), dim3(grid), dim3(64 * NW), pr_bwd_lds(2, NW), st, g, a);
    }
}

// a coupling of the batch-statistics evaluator (nf_bs_wide_run) on the patch-resident forward stages: in place on z, the moments of
// both normalisations left in mom [4][32], the per-patch log-det share added to t->eldp (NLL direction)
template <int NW>
void pr_coupling_eval(nf_trainer *t, const Geo &g, const TLayer &L, float *z, hipStream_t st, float *mom, bool inverse,
                      const float *mixA = nullptr)   // mixA: the Conv2d1x1 in front of the coupling (NLL direction), applied by its first stage
{
    constexpr int w = 32;
    const Cpl &c = t->cpl[L.aux];
    const unsigned grid = pr_grid(t, g);
    PrFwdArgs a{};
    a.img = t->pr_img + (size_t)L.aux * PR_SIZE;
    a.bn1 = t->d_flt + c.f_bn1;
    a.bn2 = t->d_flt + c.f_bn2;
    a.n = (double)g.npix * t->sync_world;
    a.nred = pr_nred(t, grid);
    a.tail3 = t->d_params + L.off + 24 * w + w * w + 36 * (w + 1);
    a.zsrc = z;
    a.stats = t->acc(c.d_st1);
    if (mixA && !inverse) {   // in place: every lane reads its own pixels and writes them back mixed
        a.A = mixA;
        a.zmixed = z;
        hipLaunchKernelGGL((k_pr_fwd<0, true, NW>), dim3(grid), dim3(64 * NW), pr_fwd_lds(0, NW), st, g, a);
        a.A = nullptr;
        a.zmixed = nullptr;
    } else {
        hipLaunchKernelGGL((k_pr_fwd<0, false, NW>), dim3(grid), dim3(64 * NW), pr_fwd_lds(0, NW), st, g, a);
    }
    sync_slots(t, t->acc(c.d_st1), 2 * w, g.nslot, st);
    a.stats_in = t->acc(c.d_st1);
    a.stats = t->acc(c.d_st2);
    a.mom = mom;
    hipLaunchKernelGGL((k_pr_fwd<1, false, NW>), dim3(grid), dim3(64 * NW), pr_fwd_lds(1, NW), st, g, a);
    sync_slots(t, t->acc(c.d_st2), 2 * w, g.nslot, st);
    a.stats_in = t->acc(c.d_st2);
    a.mom = mom ? mom + 2 * w : nullptr;
    a.zout = z;
    a.ldp = t->eldp;
    if (inverse) a.A = mixA;   // (sampling direction: mixA is the INVERSE matrix of the Conv2d1x1, applied behind the coupling)
    if (inverse)
        hipLaunchKernelGGL((k_pr_fwd<2, false, NW, false, 2>), dim3(grid), dim3(64 * NW), pr_fwd_lds(2, NW), st, g, a);
    else
        hipLaunchKernelGGL((k_pr_fwd<2, false, NW, false, 1>), dim3(grid), dim3(64 * NW), pr_fwd_lds(2, NW), st, g, a);
}

template <int W>
void coupling_forward(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float *zout, Acc ldacc,
                      const float *zpre, const float *A, hipStream_t st)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(t, g.npix);
    const int w = W, off_w1 = L.off, off_m1 = L.off + 19 * w, off_w2 = L.off + 21 * w, off_m2 = L.off + 22 * w + w * w,
              off_w3 = L.off + 24 * w + w * w;
    const double n = (double)g.npix * t->sync_world;   // the moments are over the GLOBAL minibatch when the ranks are synchronised
    // zpre != null: the preceding Conv2d1x1 is folded into l_1 (which then also writes `zin`)
    if (W == 32 && t->pr) {
        if ((t->pr & 3) == 2) pr_coupling_forward<4>(t, g, L, zin, zout, ldacc, zpre, A, st);
        else pr_coupling_forward<8>(t, g, L, zin, zout, ldacc, zpre, A, st);
        return;
    }
    const size_t z_tile = ((size_t)(g.H + 2) * (g.W + 2) * 2 + g.HW) * sizeof(float);
    constexpr bool kWide = W == 16 || W == 32;   // widths with matrix-core stage kernels
    constexpr int WM = kWide ? W : 32;           // (only instantiated for the widths they exist for)
    if (kWide && (t->wide_mfma & 256) && z_tile <= 60 * 1024) {
        const int S = patch_split(g);
        const unsigned ngrid = std::min<unsigned>((unsigned)(g.npix / g.HW) * S, (unsigned)g.nslot);
        if (zpre)
            hipLaunchKernelGGL((k_c1_fwd_mfma<WM, true>), dim3(ngrid), dim3(256), z_tile, st, g, zpre, A, const_cast<float *>(zin),
                               (const float *)t->d_params, off_w1, c.h1, t->acc(c.d_st1), S);
        else
            hipLaunchKernelGGL((k_c1_fwd_mfma<WM, false>), dim3(ngrid), dim3(256), z_tile, st, g, zin, (const float *)nullptr, (float *)nullptr,
                               (const float *)t->d_params, off_w1, c.h1, t->acc(c.d_st1), S);
    } else if (zpre) {
        hipLaunchKernelGGL((k_c1_fwd<W, true>), dim3(nb), dim3(TB), 0, st, g, zpre, A, const_cast<float *>(zin), t->d_params,
                           off_w1, c.h1, t->acc(c.d_st1));
    } else {
        hipLaunchKernelGGL((k_c1_fwd<W, false>), dim3(nb), dim3(TB), 0, st, g, zin, (const float *)nullptr, (float *)nullptr,
                           t->d_params, off_w1, c.h1, t->acc(c.d_st1));
    }
    sync_slots(t, t->acc(c.d_st1), 2 * w, g.nslot, st);
    // wide couplings: the slot sums are added up once, by a finaliser kernel, not by every workgroup of the consumer
    const bool fin = W >= 16 && (t->wide_mfma & 8);
    if (fin)
        hipLaunchKernelGGL(k_bn_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_st1), w, g.nslot, n, t->d_params, off_m1, off_m1 + w,
                           t->d_flt + c.f_bn1);
    if (kWide && (t->wide_mfma & 2))
        hipLaunchKernelGGL(k_c2_fwd_mfma<WM>, dim3(nb), dim3(256), 0, st, g, (const float *)c.h1, t->acc(c.d_st1), n, t->d_params, off_m1,
                           t->d_flt + c.f_bn1, off_w2, c.h2, t->acc(c.d_st2), (const float *)t->d_params, fin);
    else
        hipLaunchKernelGGL(k_c2_fwd<W>, dim3(nb), dim3(TB), 0, st, g, c.h1, t->acc(c.d_st1), n, t->d_params, off_m1,
                           t->d_flt + c.f_bn1, off_w2, c.h2, t->acc(c.d_st2), (const float *)t->d_params, fin);
    sync_slots(t, t->acc(c.d_st2), 2 * w, g.nslot, st);
    if (fin)
        hipLaunchKernelGGL(k_bn_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_st2), w, g.nslot, n, t->d_params, off_m2, off_m2 + w,
                           t->d_flt + c.f_bn2);
    if (kWide && fin && (t->wide_mfma & 16) && 3 * g.W <= t->band_cap) {
        // bands of BR rows: (BR + 2) rows of 36 P values per pixel in LDS
        const int BR = std::max(1, std::min(g.H, t->band_cap / g.W - 2)), units = (int)(g.npix / g.HW) * ((g.H + BR - 1) / BR);
        const size_t lds = (size_t)(((BR + 2) * g.W + 31) / 32 * 32) * 36 * sizeof(float);
        hipLaunchKernelGGL(k_c3_fwd_mfma<WM>, dim3(std::min<unsigned>((unsigned)units, (unsigned)g.nslot)), dim3(256), lds, st, g, zin,
                           (const float *)c.h2, (const float *)(t->d_flt + c.f_bn2), (const float *)t->d_params, off_w3, zout, ldacc,
                           c.u, BR);
    } else {
        hipLaunchKernelGGL(k_c3_fwd<W>, dim3(nb), dim3(TB), 0, st, g, zin, c.h2, t->acc(c.d_st2), n, t->d_params, off_m2,
                           t->d_flt + c.f_bn2, off_w3, zout, ldacc, (const float *)t->d_params, fin, c.u);
    }
}

template <int W>
void coupling_backward(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float invB, const float *zmix_in,
                       const float *A, Acc dA, hipStream_t st, const float *zlat)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(t, g.npix);
    const int w = W, off_w1 = L.off, off_b1 = L.off + 18 * w, off_w2 = L.off + 21 * w, off_w3 = L.off + 24 * w + w * w;
    const double n = (double)g.npix * t->sync_world;
    const float *bn1 = t->d_flt + c.f_bn1, *bn2 = t->d_flt + c.f_bn2;
    const Acc G = t->acc(0);
    if (W == 32 && t->pr) {
        if ((t->pr & 3) == 2) pr_coupling_backward<4>(t, g, L, zin, invB, zmix_in, A, dA, st, zlat);
        else pr_coupling_backward<8>(t, g, L, zin, invB, zmix_in, A, dA, st, zlat);
        return;
    }
    const unsigned ng = std::min(nb, 96u);   // filter-gradient kernels: grid.y multiplies the workgroup count
    // The three filter-gradient kernels only feed the parameter gradient, not d loss / d z: they run
    // on the side stream, forked after their producer, while the main stream walks on.  The
    // temporaries they read are triple-buffered by coupling index; before a buffer set is reused
    // the main stream waits for the side work of the coupling that used it last.
    const int par = L.aux % 3;
    float *t1 = t->t1[par], *t2 = t->t2[par], *gu = t->gu[par];
    hipStream_t sd = t->serial ? st : t->side;
    if (t->done_pending[par]) {
        (void)hipStreamWaitEvent(st, t->ev_done[par], 0);
        t->done_pending[par] = false;
    }
    const size_t gu_tile = ((size_t)(g.H + 2) * (g.W + 2) * 4 + g.HW) * sizeof(float);   // per-patch operand tile + pixel index
    constexpr bool kWide = W == 16 || W == 32;
    constexpr int WM = kWide ? W : 32;
    const bool mfma_dh = kWide && (t->wide_mfma & 32) && gu_tile <= 60 * 1024;
    // the elementwise stage in front of the transposed l_last runs inside it when that is the matrix-core kernel
    const bool c3b_in = mfma_dh && c.u && (t->wide_mfma & 2048);
    const C3Bwd c3b{c3b_in ? c.u : nullptr, zin, zlat, t->dz, t->dz2, gu, invB};
    const float *dz_src = c3b_in ? t->dz2 : nullptr;   // where the coupling's last stage finds d loss / d z (null: in t->dz)
    if (!c3b_in)
        hipLaunchKernelGGL(k_c3_bwd<W>, dim3(nb), dim3(TB), 0, st, g, zin, c.h2, bn2, t->d_params, off_w3, invB, t->dz, gu, G, zlat,
                           (const float *)c.u);
    // The filter gradients of l_last / l_2 inside the stage kernels (3 tensor passes less: 6.0 -> 5.6 ms per step at 1 024
    // patches) — but only when the stages fill the GPU: at 138 patches the side stream's kernels run in the CUs the stage
    // kernels leave idle, and fusing them lengthens the critical path instead (1.67 -> 1.86 ms).
    const bool fuse = (t->wide_mfma & 1) && (t->wide_mfma & 128) && g.npix >= 400 * 1024;
    const bool fuse_w3 = mfma_dh && fuse, fuse_w2 = kWide && (t->wide_mfma & 4) && fuse;
    const int S = patch_split(g);
    const unsigned npw = std::min<unsigned>((unsigned)(g.npix / g.HW) * S, (unsigned)g.nslot);   // grid of the per-patch kernels
    if (mfma_dh && fuse_w3)
        hipLaunchKernelGGL((k_c3_dh_mfma<WM, true>), dim3(npw), dim3(256), gu_tile, st, g,
                           (const float *)c.h2, bn2, (const float *)t->d_params, off_w3, (const float *)gu, t1, t->acc(c.d_bs2), G, S, c3b);
    else if (mfma_dh)
        hipLaunchKernelGGL((k_c3_dh_mfma<WM, false>), dim3(npw), dim3(256), gu_tile, st, g,
                           (const float *)c.h2, bn2, (const float *)t->d_params, off_w3, (const float *)gu, t1, t->acc(c.d_bs2), G, S, c3b);
    else
        hipLaunchKernelGGL(k_c3_dh<W>, dim3(nb), dim3(TB), 0, st, g, c.h2, bn2, t->d_params, off_w3, gu, t1, t->acc(c.d_bs2));
    sync_slots(t, t->acc(c.d_bs2), 2 * w, g.nslot, st);
    const bool fin = W >= 16 && (t->wide_mfma & 8);
    const float *pre2 = fin ? t->d_flt + c.f_bb2 : nullptr, *pre1 = fin ? t->d_flt + c.f_bb1 : nullptr;
    if (fin) hipLaunchKernelGGL(k_bnb_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_bs2), w, g.nslot, n, t->d_flt + c.f_bb2);
    if (fuse_w2)
        hipLaunchKernelGGL((k_c2_bwd_mfma<WM, true>), dim3(nb), dim3(256), 0, st, g, (const float *)c.h1, bn1, (const float *)c.h2, bn2,
                           t->acc(c.d_bs2), n, (const float *)t->d_params, off_w2, t1, t2, t->acc(c.d_bs1), G, pre2);
    else if (kWide && (t->wide_mfma & 4))
        hipLaunchKernelGGL((k_c2_bwd_mfma<WM, false>), dim3(nb), dim3(256), 0, st, g, (const float *)c.h1, bn1, (const float *)c.h2, bn2,
                           t->acc(c.d_bs2), n, (const float *)t->d_params, off_w2, t1, t2, t->acc(c.d_bs1), G, pre2);
    else
        hipLaunchKernelGGL(k_c2_bwd<W>, dim3(nb), dim3(TB), 0, st, g, c.h1, bn1, c.h2, bn2, t->acc(c.d_bs2), n, t->d_params, off_w2,
                           t1, t2, t->acc(c.d_bs1), G, pre2);
    sync_slots(t, t->acc(c.d_bs1), 2 * w, g.nslot, st);
    if (fin) hipLaunchKernelGGL(k_bnb_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_bs1), w, g.nslot, n, t->d_flt + c.f_bb1);
    if (fin && W % 4 == 0 && (256 % (W / 4)) == 0)
        hipLaunchKernelGGL(k_c1_bwd_flat<(W >= 16 ? W : 16)>, dim3(nb), dim3(256), 0, st, g, (const float *)c.h1, bn1, pre1, off_b1, t2, G);
    else
        hipLaunchKernelGGL(k_c1_bwd<W>, dim3(nb), dim3(TB), 0, st, g, c.h1, bn1, t->acc(c.d_bs1), n, off_b1, t2, G, pre1);
    // one fork per coupling (every event operation costs host time): all three producers are done
    (void)hipEventRecord(t->ev_fork[0], st);
    (void)hipStreamWaitEvent(sd, t->ev_fork[0], 0);
    // (the per-patch operand tiles of the matrix-core kernels sit in dynamic LDS: patches of up to ~3 000 pixels)
    if (kWide && (t->wide_mfma & 1) && ((size_t)(g.H + 2) * (g.W + 2) * 4 + g.HW) * sizeof(float) <= 60 * 1024) {
        const unsigned nw = std::min<unsigned>((unsigned)g.nslot, 512u);
        const size_t tile = (size_t)(g.H + 2) * (g.W + 2) * 4 * sizeof(float), lut = (size_t)g.HW * sizeof(int);
        if (!fuse_w3)
            hipLaunchKernelGGL(k_w3_grad_mfma<WM>, dim3(npw), dim3(256), tile + lut, sd, g, c.h2, bn2, (const float *)gu, off_w3, G, S);
        if (!fuse_w2) hipLaunchKernelGGL(k_w2_grad_mfma<WM>, dim3(nw), dim3(256), 0, sd, g, c.h1, bn1, (const float *)t1, off_w2, G);
        hipLaunchKernelGGL(k_w1_grad_mfma<WM>, dim3(npw), dim3(256), tile / 2 + lut, sd, g, zin, (const float *)t2, off_w1, G, S);
    } else {
        hipLaunchKernelGGL(k_w3_grad<W>, dim3(ng, 9), dim3(TB), 0, sd, g, c.h2, bn2, gu, off_w3, G);
        hipLaunchKernelGGL(k_w2_grad<W>, dim3(ng, w), dim3(TB), 0, sd, g, c.h1, bn1, t1, off_w2, G);
        hipLaunchKernelGGL(k_w1_grad<W>, dim3(ng, 9), dim3(TB), 0, sd, g, zin, t2, off_w1, G);
    }
    (void)hipEventRecord(t->ev_done[par], sd);
    t->done_pending[par] = true;
    // zmix_in != null: the backward of the preceding Conv2d1x1 is folded into this last stage
    if (kWide && (t->wide_mfma & 64) && 3 * g.W <= t->band_cap) {
        // (measured at 32x32, width 32: 46 us with 6-row bands = 8 tiles = two rounds of the 4 wavefronts, 55 us with 8-row bands;
        // the l_last forward, twice the MFMA work per tile, is better off with the smaller halo share of 8-row bands)
        const int cap = std::max(3 * g.W, std::min(t->band_cap, 256));
        const int BR = std::max(1, std::min(g.H, cap / g.W - 2)), units = (int)(g.npix / g.HW) * ((g.H + BR - 1) / BR);
        const size_t lds = (size_t)(((BR + 2) * g.W + 31) / 32 * 32) * 20 * sizeof(float);
        const unsigned ngrid = std::min<unsigned>((unsigned)units, (unsigned)g.nslot);
        if (zmix_in)
            hipLaunchKernelGGL((k_c1_dz_mfma<WM, true>), dim3(ngrid), dim3(256), lds, st, g, (const float *)t2, (const float *)t->d_params,
                               off_w1, t->dz, zmix_in, A, dA, BR, dz_src);
        else
            hipLaunchKernelGGL((k_c1_dz_mfma<WM, false>), dim3(ngrid), dim3(256), lds, st, g, (const float *)t2, (const float *)t->d_params,
                               off_w1, t->dz, (const float *)nullptr, (const float *)nullptr, dA, BR, dz_src);
    } else if (zmix_in) {
        hipLaunchKernelGGL((k_c1_dz<W, true>), dim3(nb), dim3(TB), 0, st, g, t2, t->d_params, off_w1, t->dz, zmix_in, A, dA, dz_src);
    } else {
        hipLaunchKernelGGL((k_c1_dz<W, false>), dim3(nb), dim3(TB), 0, st, g, t2, t->d_params, off_w1, t->dz,
                           (const float *)nullptr, (const float *)nullptr, dA, dz_src);
    }
}

// The tiled form of coupling_forward (k_tiled_fwd): `f1_done` = this coupling's stage 1 already ran in the launch that
// finished the coupling below it; `nxt` = the coupling above (index `ln`, behind a Conv2d1x1 when nxt_A != null) whose stage 1
// shares this coupling's last launch.
template <int W, int NT>
void coupling_forward_tiled(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float *zout, Acc ldacc,
                            const float *zpre, const float *A, hipStream_t st, bool f1_done, const TLayer *nxt, const float *nxt_A,
                            float *nxt_zin)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(t, g.npix), npatch = (unsigned)(g.npix / g.HW);
    const int w = W, off_w1 = L.off, off_m1 = L.off + 19 * w, off_w2 = L.off + 21 * w, off_m2 = L.off + 22 * w + w * w,
              off_w3 = L.off + 24 * w + w * w;
    const double n = (double)g.npix * t->sync_world;
    const size_t smem = ((size_t)(g.H + 2) * (g.W + 2) * (W + 2) + (NT / 16) * (1 + 2 * W)) * sizeof(float);
    if (!f1_done) {
        const TiledF1 f1{A, const_cast<float *>(zin), off_w1, c.h1, t->acc(c.d_st1)};
        if (zpre)
            hipLaunchKernelGGL((k_tiled_fwd<W, NT, false, true, true>), dim3(npatch), dim3(NT), smem, st, g, TiledF3{}, f1, zpre, n, t->d_params, (const float *)t->d_params);
        else
            hipLaunchKernelGGL((k_tiled_fwd<W, NT, false, false, true>), dim3(npatch), dim3(NT), smem, st, g, TiledF3{}, f1, zin, n, t->d_params, (const float *)t->d_params);
    }
    sync_slots(t, t->acc(c.d_st1), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_c2_fwd<W>, dim3(nb), dim3(TB), 0, st, g, c.h1, t->acc(c.d_st1), n, t->d_params, off_m1,
                       t->d_flt + c.f_bn1, off_w2, c.h2, t->acc(c.d_st2), (const float *)t->d_params, false);
    sync_slots(t, t->acc(c.d_st2), 2 * w, g.nslot, st);
    const TiledF3 f3{zin, c.h2, t->acc(c.d_st2), off_m2, off_w3, t->d_flt + c.f_bn2, zout, ldacc};
    if (nxt) {
        const Cpl &cn = t->cpl[nxt->aux];
        const TiledF1 f1{nxt_A, nxt_zin, nxt->off, cn.h1, t->acc(cn.d_st1)};
        if (nxt_A)
            hipLaunchKernelGGL((k_tiled_fwd<W, NT, true, true, true>), dim3(npatch), dim3(NT), smem, st, g, f3, f1, (const float *)nullptr, n, t->d_params, (const float *)t->d_params);
        else
            hipLaunchKernelGGL((k_tiled_fwd<W, NT, true, false, true>), dim3(npatch), dim3(NT), smem, st, g, f3, f1, (const float *)nullptr, n, t->d_params, (const float *)t->d_params);
    } else {
        hipLaunchKernelGGL((k_tiled_fwd<W, NT, true, false, false>), dim3(npatch), dim3(NT), smem, st, g, f3, TiledF1{}, (const float *)nullptr, n, t->d_params, (const float *)t->d_params);
    }
}

// The tiled form of coupling_backward (k_tiled_CA above): `a_done` = this coupling's stage A' already ran in the launch
// that finished the coupling above it; `nxt` = the coupling below whose A' shares this coupling's last launch (or null).
// The temporaries the side stream reads are triple-buffered by coupling index: A' of the coupling below is written while
// the filter-gradient kernels of this one and of the one above may still be running.
template <int W, int NT>
void coupling_backward_tiled(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float invB, const float *zmix_in,
                             const float *A, Acc dA, hipStream_t st, const float *zlat, bool a_done, const TLayer *nxt,
                             const float *nxt_zin)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(t, g.npix), npatch = (unsigned)(g.npix / g.HW);
    const int w = W, off_w1 = L.off, off_w2 = L.off + 21 * w, off_w3 = L.off + 24 * w + w * w;
    const double n = (double)g.npix * t->sync_world;
    const float *bn1 = t->d_flt + c.f_bn1, *bn2 = t->d_flt + c.f_bn2;
    const Acc G = t->acc(0);
    const unsigned ng = std::min(nb, 96u);
    const int set = L.aux % 3;
    float *t1 = t->t1[set], *t2 = t->t2[set], *gu = t->gu[set];
    hipStream_t sd = t->serial ? st : t->side;
    const size_t smem = ((size_t)(g.H + 2) * (g.W + 2) * (W + 4) + (NT / 16) * (W + 16 + 2 * W + 9)) * sizeof(float);
    auto wait_set = [&](int k) {
        if (t->done_pending[k]) {
            (void)hipStreamWaitEvent(st, t->ev_done[k], 0);
            t->done_pending[k] = false;
        }
    };
    const TiledA me{zin, c.h2, bn2, off_w3, gu, t1, t->acc(c.d_bs2)};
    if (!a_done) {
        wait_set(set);
        hipLaunchKernelGGL((k_tiled_CA<W, NT, false, false, true>), dim3(npatch), dim3(NT), smem, st, g, TiledC{}, me, n,
                           (const float *)t->d_params, invB, t->dz, zlat, G);
    }
    sync_slots(t, t->acc(c.d_bs2), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_c2_bwd<W>, dim3(nb), dim3(TB), 0, st, g, c.h1, bn1, c.h2, bn2, t->acc(c.d_bs2), n, t->d_params, off_w2,
                       t1, t2, t->acc(c.d_bs1), G, (const float *)nullptr);
    sync_slots(t, t->acc(c.d_bs1), 2 * w, g.nslot, st);
    const TiledC cc{zmix_in, A, c.h1, bn1, t2, off_w1, t->acc(c.d_bs1), dA};
    TiledA below{};
    if (nxt) {
        const Cpl &cn = t->cpl[nxt->aux];
        const int ns = nxt->aux % 3;
        wait_set(ns);
        below = TiledA{nxt_zin, cn.h2, t->d_flt + cn.f_bn2, nxt->off + 24 * w + w * w, t->gu[ns], t->t1[ns], t->acc(cn.d_bs2)};
    }
#define NF_TILED(MIX, NEXT)                                                                                                        \
    hipLaunchKernelGGL((k_tiled_CA<W, NT, true, MIX, NEXT>), dim3(npatch), dim3(NT), smem, st, g, cc, below, n,                    \
                       (const float *)t->d_params, invB, t->dz, (const float *)nullptr, G)
    if (zmix_in) {
        if (nxt) NF_TILED(true, true); else NF_TILED(true, false);
    } else {
        if (nxt) NF_TILED(false, true); else NF_TILED(false, false);
    }
#undef NF_TILED
    (void)hipEventRecord(t->ev_fork[0], st);
    (void)hipStreamWaitEvent(sd, t->ev_fork[0], 0);
    hipLaunchKernelGGL(k_w3_grad<W>, dim3(ng, 9), dim3(TB), 0, sd, g, c.h2, bn2, gu, off_w3, G);
    hipLaunchKernelGGL(k_w2_grad<W>, dim3(ng, w), dim3(TB), 0, sd, g, c.h1, bn1, t1, off_w2, G);
    hipLaunchKernelGGL(k_w1_grad<W>, dim3(ng, 9), dim3(TB), 0, sd, g, zin, t2, off_w1, G);
    (void)hipEventRecord(t->ev_done[set], sd);
    t->done_pending[set] = true;
}

// the tiled stages take one thread per pixel of a patch and one accumulator slot per patch
bool tiled_ok(const nf_trainer *t, const Geo &g, int width, int pass /* 1 backward, 2 forward */)
{
    // measured on the shipped model (both passes tiled): 0.61 vs 0.77 ms per step at 138 patches, 0.77 vs 0.88 at 256, break-even
    // near 400 (one 1024-thread workgroup per CU: past one round over the 256 CUs the layer kernels' finer grain wins)
    const int64_t npatch = g.npix / g.HW;
    // (NF_TRAIN_GEMM=1 sends widths 4 / 8 down the GEMM path: the create-time allocation and this dispatch share gemm_width())
    return (t->tiled & pass) && !(t->all_gemm || gemm_width(width)) && (width == 4 || width == 8) && g.HW <= 1024 && npatch <= g.nslot && npatch <= 384;
}

#define NF_WIDTH_SWITCH(w, CALL)            \
    switch (w) {                            \
    case 4: CALL(4); break;                 \
    case 8: CALL(8); break;                 \
    case 16: CALL(16); break;               \
    case 32: CALL(32); break;               \
    default: break;                         \
    }

int cond_index(const nf_cond *cond, bool needed, bool needs_cam, CondIdx &ci)
{
    ci.iso = 0.f;
    ci.iso_idx = -1;
    ci.cam_idx = 0;
    if (!needed) return NF_OK;
    if (!cond) return nf_fail(NF_EINVAL, "model has a signal-dependent layer but cond is NULL");
    static const float iso_vals[5] = {100.f, 400.f, 800.f, 1600.f, 3200.f};
    ci.iso = cond->iso;
    for (int i = 0; i < 5; ++i)
        if (iso_vals[i] == cond->iso) ci.iso_idx = i;
    int cam = -1;
    for (int i = 0; i < 5; ++i)
        if ((float)i == cond->cam) cam = i;
    if (cam < 0 && !needs_cam) cam = 0;
    if (cam < 0) return nf_fail(NF_ECOND, "unknown camera id %g (expected 0..4 = IP,GP,S6,N6,G4)", (double)cond->cam);
    ci.cam_idx = cam;
    return NF_OK;
}

}  // namespace

extern "C" {

int nf_trainer_destroy(nf_trainer *t)
{
    if (!t) return NF_OK;
    Guard guard;
    (void)guard.enter(t->device);
    if (t->side) {
        (void)hipStreamSynchronize(t->side);
        (void)hipStreamDestroy(t->side);
    }
    for (hipEvent_t ev : t->ev_fork)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : t->ev_done)
        if (ev) (void)hipEventDestroy(ev);
    for (void *p : t->owned) (void)hipFree(p);
    delete t;
    return NF_OK;
}

static int trainer_create_impl(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params,
                               int64_t max_batch, int32_t optimizer, bool eval_only, nf_trainer **out);

int nf_trainer_create(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params,
                      int64_t max_batch, int32_t optimizer, nf_trainer **out)
{
    return trainer_create_impl(cfg, layers, params, n_params, max_batch, optimizer, false, out);
}

static int trainer_create_impl(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params,
                               int64_t max_batch, int32_t optimizer, bool eval_only, nf_trainer **out)
{
    if (!out) return nf_fail(NF_EINVAL, "out is NULL");
    *out = nullptr;
    if (!cfg || !layers || !params) return nf_fail(NF_EINVAL, "null argument");
    if (cfg->channels != 4) return nf_fail(NF_EINVAL, "channels must be 4 (packed raw), got %d", cfg->channels);
    if (cfg->height < 1 || cfg->width < 1) return nf_fail(NF_EINVAL, "bad patch size %dx%d", cfg->height, cfg->width);
    if (cfg->flags != 0) return nf_fail(NF_EINVAL, "training is fp32: nf_config.flags must be 0");
    if (cfg->n_layers < 1 || cfg->n_layers > kMaxLayers) return nf_fail(NF_EINVAL, "n_layers must be in 1..%d", kMaxLayers);
    if (max_batch < 1) return nf_fail(NF_EINVAL, "max_batch must be >= 1");
    if (optimizer != NF_OPT_ADAM && optimizer != NF_OPT_MOMENTUM) return nf_fail(NF_EINVAL, "unknown optimizer %d", optimizer);
    if (n_params > (size_t)1 << 24) return nf_fail(NF_EINVAL, "n_params too large");

    nf_trainer *t = new (std::nothrow) nf_trainer();
    if (!t) return nf_fail(NF_ENOMEM, "out of host memory");
    t->cfg = *cfg;
    t->eval_only = t->all_gemm = eval_only;
    if (const char *e = getenv("NF_TRAIN_TILED")) t->tiled = atoi(e);
    if (const char *e = getenv("NF_TRAIN_WIDE_MFMA")) t->wide_mfma = atoi(e);
    if (const char *e = getenv("NF_TRAIN_SERIAL")) t->serial = atoi(e) != 0;
    if (const char *e = getenv("NF_TRAIN_PR")) t->pr = atoi(e);
    if (const char *e = getenv("NF_TRAIN_BAND")) t->band_cap = std::min(320, std::max(96, atoi(e)));
    t->max_batch = max_batch;
    t->optimizer = optimizer;
    t->n_params = (int)n_params;
    t->layers.assign(layers, layers + cfg->n_layers);
    std::vector<uint8_t> mask(n_params, 0);
    int n_mix = 0, n_cpl = 0, n_sdn = 0, n_gain = 0;
    memset(&t->tl, 0, sizeof(t->tl));
    t->tl.n = cfg->n_layers;
    int64_t prev_end = 0;
    for (int i = 0; i < cfg->n_layers; ++i) {
        const nf_layer_desc &L = layers[i];
        const int64_t cnt = nf_layer_param_count(L.type, L.width);
        if (cnt < 0 || L.param_offset < 0 || (uint64_t)L.param_offset + (uint64_t)cnt > n_params) {
            delete t;
            return nf_fail(NF_EINVAL, "layer %d: bad type / width / parameter range", i);
        }
        // the slotted-sum storage is indexed by parameter position with the l_2/W bodies cut out (`holes`, `acc()`): that
        // bookkeeping — and the one-gradient-per-variable contract — needs the layers' parameter blocks in ascending order, disjoint
        if (L.param_offset < prev_end) {
            delete t;
            return nf_fail(NF_EINVAL, "layer %d: parameter blocks must be ascending and must not overlap (offset %lld, previous block ends at %lld)",
                           i, (long long)L.param_offset, (long long)prev_end);
        }
        prev_end = L.param_offset + cnt;
        TLayer &T = t->tl.l[i];
        T.type = L.type;
        T.kind = L.type;
        T.width = L.width;
        T.off = (int)L.param_offset;
        uint8_t *mk = mask.data() + L.param_offset;
        switch (L.type) {
        case NF_LAYER_CONV1X1:
            T.aux = n_mix++;
            for (int k = 20; k < 36; ++k) mk[k] = 1;   // log_S, L_vec, U_vec (P, sign_S are constants)
            break;
        case NF_LAYER_COUPLING: {
            const int w = L.width;
            if (w < 1 || w > 512) {
                delete t;
                return nf_fail(NF_EINVAL, "layer %d: the trainer takes the coupling widths 1 .. 512 (%d given)", i, w);
            }
            if (t->width && t->width != w) {
                delete t;
                return nf_fail(NF_EINVAL, "all coupling layers must share one width");
            }
            t->width = w;
            T.aux = n_cpl++;
            for (int64_t k = 0; k < cnt; ++k) mk[k] = 1;
            for (int k = 0; k < 2 * w; ++k) mk[19 * w + k] = mk[22 * w + w * w + k] = 0;   // BN running statistics
            break;
        }
        case NF_LAYER_SDN5:
            T.aux = n_sdn++;
            for (int k = 0; k < 22; ++k) mk[k] = 1;    // c_i is a constant
            t->has_sdn = true;
            t->needs_cam = true;
            break;
        case NF_LAYER_SDN4:
            T.type = NF_LAYER_SDN5;
            T.aux = n_sdn++;
            for (int k = 0; k < 7; ++k) mk[k] = 1;
            t->has_sdn = true;
            break;
        case NF_LAYER_GAIN4:
            T.aux = n_gain++;
            mk[0] = 1;
            break;
        // the other parameterisations run on the same kernels (TLayer::type) and differ in k_prep / k_finish (TLayer::kind)
        case NF_LAYER_CONV1X1_NONE:
            T.type = NF_LAYER_CONV1X1;
            T.aux = n_mix++;
            for (int k = 0; k < 16; ++k) mk[k] = 1;
            break;
        case NF_LAYER_CONV1X1_LU2:
            T.type = NF_LAYER_CONV1X1;
            T.aux = n_mix++;
            for (int k = 16; k < 32; ++k) mk[k] = 1;   // L (P, sign_S are constants)
            for (int k = 36; k < 56; ++k) mk[k] = 1;   // log_S, U
            break;
        case NF_LAYER_PERMUTE:
            T.type = NF_LAYER_CONV1X1;
            T.aux = n_mix++;
            break;
        case NF_LAYER_SDN:
        case NF_LAYER_SDN1:
        case NF_LAYER_SDN2:
        case NF_LAYER_SDN3:
        case NF_LAYER_SDN6:
            T.type = NF_LAYER_SDN5;
            T.aux = n_sdn++;
            for (int64_t k = 0; k < (L.type == NF_LAYER_SDN6 ? 12 : cnt); ++k) mk[k] = 1;   // SDN6: c_i is a constant
            t->has_sdn = true;
            if (L.type == NF_LAYER_SDN6) t->needs_cam = true;
            break;
        case NF_LAYER_GAIN:
        case NF_LAYER_GAIN1:
        case NF_LAYER_GAIN2:
        case NF_LAYER_GAIN3:
            T.type = NF_LAYER_GAIN4;
            T.aux = n_gain++;
            for (int64_t k = 0; k < cnt; ++k) mk[k] = 1;
            t->needs_cond = true;
            break;
        default:
            delete t;
            return nf_fail(NF_EINVAL, "layer %d: unknown layer type %d", i, L.type);
        }
    }

    hipError_t e;
    if (cfg->device >= 0) {
        t->device = cfg->device;
    } else if ((e = hipGetDevice(&t->device)) != hipSuccess) {
        delete t;
        return nf_fail_hip(e, "hipGetDevice");
    }
    Guard guard;
    int rc = guard.enter(t->device);
    if (rc != NF_OK) {
        delete t;
        return rc;
    }

    const int w = t->width ? t->width : 4;
    const size_t act = (size_t)max_batch * cfg->height * cfg->width;   // pixels
    t->n_mix = n_mix;
    const bool gemm_path = t->all_gemm || gemm_width(w);
    // width 32 on 32x32 patches: the patch-resident stages (nf_train_pr.h) — no [pixel][32] tensor exists, none is allocated
    // (the batch-statistics evaluator, a trimmed trainer otherwise on the GEMM path, takes its forward stages too)
    t->pr = (t->pr && t->wide_mfma != 0 && w == 32 && (!gemm_path || eval_only) && cfg->height == 32 && cfg->width == 32 && n_cpl > 0) ? t->pr : 0;
    // double workspace
    size_t nd = eval_only ? 0 : n_params;
    t->d_dA = (int)nd; nd += eval_only ? 0 : 16 * (size_t)n_mix;
    t->d_dab = (int)nd; nd += eval_only ? 0 : 2 * (size_t)n_sdn;
    t->d_dg = (int)nd; nd += eval_only ? 0 : (size_t)n_gain;
    t->d_ld0 = (int)nd; nd += eval_only ? 0 : (size_t)cfg->n_layers;   // batch total of each layer's data-dependent log-det
    t->cpl.resize(n_cpl);
    for (Cpl &c : t->cpl) {
        if (eval_only) {            // one coupling at a time: every coupling's batch sums share the same slots
            c.d_st1 = 0;
            c.d_st2 = 2 * w;
            c.d_bs1 = c.d_bs2 = 0;
            continue;
        }
        c.d_st1 = (int)nd; nd += 2 * w;
        c.d_st2 = (int)nd; nd += 2 * w;
        c.d_bs1 = (int)nd; nd += 2 * w;
        c.d_bs2 = (int)nd; nd += 2 * w;
    }
    if (eval_only) nd = 4 * (size_t)w;
    t->d_ldc = (int)nd; nd += 1;   // last: the only value accumulated directly (k_prep), not through slots
    t->n_dbl = nd;
    if (gemm_path && !eval_only)
        for (int l = 0; l < t->tl.n; ++l) {
            const TLayer &L = t->tl.l[l];
            if (L.type != NF_LAYER_COUPLING) continue;
            t->holes.emplace_back(L.off + 21 * L.width, L.width * L.width);
            t->hole_rows += (size_t)L.width * L.width;
        }
    // float scalars
    size_t nf = 0;
    t->f_A = (int)nf; nf += 16 * (size_t)n_mix;
    t->f_ab = (int)nf; nf += 2 * (size_t)n_sdn;
    t->f_s = (int)nf; nf += (size_t)n_gain;
    for (Cpl &c : t->cpl) {
        c.f_bn1 = (int)nf; nf += 2 * w;
        c.f_bn2 = (int)nf; nf += 2 * w;
        c.f_bb1 = (int)nf; nf += 2 * w;
        c.f_bb2 = (int)nf; nf += 2 * w;
    }
    t->n_flt = nf;

#define NF_TRY(x)                 \
    if ((rc = (x)) != NF_OK) {    \
        nf_trainer_destroy(t);    \
        return rc;                \
    }
    NF_TRY(dev_alloc(t, (void **)&t->d_params, n_params * sizeof(float)));
    NF_TRY(dev_alloc(t, (void **)&t->d_dbl, nd * sizeof(double)));
    NF_TRY(dev_alloc(t, (void **)&t->d_part, t->part_floats() * sizeof(float)));
    NF_TRY(dev_alloc(t, (void **)&t->d_flt, nf * sizeof(float)));
    if (eval_only) {
        NF_TRY(dev_alloc(t, (void **)&t->ebuf, act * 4 * sizeof(float)));
        NF_TRY(dev_alloc(t, (void **)&t->eldp, (size_t)max_batch * sizeof(double)));
        NF_TRY(dev_alloc(t, (void **)&t->emom, (size_t)std::max(n_cpl, 1) * 4 * w * sizeof(float)));
        NF_TRY(dev_alloc(t, (void **)&t->eAinv, (size_t)std::max(n_mix, 1) * 16 * sizeof(float)));
        if (t->pr) {   // width 32 on 32x32 patches: the patch-resident forward stages need their packed weights and nothing else
            NF_TRY(dev_alloc(t, (void **)&t->pr_img, (size_t)n_cpl * PR_SIZE * sizeof(float)));
            if ((rc = pr_set_attributes(t->pr)) != NF_OK) {
                nf_trainer_destroy(t);
                return rc;
            }
        } else if (n_cpl > 0) {
            float *h1 = nullptr, *h2 = nullptr;
            NF_TRY(dev_alloc(t, (void **)&h1, act * w * sizeof(float)));
            NF_TRY(dev_alloc(t, (void **)&h2, act * w * sizeof(float)));
            for (Cpl &c : t->cpl) {
                c.h1 = h1;
                c.h2 = h2;
            }
            NF_TRY(dev_alloc(t, (void **)&t->gz18, act * kZ18 * sizeof(float)));
            NF_TRY(dev_alloc(t, (void **)&t->gp36, act * 36 * sizeof(float)));
            NF_TRY(dev_alloc(t, (void **)&t->gpack, gemm_pack_floats(w) * sizeof(float)));
        }
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) == hipSuccess && cus > 0) t->n_cu = cus;
        if ((e = hipMemcpy(t->d_params, params, n_params * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemset(t->d_part, 0, t->part_floats() * sizeof(float))) != hipSuccess) {
            nf_trainer_destroy(t);
            return nf_fail_hip(e, "evaluator initialisation");
        }
        *out = t;
        return NF_OK;
    }
    NF_TRY(dev_alloc(t, (void **)&t->d_m, n_params * sizeof(float)));
    NF_TRY(dev_alloc(t, (void **)&t->d_v, n_params * sizeof(float)));
    NF_TRY(dev_alloc(t, (void **)&t->d_gradf, n_params * sizeof(float)));
    NF_TRY(dev_alloc(t, (void **)&t->d_mask, n_params));
    NF_TRY(dev_alloc(t, (void **)&t->d_patch, 2 * (size_t)max_batch * sizeof(float)));
    NF_TRY(dev_alloc(t, (void **)&t->d_wpart, 2 * ((act + 63) / 64) * sizeof(float)));
    t->zs.assign(cfg->n_layers + 1, nullptr);
    for (int i = 1; i <= cfg->n_layers; ++i) NF_TRY(dev_alloc(t, (void **)&t->zs[i], act * 4 * sizeof(float)));
    for (Cpl &c : t->cpl) {
        if (!t->pr) {
            NF_TRY(dev_alloc(t, (void **)&c.h1, act * w * sizeof(float)));
            NF_TRY(dev_alloc(t, (void **)&c.h2, act * w * sizeof(float)));
        }
        if (w >= 16 || gemm_path) NF_TRY(dev_alloc(t, (void **)&c.u, act * 4 * sizeof(float)));
    }
    if (gemm_path && n_cpl > 0) {   // nf_train_gemm.h
        t->gz18_stride = act * kZ18;
        NF_TRY(dev_alloc(t, (void **)&t->gz18, (size_t)n_cpl * t->gz18_stride * sizeof(float)));
        NF_TRY(dev_alloc(t, (void **)&t->gp36, act * 36 * sizeof(float)));
        NF_TRY(dev_alloc(t, (void **)&t->gq18, act * 18 * sizeof(float)));
        NF_TRY(dev_alloc(t, (void **)&t->gpack, (size_t)n_cpl * gemm_pack_floats(w) * sizeof(float)));
        NF_TRY(dev_alloc(t, (void **)&t->gdw, (size_t)n_cpl * 3 * gemm_part_floats(w) * sizeof(float)));
        if (const char *ev = getenv("NF_TRAIN_GEMM_C1")) t->gemm_c1_fused = atoi(ev) != 0;
        t->nb_floor = w <= 128 ? 512 : 0;   // measured at 138 patches: width 64 3.35 -> 3.21 ms, 128 5.37 -> 4.93; 256 / 512 + 1 % with it
        if (const char *ev = getenv("NF_TRAIN_NB_FLOOR")) t->nb_floor = atoi(ev);   // A/B aid
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) == hipSuccess && cus > 0) t->n_cu = cus;
    }
    for (int k = 0; k < (gemm_path ? 1 : 3); ++k) {
        if (!t->pr) {
            NF_TRY(dev_alloc(t, (void **)&t->t1[k], act * w * sizeof(float)));
            NF_TRY(dev_alloc(t, (void **)&t->t2[k], act * w * sizeof(float)));
        }
        NF_TRY(dev_alloc(t, (void **)&t->gu[k], act * 4 * sizeof(float)));
    }
    NF_TRY(dev_alloc(t, (void **)&t->dz, act * 4 * sizeof(float)));
    if (w >= 16) NF_TRY(dev_alloc(t, (void **)&t->dz2, act * 4 * sizeof(float)));
    if (t->pr) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) == hipSuccess && cus > 0) t->n_cu = cus;
        if (const char *ev = getenv("NF_TRAIN_PR_GRID")) t->n_cu = std::max(1, atoi(ev));   // A/B aid: workgroups of the patch-resident stages
        if (const char *ev = getenv("NF_TRAIN_PR_SIDE_FULL")) t->pr_side_full = atoi(ev);
        if (const char *ev = getenv("NF_TRAIN_PR_BOTH_MAX")) t->pr_both_max = atoi(ev);
        NF_TRY(dev_alloc(t, (void **)&t->pr_img, (size_t)n_cpl * PR_SIZE * sizeof(float)));
        if ((rc = pr_set_attributes(t->pr)) != NF_OK) {
            nf_trainer_destroy(t);
            return rc;
        }
    }
#undef NF_TRY
    if ((e = hipMemcpy(t->d_params, params, n_params * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(t->d_mask, mask.data(), n_params, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(t->d_m, 0, n_params * sizeof(float))) != hipSuccess ||
        (e = hipMemset(t->d_v, 0, n_params * sizeof(float))) != hipSuccess ||
        (e = hipMemset(t->d_gradf, 0, n_params * sizeof(float))) != hipSuccess ||
        // slots of values no kernel ever writes (constants, the gain parameters of other ISOs) stay zero
        (e = hipMemset(t->d_part, 0, t->part_floats() * sizeof(float))) != hipSuccess) {
        nf_trainer_destroy(t);
        return nf_fail_hip(e, "trainer initialisation");
    }
    if ((e = hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking)) != hipSuccess) {
        nf_trainer_destroy(t);
        return nf_fail_hip(e, "hipStreamCreate(trainer side stream)");
    }
    for (int k = 0; k < 6; ++k) {
        hipEvent_t *ev = k < 3 ? &t->ev_fork[k] : &t->ev_done[k - 3];
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) {
            nf_trainer_destroy(t);
            return nf_fail_hip(e, "hipEventCreate(trainer)");
        }
    }
    *out = t;
    return NF_OK;
}

static int trainer_run(nf_trainer *t, const float *x, const float *y, int64_t B, const nf_cond *cond, float *grads_out,
                       float *loss_out, void *stream, bool backward);

int nf_trainer_forward_backward(nf_trainer *t, const float *x, const float *y, int64_t B, const nf_cond *cond,
                                float *grads_out, float *loss_out, void *stream)
{
    return trainer_run(t, x, y, B, cond, grads_out, loss_out, stream, true);
}

int nf_trainer_forward(nf_trainer *t, const float *x, const float *y, int64_t B, const nf_cond *cond, float *loss_out,
                       void *stream)
{
    return trainer_run(t, x, y, B, cond, nullptr, loss_out, stream, false);
}

static int trainer_run(nf_trainer *t, const float *x, const float *y, int64_t B, const nf_cond *cond, float *grads_out,
                       float *loss_out, void *stream, bool backward)
{
    if (!t) return nf_fail(NF_EINVAL, "trainer is NULL");
    if (B < 1 || B > t->max_batch) return nf_fail(NF_EINVAL, "B must be in 1..max_batch (%lld)", (long long)t->max_batch);
    if (!x) return nf_fail(NF_EINVAL, "x is NULL");
    if (t->has_sdn && !y) return nf_fail(NF_EINVAL, "model has a signal-dependent layer but y is NULL");
    CondIdx ci;
    int rc = cond_index(cond, t->has_sdn || t->needs_cond, t->needs_cam, ci);
    if (rc != NF_OK) return rc;
    Guard guard;
    if ((rc = guard.enter(t->device)) != NF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;

    Geo g;
    g.B = (int)B;
    g.H = t->cfg.height;
    g.W = t->cfg.width;
    g.HW = g.H * g.W;
    g.npix = B * (int64_t)g.HW;
    g.nloop = (g.npix + 63) & ~(int64_t)63;
    const unsigned nb = blocks_for(t, g.npix);
    g.nslot = (t->sync_fn && t->sync_world > 1) ? std::max((int)nb, 2) : (int)nb;   // the synchronised totals occupy slots 0 and 1
    t->sync_rc = 0;
    bool mm_failed = false;
    if ((t->all_gemm || gemm_width(t->width ? t->width : 4)) && !t->cpl.empty()) gemm_pack_step(t, st);
    if (t->pr) pr_pack_step(t, st);
    t->pr_f0_done = t->pr_a_done = false;
    t->pr_next = t->pr_below = nullptr;
    const float invB = 1.0f / (float)B;
    const int n = t->cfg.n_layers;
    hipError_t e;
    float *s1 = t->d_patch, *s2 = s1 + t->max_batch;
    double *G = t->d_dbl;

    hipLaunchKernelGGL(k_prep, dim3(1), dim3(64), 0, st, t->tl, t->d_params, ci, g.HW, t->d_flt + t->f_A, t->d_flt + t->f_ab, t->d_flt + t->f_s,
                       G + t->d_ldc, t->d_patch, 2 * (int)t->max_batch);
    // ---- forward ----
    t->zs[0] = const_cast<float *>(x);
    bool f1_done = false;   // stage 1 of the next coupling already ran (tiled stages)
    for (int l = 0; l < n; ++l) {
        const TLayer &L = t->tl.l[l];
        const float *zin = t->zs[l];
        float *zout = t->zs[l + 1];
        switch (L.type) {
        case NF_LAYER_SDN5:
        case NF_LAYER_SDN4:
            hipLaunchKernelGGL(k_sdn_fwd, dim3(nb), dim3(TB), 0, st, g, zin, y, t->d_flt + t->f_ab + 2 * L.aux, zout, t->acc(t->d_ld0 + l));
            break;
        case NF_LAYER_GAIN4:
            hipLaunchKernelGGL(k_scale_fwd, dim3(nb), dim3(TB), 0, st, g, zin, t->d_flt + t->f_s + L.aux, zout);
            break;
        case NF_LAYER_CONV1X1:
            if (l + 1 < n && t->tl.l[l + 1].type == NF_LAYER_COUPLING) break;   // folded into the coupling's l_1 kernel
            hipLaunchKernelGGL(k_mix_fwd, dim3(nb), dim3(TB), 0, st, g, zin, t->d_flt + t->f_A + 16 * L.aux, zout);
            break;
        case NF_LAYER_COUPLING: {
            const bool fold = l > 0 && t->tl.l[l - 1].type == NF_LAYER_CONV1X1;
            const float *zpre = fold ? t->zs[l - 1] : nullptr;
            const float *Am = fold ? t->d_flt + t->f_A + 16 * t->tl.l[l - 1].aux : nullptr;
            if (tiled_ok(t, g, L.width, 2)) {
                // the coupling above: directly, or behind one Conv2d1x1 (then folded into its stage 1)
                int ln = l + 1;
                const bool mixn = ln < n && t->tl.l[ln].type == NF_LAYER_CONV1X1;
                if (mixn) ++ln;
                const TLayer *nxt = ln < n && t->tl.l[ln].type == NF_LAYER_COUPLING && t->tl.l[ln].width == L.width ? &t->tl.l[ln] : nullptr;
                const float *An = nxt && mixn ? t->d_flt + t->f_A + 16 * t->tl.l[ln - 1].aux : nullptr;
                float *nz = nxt ? t->zs[ln] : nullptr;
#define NF_CALL(WW)                                                                                                                \
    do {                                                                                                                           \
        if (g.HW <= 256) coupling_forward_tiled<WW, 256>(t, g, L, zin, zout, t->acc(t->d_ld0 + l), zpre, Am, st, f1_done, nxt, An, nz);   \
        else if (g.HW <= 512) coupling_forward_tiled<WW, 512>(t, g, L, zin, zout, t->acc(t->d_ld0 + l), zpre, Am, st, f1_done, nxt, An, nz); \
        else coupling_forward_tiled<WW, 1024>(t, g, L, zin, zout, t->acc(t->d_ld0 + l), zpre, Am, st, f1_done, nxt, An, nz);       \
    } while (0)
                if (L.width == 4) NF_CALL(4); else NF_CALL(8);
#undef NF_CALL
                f1_done = nxt != nullptr;
            } else if (t->all_gemm || gemm_width(L.width)) {
                if (!coupling_forward_gemm(t, g, L, zin, zout, t->acc(t->d_ld0 + l), zpre, Am, st)) mm_failed = true;
            } else {
                t->pr_next = nullptr;
                if (t->pr && l + 2 < n && t->tl.l[l + 1].type == NF_LAYER_CONV1X1 && t->tl.l[l + 2].type == NF_LAYER_COUPLING &&
                    t->tl.l[l + 2].width == L.width) {
                    t->pr_next = &t->tl.l[l + 2];
                    t->pr_next_A = t->d_flt + t->f_A + 16 * t->tl.l[l + 1].aux;
                    t->pr_next_zin = t->zs[l + 2];
                }
#define NF_CALL(WW) coupling_forward<WW>(t, g, L, zin, zout, t->acc(t->d_ld0 + l), zpre, Am, st)
                NF_WIDTH_SWITCH(L.width, NF_CALL)
#undef NF_CALL
            }
            break;
        }
        }
    }
    // The reported loss / sd_z feed nothing in the backward pass: with one to follow they are computed on the side stream
    // (joined with the filter-gradient work before the step ends) while the main stream starts walking back.
    hipStream_t ls = st;
    if (backward && loss_out) {
        ls = t->serial ? st : t->side;
        (void)hipEventRecord(t->ev_fork[1], st);
        (void)hipStreamWaitEvent(ls, t->ev_fork[1], 0);
    }
    hipLaunchKernelGGL(k_prior, dim3(nb), dim3(TB), 0, ls, g, t->zs[n], s1, s2, t->d_wpart);
    if (loss_out)
        hipLaunchKernelGGL(k_loss, dim3(1), dim3(TB), 0, ls, (int)B, (double)g.HW * 4.0, t->acc(t->d_ld0), n, g.nslot, s1, s2,
                           t->d_wpart, (g.HW & 63) == 0 ? g.HW / 64 : 0, G + t->d_ldc, loss_out);
    if (ls != st) (void)hipEventRecord(t->ev_fork[2], ls);
    if (!backward) {
        t->zs[0] = nullptr;
        if ((e = hipGetLastError()) != hipSuccess) return nf_fail_hip(e, "trainer launch");
        if (t->sync_rc) return nf_fail(NF_EINVAL, "the all-reduce callback of nf_trainer_set_sync failed (status %d)", t->sync_rc);
        return NF_OK;
    }
    // ---- backward ----
    // d loss / d latent = latent / B: formed inside the first stage when the stack ends in a coupling, else by its own kernel
    const bool dz_in_first = t->tl.l[n - 1].type == NF_LAYER_COUPLING;
    if (!dz_in_first) hipLaunchKernelGGL(k_dz_init, dim3(nb), dim3(TB), 0, st, g, t->zs[n], invB, t->dz);
    bool a_done = false;   // stage A' of the next coupling already ran (tiled stages)
    for (int l = n - 1; l >= 0; --l) {
        const TLayer &L = t->tl.l[l];
        switch (L.type) {
        case NF_LAYER_SDN5:
        case NF_LAYER_SDN4:
            hipLaunchKernelGGL(k_sdn_bwd, dim3(nb), dim3(TB), 0, st, g, t->zs[l], y, t->d_flt + t->f_ab + 2 * L.aux, invB, t->dz,
                               t->acc(t->d_dab + 2 * L.aux));
            break;
        case NF_LAYER_GAIN4:
            hipLaunchKernelGGL(k_scale_bwd, dim3(nb), dim3(TB), 0, st, g, t->zs[l + 1], t->d_flt + t->f_s + L.aux, t->dz,
                               t->acc(t->d_dg + L.aux));
            break;
        case NF_LAYER_CONV1X1:
            hipLaunchKernelGGL(k_mix_bwd, dim3(nb), dim3(TB), 0, st, g, t->zs[l], t->d_flt + t->f_A + 16 * L.aux, t->dz,
                               t->acc(t->d_dA + 16 * L.aux));
            break;
        case NF_LAYER_COUPLING: {
            const bool fold = l > 0 && t->tl.l[l - 1].type == NF_LAYER_CONV1X1;
            const float *zmix_in = fold ? t->zs[l - 1] : nullptr;
            const float *Am = fold ? t->d_flt + t->f_A + 16 * t->tl.l[l - 1].aux : nullptr;
            const Acc dA = t->acc(fold ? t->d_dA + 16 * t->tl.l[l - 1].aux : 0);
            const float *zlat = (dz_in_first && l == n - 1) ? t->zs[n] : nullptr;
            if (tiled_ok(t, g, L.width, 1)) {
                const int lb = fold ? l - 2 : l - 1;   // the layer below this coupling (and its folded Conv2d1x1)
                const TLayer *nxt = lb >= 0 && t->tl.l[lb].type == NF_LAYER_COUPLING && t->tl.l[lb].width == L.width ? &t->tl.l[lb] : nullptr;
                const float *nz = nxt ? t->zs[lb] : nullptr;
#define NF_CALL(WW)                                                                                                              \
    do {                                                                                                                         \
        if (g.HW <= 256) coupling_backward_tiled<WW, 256>(t, g, L, t->zs[l], invB, zmix_in, Am, dA, st, zlat, a_done, nxt, nz);  \
        else if (g.HW <= 512) coupling_backward_tiled<WW, 512>(t, g, L, t->zs[l], invB, zmix_in, Am, dA, st, zlat, a_done, nxt, nz); \
        else coupling_backward_tiled<WW, 1024>(t, g, L, t->zs[l], invB, zmix_in, Am, dA, st, zlat, a_done, nxt, nz);             \
    } while (0)
                if (L.width == 4) NF_CALL(4); else NF_CALL(8);
#undef NF_CALL
                a_done = nxt != nullptr;
            } else if (t->all_gemm || gemm_width(L.width)) {
                if (!coupling_backward_gemm(t, g, L, t->zs[l], invB, zmix_in, Am, dA, st, zlat)) mm_failed = true;
            } else {
                t->pr_below = nullptr;
                {
                    const int lb = fold ? l - 2 : l - 1;   // the layer below this coupling (and its folded Conv2d1x1)
                    if (t->pr && lb >= 0 && t->tl.l[lb].type == NF_LAYER_COUPLING && t->tl.l[lb].width == L.width) {
                        t->pr_below = &t->tl.l[lb];
                        t->pr_below_zin = t->zs[lb];
                    }
                }
#define NF_CALL(WW) coupling_backward<WW>(t, g, L, t->zs[l], invB, zmix_in, Am, dA, st, zlat)
                NF_WIDTH_SWITCH(L.width, NF_CALL)
#undef NF_CALL
            }
            if (fold) --l;   // the Conv2d1x1 below was handled by the coupling's last stage
            break;
        }
        }
    }
    if (ls != st) (void)hipStreamWaitEvent(st, t->ev_fork[2], 0);
    for (int par = 0; par < 3; ++par)   // join the side stream: its slots are read next
        if (t->done_pending[par]) {
            (void)hipStreamWaitEvent(st, t->ev_done[par], 0);
            t->done_pending[par] = false;
        }
    if (t->all_gemm || gemm_width(t->width)) {
        // the filters of wide couplings get their gradients whole from the GEMMs: the slotted sums are added up for every other
        // value only (the runs between the filters), then the GEMM results are stored next to them
        int lo = 0;
        ReduceRuns runs;
        StoreJobs jobs;
        auto run = [&](int a, int b) { reduce_run(st, runs, g.nslot, G, t->acc(a).p, a, b); };
        for (int l = 0; l < n; ++l) {
            const TLayer &L = t->tl.l[l];
            if (L.type != NF_LAYER_COUPLING) continue;
            const int w = L.width, off_w2 = L.off + 21 * w;
            run(lo, L.off);                  // ... up to l_1/W
            run(L.off + 18 * w, off_w2);     // l_1/b (and the BN statistics, masked out)
            lo = off_w2 + w * w;             // behind l_2/W: l_2/b, BN, l_last/W (edge rows), b, logs, scale
        }
        run(lo, t->d_ldc);
        reduce_runs_flush(st, runs, g.nslot, G);
        for (int l = 0; l < n; ++l) {        // the filters: l_last/W after the run that holds its edge rows was reduced
            const TLayer &L = t->tl.l[l];
            if (L.type != NF_LAYER_COUPLING) continue;
            const int w = L.width;
            const float *gdw = t->gdw + (size_t)(3 * L.aux) * gemm_part_floats(w);
            store_grad(st, jobs, 18 * w, w, 0, gdw, t->gnp[3 * L.aux], G, L.off);
            store_grad(st, jobs, w * w, w, 0, gdw + gemm_part_floats(w), t->gnp[3 * L.aux + 1], G, L.off + 21 * w);
            store_grad(st, jobs, 36 * w, w, 1, gdw + 2 * gemm_part_floats(w), t->gnp[3 * L.aux + 2], G, L.off + 24 * w + w * w,
                       t->gdual[L.aux] ? 2 * 36 * w : 0);
        }
        store_grads_flush(st, jobs, G);
    } else {
        hipLaunchKernelGGL(k_reduce, dim3((unsigned)t->d_ldc), dim3(64), 0, st, t->d_ldc, t->d_part, g.nslot, G);
    }
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, st, t->tl, t->d_params, ci, g.HW, G + t->d_dA, G + t->d_dab, G + t->d_dg, G);
    float *gout = grads_out ? grads_out : t->d_gradf;
    hipLaunchKernelGGL(k_grads_out, dim3((t->n_params + TB - 1) / TB), dim3(TB), 0, st, t->n_params, G, t->d_mask, gout);
    t->zs[0] = nullptr;
    if ((e = hipGetLastError()) != hipSuccess) return nf_fail_hip(e, "trainer launch");
    if (mm_failed) return nf_fail(NF_EHIP, "a matrix-core GEMM of the wide-coupling training step could not be launched");
    if (t->sync_rc) return nf_fail(NF_EINVAL, "the all-reduce callback of nf_trainer_set_sync failed (status %d)", t->sync_rc);
    return NF_OK;
}

int nf_trainer_set_sync(nf_trainer *t, nf_allreduce_fn fn, void *user, double *sync_buf, int32_t world_size)
{
    if (!t) return nf_fail(NF_EINVAL, "trainer is NULL");
    if (fn && (!sync_buf || world_size < 1)) return nf_fail(NF_EINVAL, "nf_trainer_set_sync needs a device buffer of 64 doubles and world_size >= 1");
    t->sync_fn = fn;
    t->sync_user = user;
    t->sync_buf = fn ? sync_buf : nullptr;
    t->sync_world = fn ? world_size : 1;
    return NF_OK;
}

int nf_trainer_apply(nf_trainer *t, const float *grads, float lr, void *stream)
{
    if (!t) return nf_fail(NF_EINVAL, "trainer is NULL");
    Guard guard;
    int rc = guard.enter(t->device);
    if (rc != NF_OK) return rc;
    const float *gr = grads ? grads : t->d_gradf;
    const unsigned nb = (unsigned)((t->n_params + TB - 1) / TB);
    t->step += 1;
    if (t->optimizer == NF_OPT_ADAM) {
        const double b1 = 0.9, b2 = 0.999;
        const float lr_t = (float)((double)lr * sqrt(1.0 - pow(b2, (double)t->step)) / (1.0 - pow(b1, (double)t->step)));
        hipLaunchKernelGGL(k_adam, dim3(nb), dim3(TB), 0, (hipStream_t)stream, t->n_params, t->d_params, gr, t->d_m, t->d_v,
                           t->d_mask, lr_t, 0.9f, 0.999f, 1e-8f);
    } else {
        hipLaunchKernelGGL(k_momentum, dim3(nb), dim3(TB), 0, (hipStream_t)stream, t->n_params, t->d_params, gr, t->d_m,
                           t->d_mask, lr, 0.9f);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nf_fail_hip(e, "optimizer launch");
    return NF_OK;
}

int nf_trainer_step(nf_trainer *t, const float *x, const float *y, int64_t B, const nf_cond *cond, float lr,
                    float *loss_out, void *stream)
{
    int rc = nf_trainer_forward_backward(t, x, y, B, cond, nullptr, loss_out, stream);
    if (rc != NF_OK) return rc;
    return nf_trainer_apply(t, nullptr, lr, stream);
}

int nf_trainer_get_params(nf_trainer *t, float *params_out, size_t n_params, void *stream)
{
    if (!t || !params_out) return nf_fail(NF_EINVAL, "null argument");
    if (n_params != (size_t)t->n_params) return nf_fail(NF_EINVAL, "n_params mismatch (%zu vs %d)", n_params, t->n_params);
    Guard guard;
    int rc = guard.enter(t->device);
    if (rc != NF_OK) return rc;
    hipError_t e = hipMemcpyAsync(params_out, t->d_params, n_params * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return nf_fail_hip(e, "nf_trainer_get_params");
    return NF_OK;
}

int nf_trainer_set_params(nf_trainer *t, const float *params, size_t n_params, void *stream)
{
    if (!t || !params) return nf_fail(NF_EINVAL, "null argument");
    if (n_params != (size_t)t->n_params) return nf_fail(NF_EINVAL, "n_params mismatch (%zu vs %d)", n_params, t->n_params);
    Guard guard;
    int rc = guard.enter(t->device);
    if (rc != NF_OK) return rc;
    hipError_t e = hipMemcpyAsync(t->d_params, params, n_params * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return nf_fail_hip(e, "nf_trainer_set_params");
    return NF_OK;
}

int64_t nf_trainer_steps(const nf_trainer *t) { return t ? t->step : -1; }

}  // extern "C"

// ---- evaluation under batch statistics on the GEMM path (not part of the ABI: nf_*_batchstats of nf_host.hip call these) ---------
int nf_bs_wide_create(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params, int64_t max_batch,
                      nf_trainer **out)
{
    nf_config c = *cfg;
    c.flags = 0;
    return trainer_create_impl(&c, layers, params, n_params, max_batch, NF_OPT_ADAM, true, out);
}

int64_t nf_bs_wide_capacity(const nf_trainer *t) { return t ? t->max_batch : 0; }

int nf_bs_wide_run(nf_trainer *t, const nf_bs_wide_args &a, hipStream_t st)
{
    if (!t || !t->eval_only) return nf_fail(NF_EINVAL, "internal: not an evaluator");
    if (a.B < 1 || a.B > t->max_batch) return nf_fail(NF_EINVAL, "internal: evaluator capacity %lld < B", (long long)t->max_batch);
    CondIdx ci;
    int rc = cond_index(a.cond, t->has_sdn || t->needs_cond, t->needs_cam, ci);
    if (rc != NF_OK) return rc;
    if (t->has_sdn && !a.y) return nf_fail(NF_EINVAL, "model has a signal-dependent layer but y is NULL");
    Guard guard;
    if ((rc = guard.enter(t->device)) != NF_OK) return rc;
    t->sync_fn = a.sync_fn;
    t->sync_user = a.sync_user;
    t->sync_buf = a.sync_fn ? a.sync_buf : nullptr;
    t->sync_world = a.sync_fn ? a.sync_world : 1;
    t->sync_rc = 0;

    Geo g;
    g.B = (int)a.B;
    g.H = t->cfg.height;
    g.W = t->cfg.width;
    g.HW = g.H * g.W;
    g.npix = a.B * (int64_t)g.HW;
    g.nloop = (g.npix + 63) & ~(int64_t)63;
    const unsigned nb = blocks_for(t, g.npix);
    g.nslot = (t->sync_fn && t->sync_world > 1) ? std::max((int)nb, 2) : (int)nb;
    const int n = t->cfg.n_layers, w = t->width ? t->width : 4;
    const unsigned nB = (unsigned)a.B;
    double *G = t->d_dbl;
    float *z = a.out ? a.out : t->ebuf;
    bool ok = true;
    hipError_t e;

    hipLaunchKernelGGL(k_prep, dim3(1), dim3(64), 0, st, t->tl, t->d_params, ci, g.HW, t->d_flt + t->f_A, t->d_flt + t->f_ab, t->d_flt + t->f_s,
                       G + t->d_ldc, t->d_flt, 0);
    if ((e = hipMemsetAsync(t->eldp, 0, (size_t)a.B * sizeof(double), st)) != hipSuccess) return nf_fail_hip(e, "evaluator set-up");
    hipLaunchKernelGGL(k_e_input, dim3(nb), dim3(TB), 0, st, g, a.in, a.in_scale, a.seed, a.patch_base, z);
    if (t->pr) pr_pack_step(t, st);
    if (a.direction == 0) {
        for (int l = 0; l < n; ++l) {
            const TLayer &L = t->tl.l[l];
            switch (L.type) {
            case NF_LAYER_SDN5:
            case NF_LAYER_SDN4:
                hipLaunchKernelGGL(k_e_sdn<false>, dim3(nB), dim3(TB), 0, st, g.HW, z, a.y, t->d_flt + t->f_ab + 2 * L.aux, t->eldp);
                break;
            case NF_LAYER_GAIN4:
                hipLaunchKernelGGL(k_scale_fwd, dim3(nb), dim3(TB), 0, st, g, (const float *)z, t->d_flt + t->f_s + L.aux, z);
                break;
            case NF_LAYER_CONV1X1:
                if (t->pr && l + 1 < n && t->tl.l[l + 1].type == NF_LAYER_COUPLING) break;   // folded into the coupling's first stage
                hipLaunchKernelGGL(k_mix_fwd, dim3(nb), dim3(TB), 0, st, g, (const float *)z, t->d_flt + t->f_A + 16 * L.aux, z);
                break;
            case NF_LAYER_COUPLING:
                if (t->pr) {
                    const float *Am = (l > 0 && t->tl.l[l - 1].type == NF_LAYER_CONV1X1) ? t->d_flt + t->f_A + 16 * t->tl.l[l - 1].aux : nullptr;
                    if ((t->pr & 3) == 2) pr_coupling_eval<4>(t, g, L, z, st, t->emom + (size_t)L.aux * 4 * w, false, Am);
                    else pr_coupling_eval<8>(t, g, L, z, st, t->emom + (size_t)L.aux * 4 * w, false, Am);
                    break;
                }
                ok = coupling_cnn_gemm(t, g, L, z, st, t->emom + (size_t)L.aux * 4 * w) && ok;
                hipLaunchKernelGGL(k_e_c3<false>, dim3(nB), dim3(TB), 0, st, g.H, g.W, L.width, z, (const float *)t->gp36, (const float *)t->d_params,
                                   L.off + 24 * L.width + L.width * L.width, t->eldp);
                break;
            }
        }
        if (a.nll_out || a.sd_out || a.ld_out || a.sums)
            hipLaunchKernelGGL(k_e_finish, dim3(nB), dim3(TB), 0, st, g.HW, (const float *)z, (const double *)t->eldp, (const double *)(G + t->d_ldc),
                               a.prior ? 1 : 0, a.nll_out, a.sd_out, a.ld_out, a.sums);
    } else {
        if (t->n_mix > 0)
            hipLaunchKernelGGL(k_e_inv4, dim3((unsigned)((t->n_mix + 63) / 64)), dim3(64), 0, st, t->n_mix, (const float *)(t->d_flt + t->f_A), t->eAinv);
        for (int l = n - 1; l >= 0; --l) {
            const TLayer &L = t->tl.l[l];
            switch (L.type) {
            case NF_LAYER_SDN5:
            case NF_LAYER_SDN4:
                hipLaunchKernelGGL(k_e_sdn<true>, dim3(nB), dim3(TB), 0, st, g.HW, z, a.y, t->d_flt + t->f_ab + 2 * L.aux, t->eldp);
                break;
            case NF_LAYER_GAIN4:
                hipLaunchKernelGGL(k_e_scale_mul, dim3(nb), dim3(TB), 0, st, g, z, (const float *)(t->d_flt + t->f_s + L.aux));
                break;
            case NF_LAYER_CONV1X1:
                if (t->pr && l + 1 < n && t->tl.l[l + 1].type == NF_LAYER_COUPLING) break;   // applied by the coupling's last stage
                hipLaunchKernelGGL(k_mix_fwd, dim3(nb), dim3(TB), 0, st, g, (const float *)z, (const float *)(t->eAinv + 16 * L.aux), z);
                break;
            case NF_LAYER_COUPLING:
                if (t->pr) {
                    const float *Ai = (l > 0 && t->tl.l[l - 1].type == NF_LAYER_CONV1X1) ? t->eAinv + 16 * t->tl.l[l - 1].aux : nullptr;
                    if ((t->pr & 3) == 2) pr_coupling_eval<4>(t, g, L, z, st, t->emom + (size_t)L.aux * 4 * w, true, Ai);
                    else pr_coupling_eval<8>(t, g, L, z, st, t->emom + (size_t)L.aux * 4 * w, true, Ai);
                    break;
                }
                ok = coupling_cnn_gemm(t, g, L, z, st, t->emom + (size_t)L.aux * 4 * w) && ok;
                hipLaunchKernelGGL(k_e_c3<true>, dim3(nB), dim3(TB), 0, st, g.H, g.W, L.width, z, (const float *)t->gp36, (const float *)t->d_params,
                                   L.off + 24 * L.width + L.width * L.width, t->eldp);
                break;
            }
        }
    }
    if ((e = hipGetLastError()) != hipSuccess) return nf_fail_hip(e, "evaluator launch");
    if (!ok) return nf_fail(NF_EHIP, "a matrix-core GEMM of the batch-statistics evaluation could not be launched");
    const size_t nmom = t->cpl.size() * 4 * (size_t)w;
    if (a.moments_out && nmom) e = hipMemcpyAsync(a.moments_out, t->emom, nmom * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // like every nf_*_batchstats call
    if (e != hipSuccess) return nf_fail_hip(e, "batch-statistics evaluation");
    if (t->sync_rc) return nf_fail(NF_EINVAL, "the all-reduce callback of nf_set_sync failed (status %d)", t->sync_rc);
    return NF_OK;
}
