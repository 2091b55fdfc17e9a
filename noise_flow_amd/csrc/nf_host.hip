// Host side of the C ABI (include/noiseflow_hip.h): parameter folding, program
// construction, conditioning scalars, launches.  No torch, no Python here.
//
// Reference call sites replaced (paths relative to /root/reference):
//   matrix_param.py:100-140      PLU -> A, A^-1, log|det|        (fold_conv1x1)
//   layers.py:378-401            BN eval folded into l_1 / l_2    (fold_coupling)
//   layers.py:555-583,651-674    edge channel + exp(3*logs)       (fold_coupling)
//   cond_utils.py:205-239        sdn5 scalars                     (sdn5_scalars)
//   cond_utils.py:432-440        gain4                            (build_program)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/noiseflow_hip.h"
#include "nf_device.h"
#include "nf_gemm_layout.h"
#include "nf_internal.h"

hipError_t nf_launch_flow(const NfProgram &prog, const NfLaunch &a, int n_cu, hipStream_t stream, bool matrix_core);
hipError_t nf_launch_wide(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream);
hipError_t nf_launch_wide16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream);
hipError_t nf_launch_gemm(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream);
hipError_t nf_launch_gemm16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream);
hipError_t nf_launch_gemm16b(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream);
bool nf_gemm_shape_ok(int H, int W);
hipError_t nf_launch_gemmb(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream);
bool nf_gemmb_shape_ok(int wp, int H, int W);
hipError_t nf_launch_synth(uint64_t seed, int64_t patch_base, int64_t B, int HW, float beta1, float beta2,
                           float *y_out, float *x_out, hipStream_t stream);
hipError_t nf_launch_eps(uint64_t seed, int64_t patch_base, int64_t B, int HW, float *eps_out, hipStream_t stream);
hipError_t nf_launch_stats_compact(const double *stats, int nvals, double *buf, hipStream_t stream);
hipError_t nf_launch_stats_scatter(double *stats, int nvals, const double *buf, hipStream_t stream);
hipError_t nf_launch_sums_reduce(const double *wide, double *out3, bool accumulate, hipStream_t stream);
hipError_t nf_launch_bs_finalize(double *stats, int w, double n, float *Wm, int rows, float *Bv, float *mean_out, float *var_out,
                                 hipStream_t stream);
hipError_t nf_launch_tile_combine(const float *part, const NfTileParts &tp, int64_t B, double n, double ld_const, uint32_t flags,
                                  float *nll_out, float *sd_out, float *ld_out, double *sums, hipStream_t stream);
hipError_t nf_launch_gather(float *dst, const float *src, const int32_t *pairs, int n, hipStream_t stream);

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    return fail(NF_EHIP, "%s: %s", what, hipGetErrorString(e));
}

constexpr double kBnEps = 1e-4;          // layers.py:378
constexpr double kLogscaleFactor = 3.0;  // layers.py:653
constexpr int kC = 4;

int64_t layer_param_count(int32_t type, int32_t w)
{
    switch (type) {
    case NF_LAYER_CONV1X1: return 16 + 4 + 4 + 6 + 6;
    case NF_LAYER_CONV1X1_NONE: return 16;
    case NF_LAYER_CONV1X1_LU2: return 16 + 16 + 4 + 4 + 16;
    case NF_LAYER_PERMUTE: return 0;
    case NF_LAYER_COUPLING:
        if (w <= 0) return -1;
        return 9 * 2 * (int64_t)w + w + w + w      // l_1/W, l_1/b, bn1 mean, var
               + (int64_t)w * w + w + w + w        // l_2/W, l_2/b, bn2 mean, var
               + 9 * ((int64_t)w + 1) * 4 + 4 + 4  // l_last/W, b, logs
               + 1;                                // rescaling_scale
    case NF_LAYER_SDN5: return 1 + 1 + 5 + 15 + 1;
    case NF_LAYER_GAIN4: return 1;
    case NF_LAYER_SDN4: return 7;
    case NF_LAYER_SDN: return 2;
    case NF_LAYER_GAIN: return 2;
    case NF_LAYER_SDN1:
    case NF_LAYER_SDN2:
    case NF_LAYER_SDN3: return 7;
    case NF_LAYER_SDN6: return 13;
    case NF_LAYER_GAIN1: return 2;
    case NF_LAYER_GAIN2:
    case NF_LAYER_GAIN3: return 5;
    default: return -1;
    }
}

// ---- PLU parameterisation (matrix_param.py:31-56, 100-140) -----------------
struct Mat4 {
    double m[4][4];
};

Mat4 matmul(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}

void fold_conv1x1(const float *p, Mat4 &A, Mat4 &Ainv, double &log_abs_det)
{
    const float *P = p, *sign_s = p + 16, *log_s = p + 20, *lv = p + 24, *uv = p + 30;
    Mat4 Pm, L, U;
    memset(&L, 0, sizeof(L));
    memset(&U, 0, sizeof(U));
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) Pm.m[i][j] = P[i * 4 + j];
    // tfdist.fill_triangular(v, lower) for the 3x3 strict triangle, padded by one
    // zero row on top and one zero column on the right:  [[v3,0,0],[v5,v4,0],[v2,v1,v0]]
    L.m[1][0] = lv[3];
    L.m[2][0] = lv[5]; L.m[2][1] = lv[4];
    L.m[3][0] = lv[2]; L.m[3][1] = lv[1]; L.m[3][2] = lv[0];
    for (int i = 0; i < 4; ++i) L.m[i][i] = 1.0;
    // upper: [[v0,v1,v2],[0,v4,v5],[0,0,v3]] padded by a zero row at the bottom and a zero column on the left
    U.m[0][1] = uv[0]; U.m[0][2] = uv[1]; U.m[0][3] = uv[2];
    U.m[1][2] = uv[4]; U.m[1][3] = uv[5];
    U.m[2][3] = uv[3];
    log_abs_det = 0.0;
    for (int i = 0; i < 4; ++i) {
        U.m[i][i] = (double)sign_s[i] * exp((double)log_s[i]);
        log_abs_det += (double)log_s[i];
    }
    A = matmul(Pm, matmul(L, U));
    // A^-1 = U^-1 L^-1 P^T by two triangular solves (matrix_param.py:132-136)
    Mat4 X;   // X = L^-1 P^T  (forward substitution, unit diagonal)
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i) {
            double s = Pm.m[c][i];   // P^T[i][c]
            for (int k = 0; k < i; ++k) s -= L.m[i][k] * X.m[k][c];
            X.m[i][c] = s;
        }
    for (int c = 0; c < 4; ++c)      // back substitution with U
        for (int i = 3; i >= 0; --i) {
            double s = X.m[i][c];
            for (int k = i + 1; k < 4; ++k) s -= U.m[i][k] * Ainv.m[k][c];
            Ainv.m[i][c] = s / U.m[i][i];
        }
}

// General 4x4 inverse + log|det| by Gauss-Jordan with partial pivoting, in double (tf.matrix_inverse / tf.linalg.slogdet
// of matrix_param.py:26-27, :177-179).  false = singular.
bool invert4(const Mat4 &M, Mat4 &inv, double &log_abs_det)
{
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = M.m[i][j];
            a[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    log_abs_det = 0.0;
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (!(fabs(a[piv][c]) > 0.0)) return false;
        if (piv != c)
            for (int j = 0; j < 8; ++j) std::swap(a[piv][j], a[c][j]);
        const double d = a[c][c];
        log_abs_det += log(fabs(d));
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.0)
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv.m[i][j] = a[i][4 + j];
    return true;
}

// decomp = 'NONE' (matrix_param.py:23-29): the matrix is the variable.
bool fold_conv1x1_none(const float *p, Mat4 &A, Mat4 &Ainv, double &log_abs_det)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) A.m[i][j] = p[i * 4 + j];
    return invert4(A, Ainv, log_abs_det);
}

// decomp = 'LU2' (matrix_param.py:143-188): full-matrix L / U variables masked to their strict triangles, evaluated in
// float64; A^-1 = U^-1 L^-1 P^-1 from the three separate inverses, as the reference forms it (:177-180).
bool fold_conv1x1_lu2(const float *p, Mat4 &A, Mat4 &Ainv, double &log_abs_det)
{
    const float *P = p, *Lf = p + 16, *sign_s = p + 32, *log_s = p + 36, *Uf = p + 40;
    Mat4 Pm, L, U, Pi, Li, Ui;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            Pm.m[i][j] = P[i * 4 + j];
            L.m[i][j] = j < i ? (double)Lf[i * 4 + j] : (i == j ? 1.0 : 0.0);
            U.m[i][j] = j > i ? (double)Uf[i * 4 + j] : (i == j ? (double)sign_s[i] * exp((double)log_s[i]) : 0.0);
        }
    log_abs_det = 0.0;
    for (int i = 0; i < 4; ++i) log_abs_det += (double)log_s[i];           // :187
    A = matmul(Pm, matmul(L, U));
    double d;
    if (!invert4(Pm, Pi, d) || !invert4(L, Li, d) || !invert4(U, Ui, d)) return false;
    Ainv = matmul(Ui, matmul(Li, Pi));
    return true;
}

// ---- coupling CNN folding ---------------------------------------------------
void fold_coupling(const float *p, int w, float *out)
{
    const float *W1 = p;
    const float *b1 = W1 + 18 * w;
    const float *m1 = b1 + w;
    const float *v1 = m1 + w;
    const float *W2 = v1 + w;
    const float *b2 = W2 + w * w;
    const float *m2 = b2 + w;
    const float *v2 = m2 + w;
    const float *W3 = v2 + w;            // [3][3][w+1][4]
    const float *b3 = W3 + 9 * (w + 1) * 4;
    const float *logs = b3 + 4;
    const float *resc = logs + 4;

    std::vector<double> s1(w), s2(w);
    for (int j = 0; j < w; ++j) {
        s1[j] = 1.0 / sqrt((double)v1[j] + kBnEps);
        s2[j] = 1.0 / sqrt((double)v2[j] + kBnEps);
    }
    double es[4];
    for (int j = 0; j < 4; ++j) es[j] = exp(kLogscaleFactor * (double)logs[j]);

    float *E = out + nf_cpl_off_E(w);
    for (int mask = 0; mask < 16; ++mask) {
        const bool top = mask & 1, bottom = mask & 2, left = mask & 4, right = mask & 8;
        for (int j = 0; j < 4; ++j) {
            double s = b3[j];
            for (int di = 0; di < 3; ++di)
                for (int dj = 0; dj < 3; ++dj) {
                    const bool outside = (di == 0 && top) || (di == 2 && bottom) || (dj == 0 && left) || (dj == 2 && right);
                    if (outside) s += (double)W3[((di * 3 + dj) * (w + 1) + w) * 4 + j];
                }
            E[mask * 4 + j] = (float)(s * es[j]);
        }
    }
    float *W3o = out + nf_cpl_off_W3(w);
    for (int tap = 0; tap < 9; ++tap)
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < 4; ++j)
                W3o[(tap * w + i) * 4 + j] = (float)((double)W3[(tap * (w + 1) + i) * 4 + j] * es[j]);
    float *W1o = out + nf_cpl_off_W1(w);
    for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < w; ++j)
                W1o[(tap * 2 + c) * w + j] = (float)((double)W1[(tap * 2 + c) * w + j] * s1[j]);
    float *B1o = out + nf_cpl_off_B1(w);
    for (int j = 0; j < w; ++j) B1o[j] = (float)(((double)b1[j] - (double)m1[j]) * s1[j]);
    float *W2o = out + nf_cpl_off_W2(w);
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < w; ++j) W2o[i * w + j] = (float)((double)W2[i * w + j] * s2[j]);
    float *B2o = out + nf_cpl_off_B2(w);
    for (int j = 0; j < w; ++j) B2o[j] = (float)(((double)b2[j] - (double)m2[j]) * s2[j]);
    float *S = out + nf_cpl_off_S(w);
    S[0] = resc[0];
    S[1] = S[2] = S[3] = 0.0f;
}

// The same folded coupling block re-laid out j-major for the matrix-core kernel
// (nf_device.h, NF2_CPL_*), width 4 only.
void relayout_coupling_v2(const float *v1, float *out)
{
    const int w = 4;
    const double k2 = 2.0 * 1.4426950408889634;   // 2*log2(e): raw columns feed exp2() directly
    const double log2e = 1.4426950408889634;
    memcpy(out + NF2_CPL_E, v1 + nf_cpl_off_E(w), 64 * sizeof(float));
    for (int m = 0; m < 16; ++m)
        for (int j = 2; j < 4; ++j) out[NF2_CPL_E + 4 * m + j] = (float)((double)v1[nf_cpl_off_E(w) + 4 * m + j] * k2);
    memcpy(out + NF2_CPL_B1, v1 + nf_cpl_off_B1(w), 4 * sizeof(float));
    memcpy(out + NF2_CPL_B2, v1 + nf_cpl_off_B2(w), 4 * sizeof(float));
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF2_CPL_S + 0] = (float)sc;
    out[NF2_CPL_S + 1] = (float)(sc * log2e);          // scl:   ls*log2(e) = scl*tanh(raw)
    out[NF2_CPL_S + 2] = (float)(-2.0 * sc * log2e);   // -2*scl
    out[NF2_CPL_S + 3] = 0.0f;
    for (int j = 0; j < 4; ++j) {
        for (int di = 0; di < 3; ++di)
            for (int q = 0; q < 8; ++q)
                out[NF2_CPL_W1T + 24 * j + 8 * di + q] = q < 6 ? v1[nf_cpl_off_W1(w) + (di * 6 + q) * 4 + j] : 0.0f;
        for (int i = 0; i < 4; ++i) out[NF2_CPL_W2T + 4 * j + i] = v1[nf_cpl_off_W2(w) + i * 4 + j];
        for (int k = 0; k < 36; ++k) {
            const double wv = v1[nf_cpl_off_W3(w) + k * 4 + j];
            out[NF2_CPL_W3T + 36 * j + k] = (float)(j >= 2 ? wv * k2 : wv);
        }
    }
}

// fp16-CNN re-layout (nf_device.h, NF3_CPL_*): folded weights rounded to IEEE half.
uint16_t to_half(float f)
{
    _Float16 h = (_Float16)f;   // round-to-nearest-even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

// l_last weights of the fp16-CNN layouts: output channel j of 4 = (shift, shift, raw, raw).  The raw columns carry the 2 log2(e)
// of  t = exp2(2 log2(e) raw) = exp(2 raw)  INSIDE the rounded weight — folded before the rounding to half, like the batch-norm
// scale and exp(3 logs) — so that no kernel multiplies behind its matrix instructions (oracle/nf_oracle.py::coupling_cnn_fp16
// rounds at the same point).
uint16_t to_half_w3(float wv, int j)
{
    const double k2 = 2.0 * 1.4426950408889634;
    _Float16 h = (_Float16)(j >= 2 ? (double)wv * k2 : (double)wv);
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

void relayout_coupling_v3(const float *v1, float *out)
{
    const int w = 4;
    const double log2e = 1.4426950408889634, k2 = 2.0 * log2e;
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double e = v1[nf_cpl_off_E(w) + 4 * m + j];
            out[NF3_CPL_E + 4 * m + j] = (float)(j >= 2 ? e * k2 : e);   // raw columns feed exp2() directly
        }
    memcpy(out + NF3_CPL_B1, v1 + nf_cpl_off_B1(w), 4 * sizeof(float));
    memcpy(out + NF3_CPL_B2, v1 + nf_cpl_off_B2(w), 4 * sizeof(float));
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF3_CPL_S + 0] = (float)sc;
    out[NF3_CPL_S + 1] = (float)(sc * log2e);
    out[NF3_CPL_S + 2] = (float)(-2.0 * sc * log2e);
    out[NF3_CPL_S + 3] = 0.0f;
    uint16_t *h1 = reinterpret_cast<uint16_t *>(out + NF3_CPL_W1H);
    uint16_t *h2 = reinterpret_cast<uint16_t *>(out + NF3_CPL_W2H);
    uint16_t *h3 = reinterpret_cast<uint16_t *>(out + NF3_CPL_W3H);
    auto W1 = [&](int di, int dj, int c, int j) { return to_half(v1[nf_cpl_off_W1(w) + ((di * 3 + dj) * 2 + c) * 4 + j]); };
    for (int j = 0; j < 4; ++j) {
        for (int di = 0; di < 3; ++di) {
            uint16_t *g = h1 + ((j * 3 + di) * 4) * 4;   // 4 groups x 4 halves
            const uint16_t z = 0;
            const uint16_t grp[4][4] = {
                {W1(di, 0, 0, j), W1(di, 0, 1, j), W1(di, 1, 0, j), W1(di, 1, 1, j)},   // dx=0, window pair (wc0,wc1)
                {W1(di, 2, 0, j), W1(di, 2, 1, j), z, z},                               // dx=0, pair (wc2,wc3)
                {z, z, W1(di, 0, 0, j), W1(di, 0, 1, j)},                               // dx=1, pair (wc0,wc1)
                {W1(di, 1, 0, j), W1(di, 1, 1, j), W1(di, 2, 0, j), W1(di, 2, 1, j)}};  // dx=1, pair (wc2,wc3)
            memcpy(g, grp, sizeof(grp));
        }
        for (int i = 0; i < 4; ++i) h2[j * 4 + i] = to_half(v1[nf_cpl_off_W2(w) + i * 4 + j]);
        for (int tap = 0; tap < 9; ++tap)
            for (int i = 0; i < 4; ++i) h3[j * (2 * NF3_W3H_STRIDE) + tap * 4 + i] = to_half_w3(v1[nf_cpl_off_W3(w) + (tap * 4 + i) * 4 + j], j);
    }
}

// fp16-CNN re-layout for v_mfma_f32_16x16x32_f16 (nf_device.h, NF11_*): the A operands in the order the lanes fetch them.
void relayout_coupling_v11(const float *v1, float *out, bool with_a2)
{
    const int w = 4;
    const double log2e = 1.4426950408889634, k2 = 2.0 * log2e;
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double e = v1[nf_cpl_off_E(w) + 4 * m + j];
            out[NF11_CPL_E + 4 * m + j] = (float)(j >= 2 ? e * k2 : e);   // raw columns feed exp2() directly
        }
    memcpy(out + NF11_CPL_B1, v1 + nf_cpl_off_B1(w), 4 * sizeof(float));
    memcpy(out + NF11_CPL_B2, v1 + nf_cpl_off_B2(w), 4 * sizeof(float));
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF11_CPL_S + 0] = (float)sc;
    out[NF11_CPL_S + 1] = (float)(sc * log2e);
    out[NF11_CPL_S + 2] = (float)(-2.0 * sc * log2e);
    out[NF11_CPL_S + 3] = 0.0f;
    uint16_t *h2 = reinterpret_cast<uint16_t *>(out + NF11_CPL_W2H);
    uint16_t *a1 = reinterpret_cast<uint16_t *>(out + NF11_CPL_A1);
    uint16_t *a3 = reinterpret_cast<uint16_t *>(out + NF11_CPL_A3);
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) h2[j * 4 + i] = to_half(v1[nf_cpl_off_W2(w) + i * 4 + j]);
    auto tap_ok = [](int d) { return d >= 0 && d <= 2; };
    for (int l = 0; l < 64; ++l) {
        const int gk = l >> 4, m = l & 15, a = m >> 3, p = (m >> 2) & 1, j = m & 3;
        for (int e = 0; e < 8; ++e) {
            {   // l_1: element e = 2 wc + c of window row nf11_l1_row(gk)
                const int wc = e >> 1, c = e & 1, di = nf11_l1_row(gk) - a, dj = wc - p;
                a1[l * 8 + e] = tap_ok(di) && tap_ok(dj) ? to_half(v1[nf_cpl_off_W1(w) + ((di * 3 + dj) * 2 + c) * 4 + j]) : (uint16_t)0;
            }
            for (int m3 = 0; m3 < 2; ++m3) {   // l_last: element e = 4 px + c of window row nf11_l3_row(gk, m3), column pair gk >> 1
                const int wc = 2 * (gk >> 1) + (e >> 2), c = e & 3, di = nf11_l3_row(gk, m3) - a, dj = wc - p;
                a3[(m3 * 64 + l) * 8 + e] = tap_ok(di) && tap_ok(dj) ? to_half_w3(v1[nf_cpl_off_W3(w) + ((di * 3 + dj) * 4 + c) * 4 + j], j) : (uint16_t)0;
            }
        }
    }
    if (with_a2) {   // l_2 on the same instruction (nf_device.h, NF11_CPL_A2): block-diagonal over the lane's own K slot
        uint16_t *a2 = reinterpret_cast<uint16_t *>(out + NF11_CPL_A2);
        for (int half = 0; half < 2; ++half)
            for (int l = 0; l < 64; ++l) {
                const int gk = l >> 4, m = l & 15, g = m >> 2, j = m & 3;
                for (int e = 0; e < 8; ++e)
                    a2[(half * 64 + l) * 8 + e] = (gk == g && (e >> 2) == half) ? to_half(v1[nf_cpl_off_W2(w) + (e & 3) * 4 + j]) : (uint16_t)0;
            }
    }
}

// Wide-CNN re-layout (nf_device.h, NF4_*; coupling width 32): every weight in the order the lanes of
// v_mfma_f32_32x32x2_f32 / v_mfma_f32_4x4x1 fetch their A operands (nf_wide.hip).
// `w` = the coupling's own width (8, 16 or 32): narrower CNNs are zero-padded to 32 hidden channels (exact: a padded
// channel has zero weights and zero bias in l_1 / l_2, hence zero activation, and zero l_last weights).
void relayout_coupling_wide32(const float *v1, int w, float *out)
{
    const double k2 = 2.0 * 1.4426950408889634, log2e = 1.4426950408889634;
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double e = v1[nf_cpl_off_E(w) + 4 * m + j];
            out[NF4_CPL_E + 4 * m + j] = (float)(j >= 2 ? e * k2 : e);   // raw columns feed exp2() directly
        }
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF4_CPL_S + 0] = (float)sc;
    out[NF4_CPL_S + 1] = (float)(sc * log2e);
    out[NF4_CPL_S + 2] = (float)(-2.0 * sc * log2e);
    out[NF4_CPL_S + 3] = 0.0f;
    float *img = out + NF4_CPL_IMG;
    const float *W1 = v1 + nf_cpl_off_W1(w), *B1 = v1 + nf_cpl_off_B1(w), *W2 = v1 + nf_cpl_off_W2(w);
    const float *B2 = v1 + nf_cpl_off_B2(w), *W3 = v1 + nf_cpl_off_W3(w);
    for (int step = 0; step < 12; ++step)          // l_1: step = tap, K slice = input channel
        for (int l = 0; l < 64; ++l)
            img[NF4_IMG_A1 + ((step >> 2) * 64 + l) * 4 + (step & 3)] =
                (step < 9 && (l & 31) < w) ? W1[(step * 2 + (l >> 5)) * w + (l & 31)] : 0.0f;
    for (int g = 0; g < 2; ++g)
        for (int v = 0; v < 16; ++v) {
            const int ch = nf4_chan(v, g);
            img[NF4_IMG_B1 + g * 16 + v] = ch < w ? B1[ch] : 0.0f;
            img[NF4_IMG_B2 + g * 16 + v] = ch < w ? B2[ch] : 0.0f;
        }
    // taps (di,dj) of the 8 off-centre rows groups of P, by (a, g')
    static const int tap_of[4][2] = {{0 * 3 + 0, 2 * 3 + 0}, {0 * 3 + 2, 2 * 3 + 2}, {0 * 3 + 1, 2 * 3 + 1}, {1 * 3 + 0, 1 * 3 + 2}};
    for (int s = 0; s < 16; ++s)
        for (int l = 0; l < 64; ++l) {
            const int cin = nf4_chan(s, l >> 5), i = l & 31;
            img[NF4_IMG_A2 + ((s >> 2) * 64 + l) * 4 + (s & 3)] = (cin < w && i < w) ? W2[cin * w + i] : 0.0f;
            const int a = i >> 3, gp = (i >> 2) & 1, j = i & 3;
            const double wv = cin < w ? W3[(tap_of[a][gp] * w + cin) * 4 + j] : 0.0;
            img[NF4_IMG_A3 + ((s >> 2) * 64 + l) * 4 + (s & 3)] = (float)(j >= 2 ? wv * k2 : wv);
        }
    for (int s = 0; s < 16; ++s)
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 4; ++j) {
                const double wv = nf4_chan(s, g) < w ? W3[(4 * w + nf4_chan(s, g)) * 4 + j] : 0.0;   // centre tap (1,1)
                img[NF4_IMG_A3C + ((s >> 2) * 8 + g * 4 + j) * 4 + (s & 3)] = (float)(j >= 2 ? wv * k2 : wv);
            }
}

// GEMM re-layout (nf_device.h, NF7_*; widths 33 .. 512 zero-padded to wp = 64 / 128 / 256 / 512): every weight in the order the
// wavefront that consumes it fetches it from L2 (nf_gemm.hip).
void relayout_coupling_gemm(const float *v1, int w, int wp, float *out)
{
    const double k2 = 2.0 * 1.4426950408889634, log2e = 1.4426950408889634;
    const int MT = wp / 32, KC = wp / 8;
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double e = v1[nf_cpl_off_E(w) + 4 * m + j];
            out[NF7_CPL_E + 4 * m + j] = (float)(j >= 2 ? e * k2 : e);   // raw columns feed exp2() directly
        }
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF7_CPL_S + 0] = (float)sc;
    out[NF7_CPL_S + 1] = (float)(sc * log2e);
    out[NF7_CPL_S + 2] = (float)(-2.0 * sc * log2e);
    out[NF7_CPL_S + 3] = 0.0f;
    float *img = out + NF7_CPL_IMG;
    const float *W1 = v1 + nf_cpl_off_W1(w), *B1 = v1 + nf_cpl_off_B1(w), *W2 = v1 + nf_cpl_off_W2(w);
    const float *B2 = v1 + nf_cpl_off_B2(w), *W3 = v1 + nf_cpl_off_W3(w);
    for (int m = 0; m < MT; ++m) {
        for (int step = 0; step < 12; ++step)          // l_1: step = tap, K slice = input channel
            for (int l = 0; l < 64; ++l) {
                const int oc = 32 * m + (l & 31);
                img[nf7_img_A1(wp) + ((m * 3 + (step >> 2)) * 64 + l) * 4 + (step & 3)] =
                    (step < 9 && oc < w) ? W1[(step * 2 + (l >> 5)) * w + oc] : 0.0f;
            }
        for (int g = 0; g < 2; ++g)
            for (int v = 0; v < 16; ++v) {
                const int ch = 32 * m + nf4_chan(v, g);
                img[nf7_img_B1(wp) + m * 32 + g * 16 + v] = ch < w ? B1[ch] : 0.0f;
                img[nf7_img_B2(wp) + m * 32 + g * 16 + v] = ch < w ? B2[ch] : 0.0f;
            }
        for (int kc = 0; kc < KC; ++kc)                // l_2: K step kk = 4 kc + s consumes input tile kk / 16, register kk % 16
            for (int l = 0; l < 64; ++l)
                for (int s2 = 0; s2 < 4; ++s2) {
                    const int kk = 4 * kc + s2, cin = 32 * (kk / 16) + nf4_chan(kk % 16, l >> 5), oc = 32 * m + (l & 31);
                    img[nf7_img_A2(wp) + (((size_t)m * KC + kc) * 64 + l) * 4 + s2] = (cin < w && oc < w) ? W2[(size_t)cin * w + oc] : 0.0f;
                }
    }
    for (int mi = 0; mi < MT; ++mi)                    // P = W3^T h2: taps 0 .. 7 as one 32-row GEMM (row i = 4 tap + j) ...
        for (int v = 0; v < 16; ++v)
            for (int l = 0; l < 64; ++l) {
                const int row = l & 31, cin = 32 * mi + nf4_chan(v, l >> 5);
                double wv = 0.0;
                if (cin < w) {
                    const int tap = row >> 2, j = row & 3;
                    wv = W3[((size_t)tap * w + cin) * 4 + j];
                    if (j >= 2) wv *= k2;
                }
                img[nf7_img_A3(wp) + (((size_t)mi * 4 + (v >> 2)) * 64 + l) * 4 + (v & 3)] = (float)wv;
            }
    for (int mi = 0; mi < MT; ++mi)                    // ... and tap 8 on v_mfma_f32_4x4x1 (4 rows, not a 32-row tile)
        for (int v = 0; v < 16; ++v)
            for (int g = 0; g < 2; ++g)
                for (int j = 0; j < 4; ++j) {
                    const int cin = 32 * mi + nf4_chan(v, g);
                    double wv = cin < w ? (double)W3[((size_t)8 * w + cin) * 4 + j] : 0.0;
                    if (j >= 2) wv *= k2;
                    img[nf7_img_A3C(wp) + (((size_t)mi * 4 + (v >> 2)) * 8 + g * 4 + j) * 4 + (v & 3)] = (float)wv;
                }
}

// Variant B of the GEMM layout (NF10_*, widths <= 128): the NF7 values, one contiguous slab per channel tile.
void relayout_coupling_gemmb(const float *v1, int w, int wp, float *out)
{
    std::vector<float> a(nf7_cpl_size(wp));
    relayout_coupling_gemm(v1, w, wp, a.data());
    const int MT = wp / 32, KC = wp / 8;
    memcpy(out, a.data(), (size_t)(NF7_CPL_IMG + MT * 832) * sizeof(float));      // E, S, A1, B1 (+ NF7's B2 block, unused)
    const float *ia = a.data() + NF7_CPL_IMG;
    float *io = out + NF7_CPL_IMG;
    for (int m = 0; m < MT; ++m) {
        float *sl = io + nf10_img_SLAB(wp) + (size_t)m * nf10_slab_floats(wp);
        memcpy(sl, ia + nf7_img_A2(wp) + (size_t)m * KC * 256, (size_t)KC * 256 * sizeof(float));
        memcpy(sl + nf10_slab_A3(wp), ia + nf7_img_A3(wp) + (size_t)m * 1024, 1024 * sizeof(float));
        memcpy(sl + nf10_slab_A3C(wp), ia + nf7_img_A3C(wp) + (size_t)m * 128, 128 * sizeof(float));
        memcpy(sl + nf10_slab_B2(wp), ia + nf7_img_B2(wp) + (size_t)m * 32, 32 * sizeof(float));
    }
}

// fp16-CNN GEMM re-layout (nf_gemm_layout.h, NF8_*; NF_CFG_FP16_CNN at widths 33 .. 512): fetch order of v_mfma_f32_32x32x16_f16,
// folded weights rounded to half once (the oracle's rounding points), biases / border table fp32.
void relayout_coupling_gemm16(const float *v1, int w, int wp, float *out)
{
    const double k2 = 2.0 * 1.4426950408889634, log2e = 1.4426950408889634;
    const int MT = wp / 32, KS = wp / 16;
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double e = v1[nf_cpl_off_E(w) + 4 * m + j];
            out[NF8_CPL_E + 4 * m + j] = (float)(j >= 2 ? e * k2 : e);
        }
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF8_CPL_S + 0] = (float)sc;
    out[NF8_CPL_S + 1] = (float)(sc * log2e);
    out[NF8_CPL_S + 2] = (float)(-2.0 * sc * log2e);
    out[NF8_CPL_S + 3] = 0.0f;
    float *img = out + NF8_CPL_IMG;
    memset(img, 0, (size_t)nf8_img_size(wp) * sizeof(float));
    uint16_t *h = reinterpret_cast<uint16_t *>(img);   // half index = 2 * dword index
    const float *W1 = v1 + nf_cpl_off_W1(w), *B1 = v1 + nf_cpl_off_B1(w), *W2 = v1 + nf_cpl_off_W2(w);
    const float *B2 = v1 + nf_cpl_off_B2(w), *W3 = v1 + nf_cpl_off_W3(w);
    for (int m = 0; m < MT; ++m) {
        for (int l = 0; l < 64; ++l) {
            const int oc = 32 * m + (l & 31), g = l >> 5;
            for (int q = 0; q < 8; ++q) {
                // l_1, instruction 0: taps 4g .. 4g+3; instruction 1: tap 8 on lane half 0
                const int tap = 4 * g + (q >> 1), ch = q & 1;
                h[2 * ((size_t)nf8_img_A1H(wp) + ((m * 2 + 0) * 64 + l) * 4) + q] = oc < w ? to_half(W1[(tap * 2 + ch) * w + oc]) : 0;
                h[2 * ((size_t)nf8_img_A1H(wp) + ((m * 2 + 1) * 64 + l) * 4) + q] = (oc < w && g == 0 && q < 2) ? to_half(W1[(8 * 2 + q) * w + oc]) : 0;
                for (int ks = 0; ks < KS; ++ks) {
                    const int cin = 32 * (ks >> 1) + nf4_chan(8 * (ks & 1) + q, g);
                    h[2 * ((size_t)nf8_img_A2H(wp) + (((size_t)m * KS + ks) * 64 + l) * 4) + q] =
                        (cin < w && oc < w) ? to_half(W2[(size_t)cin * w + oc]) : 0;
                }
                for (int m2 = 0; m2 < 2; ++m2) {       // P rows of taps 0 .. 7 from INPUT tile m
                    const int cin = 32 * m + nf4_chan(8 * m2 + q, g), row = l & 31, tap = row >> 2, j = row & 3;
                    h[2 * ((size_t)nf8_img_A3H(wp) + ((m * 2 + m2) * 64 + l) * 4) + q] = cin < w ? to_half_w3(W3[((size_t)tap * w + cin) * 4 + j], j) : 0;
                }
            }
        }
        for (int g = 0; g < 2; ++g)
            for (int v = 0; v < 16; ++v) {
                const int ch = 32 * m + nf4_chan(v, g);
                img[nf8_img_B1(wp) + m * 32 + g * 16 + v] = ch < w ? B1[ch] : 0.0f;
                img[nf8_img_B2(wp) + m * 32 + g * 16 + v] = ch < w ? B2[ch] : 0.0f;
            }
        for (int q4 = 0; q4 < 4; ++q4)                 // tap 8 on v_mfma_f32_4x4x4_16b_f16
            for (int g = 0; g < 2; ++g)
                for (int j = 0; j < 4; ++j)
                    for (int r = 0; r < 4; ++r) {
                        const int cin = 32 * m + nf4_chan(4 * q4 + r, g);
                        h[2 * ((size_t)nf8_img_A3CH(wp) + ((m * 4 + q4) * 8 + g * 4 + j) * 2) + r] =
                            cin < w ? to_half_w3(W3[((size_t)8 * w + cin) * 4 + j], j) : 0;
                    }
    }
}

// Variant B of the fp16-CNN GEMM layout (NF9_*): the same values, one contiguous slab per channel tile.
void relayout_coupling_gemm16b(const float *v1, int w, int wp, float *out)
{
    std::vector<float> a(nf8_cpl_size(wp));
    relayout_coupling_gemm16(v1, w, wp, a.data());
    const int MT = wp / 32, KS = wp / 16;
    memcpy(out, a.data(), (size_t)(NF8_CPL_IMG + MT * 544) * sizeof(float));      // E, S, A1H, B1 (same offsets)
    const float *ia = a.data() + NF8_CPL_IMG;
    float *io = out + NF8_CPL_IMG;
    for (int m = 0; m < MT; ++m) {
        float *sl = io + nf9_img_SLAB(wp) + (size_t)m * nf9_slab_dwords(wp);
        memcpy(sl, ia + nf8_img_A2H(wp) + (size_t)m * KS * 256, (size_t)KS * 256 * sizeof(float));
        memcpy(sl + nf9_slab_A3H(wp), ia + nf8_img_A3H(wp) + (size_t)m * 512, 512 * sizeof(float));
        memcpy(sl + nf9_slab_A3CH(wp), ia + nf8_img_A3CH(wp) + (size_t)m * 64, 64 * sizeof(float));
        memcpy(sl + nf9_slab_B2(wp), ia + nf8_img_B2(wp) + (size_t)m * 32, 32 * sizeof(float));
    }
}

// Width-16 re-layout (nf_device.h, NF6_*; `w` = 16, or 8 zero-padded): fetch order of v_mfma_f32_16x16x4_f32.
void relayout_coupling_wide16(const float *v1, int w, float *out)
{
    const double k2 = 2.0 * 1.4426950408889634, log2e = 1.4426950408889634;
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double e = v1[nf_cpl_off_E(w) + 4 * m + j];
            out[NF4_CPL_E + 4 * m + j] = (float)(j >= 2 ? e * k2 : e);
        }
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF4_CPL_S + 0] = (float)sc;
    out[NF4_CPL_S + 1] = (float)(sc * log2e);
    out[NF4_CPL_S + 2] = (float)(-2.0 * sc * log2e);
    out[NF4_CPL_S + 3] = 0.0f;
    float *img = out + NF4_CPL_IMG;
    memset(img, 0, NF6_IMG_SIZE * sizeof(float));
    const float *W1 = v1 + nf_cpl_off_W1(w), *B1 = v1 + nf_cpl_off_B1(w), *W2 = v1 + nf_cpl_off_W2(w);
    const float *B2 = v1 + nf_cpl_off_B2(w), *W3 = v1 + nf_cpl_off_W3(w);
    static const int tapA[4] = {0 * 3 + 0, 1 * 3 + 0, 2 * 3 + 0, 0 * 3 + 1}, tapB[4] = {0 * 3 + 2, 1 * 3 + 2, 2 * 3 + 2, 2 * 3 + 1};
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, g = l >> 4;
        for (int s = 0; s < 5; ++s) {
            const int kk = 4 * s + g;
            const float v = (kk < 18 && i < w) ? W1[kk * w + i] : 0.0f;   // kk = tap * 2 + ch
            if (s < 4) img[NF6_IMG_A1 + l * 4 + s] = v;
            else img[NF6_IMG_A1 + 256 + l] = v;
        }
        for (int s = 0; s < 4; ++s) {
            const int cin = 4 * g + s;
            img[NF6_IMG_A2 + l * 4 + s] = (cin < w && i < w) ? W2[cin * w + i] : 0.0f;
            const int gp = i >> 2, j = i & 3;
            const double wa = cin < w ? W3[(tapA[gp] * w + cin) * 4 + j] : 0.0, wb = cin < w ? W3[(tapB[gp] * w + cin) * 4 + j] : 0.0;
            img[NF6_IMG_A3A + l * 4 + s] = (float)(j >= 2 ? wa * k2 : wa);
            img[NF6_IMG_A3B + l * 4 + s] = (float)(j >= 2 ? wb * k2 : wb);
        }
    }
    for (int g = 0; g < 4; ++g)
        for (int v = 0; v < 4; ++v) {
            const int ch = 4 * g + v;
            img[NF6_IMG_B1 + g * 4 + v] = ch < w ? B1[ch] : 0.0f;
            img[NF6_IMG_B2 + g * 4 + v] = ch < w ? B2[ch] : 0.0f;
            for (int j = 0; j < 4; ++j) {   // here v plays the K step s
                const double wv = ch < w ? W3[(4 * w + ch) * 4 + j] : 0.0;
                img[NF6_IMG_A3C + (g * 4 + j) * 4 + v] = (float)(j >= 2 ? wv * k2 : wv);
            }
        }
}

// fp16 variant of the wide layout (nf_device.h, NF5_*): folded weights rounded to half, in the fetch order of
// v_mfma_f32_32x32x16_f16 (8 halves per lane and instruction) / v_mfma_f32_4x4x4_16b_f16 (centre tap).
void relayout_coupling_wide32_fp16(const float *v1, int w, float *out)
{
    const double k2 = 2.0 * 1.4426950408889634, log2e = 1.4426950408889634;
    for (int m = 0; m < 16; ++m)
        for (int j = 0; j < 4; ++j) {
            const double e = v1[nf_cpl_off_E(w) + 4 * m + j];
            out[NF4_CPL_E + 4 * m + j] = (float)(j >= 2 ? e * k2 : e);
        }
    const double sc = v1[nf_cpl_off_S(w)];
    out[NF4_CPL_S + 0] = (float)sc;
    out[NF4_CPL_S + 1] = (float)(sc * log2e);
    out[NF4_CPL_S + 2] = (float)(-2.0 * sc * log2e);
    out[NF4_CPL_S + 3] = 0.0f;
    float *img = out + NF4_CPL_IMG;
    memset(img, 0, NF5_IMG_SIZE * sizeof(float));
    uint16_t *h = reinterpret_cast<uint16_t *>(img);   // half index = 2 * dword index
    const float *W1 = v1 + nf_cpl_off_W1(w), *B1 = v1 + nf_cpl_off_B1(w), *W2 = v1 + nf_cpl_off_W2(w);
    const float *B2 = v1 + nf_cpl_off_B2(w), *W3 = v1 + nf_cpl_off_W3(w);
    static const int tap_of[4][2] = {{0 * 3 + 0, 2 * 3 + 0}, {0 * 3 + 2, 2 * 3 + 2}, {0 * 3 + 1, 2 * 3 + 1}, {1 * 3 + 0, 1 * 3 + 2}};
    for (int l = 0; l < 64; ++l) {
        const int i = l & 31, g = l >> 5;
        for (int q = 0; q < 8; ++q) {
            // l_1, instruction 0: taps 4g .. 4g+3; instruction 1: tap 8 on lane half 0
            const int tap = 4 * g + (q >> 1), ch = q & 1;
            h[2 * (NF5_IMG_A1H + (0 * 64 + l) * 4) + q] = i < w ? to_half(W1[(tap * 2 + ch) * w + i]) : 0;
            h[2 * (NF5_IMG_A1H + (1 * 64 + l) * 4) + q] = (i < w && g == 0 && q < 2) ? to_half(W1[(8 * 2 + q) * w + i]) : 0;
            for (int m = 0; m < 2; ++m) {
                const int cin = nf4_chan(8 * m + q, g);
                h[2 * (NF5_IMG_A2H + (m * 64 + l) * 4) + q] = (cin < w && i < w) ? to_half(W2[cin * w + i]) : 0;
                const int a = i >> 3, gp = (i >> 2) & 1, j = i & 3;
                h[2 * (NF5_IMG_A3H + (m * 64 + l) * 4) + q] = cin < w ? to_half_w3(W3[(tap_of[a][gp] * w + cin) * 4 + j], j) : 0;
            }
        }
    }
    for (int g = 0; g < 2; ++g)
        for (int v = 0; v < 16; ++v) {
            const int ch = nf4_chan(v, g);
            img[NF5_IMG_B1 + g * 16 + v] = ch < w ? B1[ch] : 0.0f;
            img[NF5_IMG_B2 + g * 16 + v] = ch < w ? B2[ch] : 0.0f;
        }
    for (int q = 0; q < 4; ++q)
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 4; ++r) {
                    const int cin = nf4_chan(4 * q + r, g);
                    h[2 * (NF5_IMG_A3CH + (q * 8 + g * 4 + j) * 2) + r] = cin < w ? to_half_w3(W3[(4 * w + cin) * 4 + j], j) : 0;   // centre tap
                }
}

// ---- sdn5 host scalars (cond_utils.py:205-239) -------------------------------
int sdn5_scalars(const float *sp, const nf_cond *cond, double out[2])
{
    if (!cond) return fail(NF_EINVAL, "model has an SDN5 layer but cond is NULL");
    const double beta1_v = sp[0], beta2_v = sp[1];
    const float *gain_params = sp + 2, *cam_params = sp + 7;
    const double c_i = sp[22];
    int cam_idx = -1;
    for (int i = 0; i < 5; ++i)
        if ((float)i == cond->cam) cam_idx = i;
    if (cam_idx < 0) return fail(NF_ECOND, "unknown camera id %g (expected 0..4 = IP,GP,S6,N6,G4)", (double)cond->cam);
    double cp[3];
    for (int r = 0; r < 3; ++r) cp[r] = exp(c_i * (double)cam_params[r * 5 + cam_idx]);
    static const float iso_vals[5] = {100.f, 400.f, 800.f, 1600.f, 3200.f};
    double g = 0.0;   // unknown ISO -> empty one-hot -> reduce_sum = 0 (cond_utils.py:227-229)
    for (int i = 0; i < 5; ++i)
        if (iso_vals[i] == cond->iso) g = gain_params[i];
    const double gain = exp(c_i * g * cp[2]) * (double)cond->iso;
    const double beta1 = exp(c_i * beta1_v * cp[0]);
    const double beta2 = exp(c_i * beta2_v * cp[1]);
    out[0] = beta1 / gain;
    out[1] = beta2;
    return NF_OK;
}

struct CondLayer {
    int kind;                  // NF_LAYER_SDN5 / SDN4 / SDN / GAIN
    std::vector<float> p;      // its raw parameters
};

double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// per-ISO tables of the Ex1-Ex3 layers: nested tf.cond on iso == 100, 400, 800, 1600, 3200 whose last branch is
// the ISO-800 entry (cond_utils.py:69-88)
int iso_table_index(float iso)
{
    static const float iso_vals[5] = {100.f, 400.f, 800.f, 1600.f, 3200.f};
    for (int i = 0; i < 5; ++i)
        if (iso_vals[i] == iso) return i;
    return 2;
}

inline bool is_gain_kind(int k) { return k == NF_LAYER_GAIN || k == NF_LAYER_GAIN1 || k == NF_LAYER_GAIN2 || k == NF_LAYER_GAIN3; }

// Per-call scalars of one conditional layer -> (a, b): SDN kinds: scale^2 = a*y + b; GAIN: scale = a.
int cond_scalars(const CondLayer &L, const nf_cond *cond, double out[2])
{
    switch (L.kind) {
    case NF_LAYER_SDN5: return sdn5_scalars(L.p.data(), cond, out);
    case NF_LAYER_SDN4: {   // cond_utils.py:178-202 (c = 1)
        if (!cond) return fail(NF_EINVAL, "model has an SDN4 layer but cond is NULL");
        static const float iso_vals[5] = {100.f, 400.f, 800.f, 1600.f, 3200.f};
        double g = 0.0;
        for (int i = 0; i < 5; ++i)
            if (iso_vals[i] == cond->iso) g = L.p[2 + i];
        const double gain = exp(g) * (double)cond->iso;
        out[0] = exp((double)L.p[0]) / gain;
        out[1] = exp((double)L.p[1]);
        return NF_OK;
    }
    case NF_LAYER_SDN:      // cond_utils.py:41-52
        out[0] = sigmoid(L.p[0]);
        out[1] = sigmoid(L.p[1]);
        return NF_OK;
    case NF_LAYER_GAIN:     // cond_utils.py:319-330 with gain = iso (AffineCouplingGain.py:52,113)
        if (!cond) return fail(NF_EINVAL, "model has a GAIN layer but cond is NULL");
        out[0] = sigmoid(L.p[0]) * (double)cond->iso + sigmoid(L.p[1]);
        out[1] = 0.0;
        if (!(out[0] > 0.0)) return fail(NF_EINVAL, "gain scale must be > 0");
        return NF_OK;
    case NF_LAYER_SDN1:     // cond_utils.py:55-98
    case NF_LAYER_SDN2:     // cond_utils.py:101-138
    case NF_LAYER_SDN3: {   // cond_utils.py:141-175
        if (!cond) return fail(NF_EINVAL, "model has an SDN layer with a per-ISO gain table but cond is NULL");
        const double c = L.kind == NF_LAYER_SDN1 ? 1e-2 : 1e-1;
        const double gain = exp(c * (double)L.p[2 + iso_table_index(cond->iso)]) * (double)cond->iso;
        const double b1 = sigmoid(L.p[0]), b2 = sigmoid(L.p[1]);
        if (L.kind == NF_LAYER_SDN1) {          // sqrt(b1 y / r_gain + b2)
            out[0] = b1 / gain;
            out[1] = b2;
        } else if (L.kind == NF_LAYER_SDN2) {   // sqrt(gain (b1 y / gain + b2))
            out[0] = gain * (b1 / gain);
            out[1] = gain * b2;
        } else {                                // gain sqrt(b1 y / gain + b2)
            out[0] = gain * gain * (b1 / gain);
            out[1] = gain * gain * b2;
        }
        return NF_OK;
    }
    case NF_LAYER_SDN6: {   // cond_utils.py:242-276: one camera parameter, applied to the gain exponent only
        if (!cond) return fail(NF_EINVAL, "model has an SDN6 layer but cond is NULL");
        const double c_i = L.p[12];
        int cam_idx = -1;
        for (int i = 0; i < 5; ++i)
            if ((float)i == cond->cam) cam_idx = i;
        if (cam_idx < 0) return fail(NF_ECOND, "unknown camera id %g (expected 0..4 = IP,GP,S6,N6,G4)", (double)cond->cam);
        const double cp = exp(c_i * (double)L.p[7 + cam_idx]);
        static const float iso_vals[5] = {100.f, 400.f, 800.f, 1600.f, 3200.f};
        double g = 0.0;   // unknown ISO -> empty one-hot -> 0
        for (int i = 0; i < 5; ++i)
            if (iso_vals[i] == cond->iso) g = L.p[2 + i];
        const double gain = exp(c_i * g * cp) * (double)cond->iso;
        out[0] = exp(c_i * (double)L.p[0]) / gain;
        out[1] = exp(c_i * (double)L.p[1]);
        return NF_OK;
    }
    case NF_LAYER_GAIN1:    // cond_utils.py:333-350 with gain = iso
        if (!cond) return fail(NF_EINVAL, "model has a GAIN1 layer but cond is NULL");
        out[0] = exp(1e-5 * (double)L.p[0]) * (double)cond->iso + exp(1e-5 * (double)L.p[1]);
        out[1] = 0.0;
        return NF_OK;
    case NF_LAYER_GAIN2:    // cond_utils.py:353-392
    case NF_LAYER_GAIN3:    // cond_utils.py:395-429
        if (!cond) return fail(NF_EINVAL, "model has a per-ISO GAIN layer but cond is NULL");
        if (L.kind == NF_LAYER_GAIN2) out[0] = exp(1e-1 * (double)L.p[iso_table_index(cond->iso)]) * (double)cond->iso;
        else out[0] = exp(1e-5 * (double)L.p[iso_table_index(cond->iso)]);
        out[1] = 0.0;
        if (!(out[0] > 0.0)) return fail(NF_EINVAL, "gain scale must be > 0");
        return NF_OK;
    }
    return fail(NF_EINVAL, "bad conditional layer");
}

// ---- program construction -----------------------------------------------------
struct Built {
    std::vector<CondLayer> cond;   // conditioning slots, in the order the ops reference them
    NfProgram prog;
    std::vector<float> block;
    NfProgram prog2;             // matrix-core (MFMA) layout, width 4 only
    std::vector<float> block2;   // empty when unavailable
    NfProgram prog3;             // fp16-CNN layout (NF_CFG_FP16_CNN), width 4 only
    std::vector<float> block3;
    bool fp16_big = false;       // block3 is the NF11_* layout (v_mfma_f32_16x16x32_f16) instead of NF3_* (4x4x4)
    int raw_width = 4;           // coupling width of the model's variables; prog.width is the (zero-padded) width the kernels run
    NfProgram prog4;             // wide-CNN layout (NF4_*), width 32 (8 / 16 zero-padded on large patches)
    std::vector<float> block4;
    NfProgram prog5;             // wide-CNN fp16 layout (NF5_*): NF_CFG_FP16_CNN at widths 8 / 16 / 32
    std::vector<float> block5;
    NfProgram prog6;             // width-16 layout (NF6_*)
    std::vector<float> block6;
    NfProgram prog7;             // GEMM layout (NF7_*): widths 33 .. 512, zero-padded to 64 / 128 / 256 / 512
    std::vector<float> block7;
    bool gemm_b = false;         // block7 is in the variant-B layout (NF10_*: widths <= 128, weights resident in LDS)
    NfProgram prog8;             // fp16-CNN GEMM layout (NF8_*): NF_CFG_FP16_CNN at widths 33 .. 512
    std::vector<float> block8;
    bool gemm16_b = false;       // block8 is in the variant-B layout (NF9_*: widths <= 128)
    double ld_const = 0.0;
    bool has_sdn = false;      // some op reads the clean image y
    // images beyond 64x64 (nf_device.h, NF_K_TILED): the kernels run on overlapping tile_h x tile_w tiles, the program cut
    // into segments (ops [op0, op1) each, its own halo and tile counts)
    bool tiled = false;
    int tile_h = 0, tile_w = 0;
    struct TileSeg {
        int op0, op1, halo, ny, nx;
    };
    std::vector<TileSeg> segs;
};

// Cut a tiled program into segments (nf_device.h): the couplings dealt as evenly as possible to S consecutive groups, a
// segment ending right after its last coupling (the pointwise ops behind the last coupling stay with the last segment), S
// chosen for the least estimated work, in units of one coupling on one tile: tiles of a segment x (its couplings + 0.75 for the
// tile's load / store / set-up and the pointwise layers) + 1 per 4096 image pixels and segment boundary (32 B per pixel
// through HBM) — fitted to tools/time_large_patches.py at 256^2 and 1024^2.  NF_TILE_SEGMENTS=<n> forces S (A/B and test aid).
static int plan_tile_segments(const NfProgram &prog, int H, int W, int th, int tw, std::vector<Built::TileSeg> &segs)
{
    std::vector<int> cpl;
    for (int i = 0; i < prog.n_ops; ++i)
        if (prog.ops[i].type == NF_OP_COUPLING_FWD || prog.ops[i].type == NF_OP_COUPLING_REV) cpl.push_back(i);
    const int n_cpl = (int)cpl.size();
    const char *e = getenv("NF_TILE_SEGMENTS");
    const int forced = e ? atoi(e) : 0;
    double best = 0.0;
    segs.clear();
    const int s_max = n_cpl < 1 ? 1 : n_cpl < NF_MAX_TILE_SEGS ? n_cpl : NF_MAX_TILE_SEGS;
    const int s_only = forced > 0 ? (forced < s_max ? forced : s_max) : 0;
    for (int S = 1; S <= s_max; ++S) {
        if (s_only && S != s_only) continue;
        std::vector<Built::TileSeg> cand;
        double cost = 0.0;
        bool ok = true;
        int c0 = 0, op0 = 0;
        for (int sgm = 0; sgm < S; ++sgm) {
            const int nc = n_cpl / S + (sgm < n_cpl % S ? 1 : 0);
            Built::TileSeg t;
            t.op0 = op0;
            t.op1 = sgm == S - 1 ? prog.n_ops : cpl[c0 + nc - 1] + 1;
            t.halo = 2 * nc;
            if ((H > th || W > tw) && 64 - 2 * t.halo < 8) ok = false;   // no core left in a 64-pixel tile
            if (!ok) break;
            t.ny = nf_tile_count(H, th, t.halo);
            t.nx = nf_tile_count(W, tw, t.halo);
            cost += (double)t.ny * t.nx * (nc + 0.75);
            if (sgm > 0) cost += (double)H * W / 4096.0;   // the tensor between two segments: ~ one tile-coupling per 4096 pixels
            cand.push_back(t);
            c0 += nc;
            op0 = t.op1;
        }
        if (!ok) continue;
        if (segs.empty() || cost < best) {
            best = cost;
            segs.swap(cand);
        }
    }
    if (segs.empty()) return fail(NF_EINVAL, "patches beyond 64x64: %d coupling layers cannot be cut into at most %d tiled segments", n_cpl, NF_MAX_TILE_SEGS);
    return NF_OK;
}


int build_program(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params,
                  int direction, Built &out)
{
    if (!cfg || !layers || !params) return fail(NF_EINVAL, "null argument");
    if (cfg->channels != kC) return fail(NF_EINVAL, "channels must be 4 (packed raw), got %d", cfg->channels);
    if (cfg->height < 1 || cfg->width < 1 || cfg->height > NF_MAX_IMAGE_SIDE || cfg->width > NF_MAX_IMAGE_SIDE)
        return fail(NF_EINVAL, "patch size %dx%d unsupported (1 .. %d per side)", cfg->height, cfg->width, NF_MAX_IMAGE_SIDE);
    // one workgroup holds up to 64x64 pixels; larger images are evaluated as overlapping tiles of th x tw pixels, and every
    // kernel-shape decision below is about the tile
    const int th = cfg->height < 64 ? cfg->height : 64, tw = cfg->width < 64 ? cfg->width : 64;
    out.tiled = cfg->height > 64 || cfg->width > 64;
    out.tile_h = th;
    out.tile_w = tw;
    out.segs.clear();
    if (cfg->n_layers < 1) return fail(NF_EINVAL, "n_layers must be >= 1");
    if (cfg->flags & ~NF_CFG_FP16_CNN) return fail(NF_EINVAL, "nf_config.flags has unknown bits set");
    const double HW = (double)cfg->height * cfg->width;

    // intermediate list in NLL order
    struct Item {
        int type;               // NF_OP_* of the NLL direction
        std::vector<float> blk; // folded block (fwd)
        Mat4 A, Ainv;
        double scale = 1.0;     // gain value for SCALE items
        int w = 0;
        int slot = 0;           // conditioning slot of SDN / SCALE_COND items
    };
    std::vector<Item> items;
    int width = 0, raw_width = 0;   // width of the kernels' layouts / of the model's variables (differ when zero-padded)
    out.ld_const = 0.0;
    out.has_sdn = false;
    out.cond.clear();
    for (int li = 0; li < cfg->n_layers; ++li) {
        const nf_layer_desc &L = layers[li];
        const int64_t cnt = layer_param_count(L.type, L.width);
        if (cnt < 0) return fail(NF_EINVAL, "layer %d: unknown type %d / width %d", li, L.type, L.width);
        if (L.param_offset < 0 || (uint64_t)L.param_offset + (uint64_t)cnt > n_params)
            return fail(NF_EINVAL, "layer %d: parameters [%lld, +%lld) exceed n_params=%zu", li,
                        (long long)L.param_offset, (long long)cnt, n_params);
        const float *p = params + L.param_offset;
        Item it;
        switch (L.type) {
        case NF_LAYER_CONV1X1: {
            double lad;
            fold_conv1x1(p, it.A, it.Ainv, lad);
            it.type = NF_OP_MIX;
            out.ld_const += HW * lad;                       // layers.py:129-130
            break;
        }
        case NF_LAYER_CONV1X1_NONE:
        case NF_LAYER_CONV1X1_LU2: {
            double lad;
            const bool ok = L.type == NF_LAYER_CONV1X1_NONE ? fold_conv1x1_none(p, it.A, it.Ainv, lad)
                                                            : fold_conv1x1_lu2(p, it.A, it.Ainv, lad);
            if (!ok) return fail(NF_EINVAL, "layer %d: the 1x1 matrix is singular", li);
            it.type = NF_OP_MIX;
            out.ld_const += HW * lad;
            break;
        }
        case NF_LAYER_PERMUTE:                              // tfb.Permute(channels reversed), noise_flow_model.py:80-84
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) it.A.m[r][c] = it.Ainv.m[r][c] = (r + c == 3) ? 1.0 : 0.0;
            it.type = NF_OP_MIX;
            break;
        case NF_LAYER_COUPLING: {
            if (L.width < 1 || L.width > 512)
                return fail(NF_EINVAL, "layer %d: coupling width %d unsupported (1 .. 512)", li, L.width);
            // layers.py:452-498 takes any width; the kernels exist for 4 / 8 / 16 / 32 and, zero-padded inside their layouts, 33 .. 512.
            // Widths in between run on the next kernel up, zero-padded HERE: a hidden channel with zero weights and zero bias is
            // relu(0) = 0 in both hidden layers and contributes nothing to l_2 / l_last — exact.
            const int wk = L.width > 32 ? L.width : L.width <= 4 ? 4 : L.width <= 8 ? 8 : L.width <= 16 ? 16 : 32;
            if (L.width > 32 && !nf_gemm_shape_ok(th, tw))
                return fail(NF_EINVAL, "layer %d: coupling width %d covers patches of up to %d pixels (%dx%d given)", li, L.width,
                            NF7_MAX_PIXELS, cfg->height, cfg->width);
            if (raw_width && raw_width != L.width) return fail(NF_EINVAL, "all coupling layers must share one width");
            raw_width = L.width;
            width = wk;
            it.type = NF_OP_COUPLING_FWD;
            it.w = wk;
            it.blk.assign(nf_cpl_size(wk), 0.0f);
            if (wk == L.width) {
                fold_coupling(p, L.width, it.blk.data());
            } else {
                const int w = L.width;
                std::vector<float> f(nf_cpl_size(w));
                fold_coupling(p, w, f.data());
                float *o = it.blk.data();
                memcpy(o + nf_cpl_off_E(wk), f.data() + nf_cpl_off_E(w), 64 * sizeof(float));
                for (int tap = 0; tap < 9; ++tap) {
                    memcpy(o + nf_cpl_off_W3(wk) + tap * wk * 4, f.data() + nf_cpl_off_W3(w) + tap * w * 4, (size_t)w * 4 * sizeof(float));
                    for (int c = 0; c < 2; ++c)
                        memcpy(o + nf_cpl_off_W1(wk) + (tap * 2 + c) * wk, f.data() + nf_cpl_off_W1(w) + (tap * 2 + c) * w, (size_t)w * sizeof(float));
                }
                memcpy(o + nf_cpl_off_B1(wk), f.data() + nf_cpl_off_B1(w), (size_t)w * sizeof(float));
                memcpy(o + nf_cpl_off_B2(wk), f.data() + nf_cpl_off_B2(w), (size_t)w * sizeof(float));
                for (int i = 0; i < w; ++i)
                    memcpy(o + nf_cpl_off_W2(wk) + i * wk, f.data() + nf_cpl_off_W2(w) + i * w, (size_t)w * sizeof(float));
                memcpy(o + nf_cpl_off_S(wk), f.data() + nf_cpl_off_S(w), 4 * sizeof(float));
            }
            break;
        }
        case NF_LAYER_SDN5:
        case NF_LAYER_SDN4:
        case NF_LAYER_SDN:
        case NF_LAYER_GAIN:
        case NF_LAYER_SDN1:
        case NF_LAYER_SDN2:
        case NF_LAYER_SDN3:
        case NF_LAYER_SDN6:
        case NF_LAYER_GAIN1:
        case NF_LAYER_GAIN2:
        case NF_LAYER_GAIN3: {
            if (out.cond.size() >= 4) return fail(NF_EINVAL, "at most 4 conditional (sdn/gain) layers per model");
            it.type = is_gain_kind(L.type) ? NF_OP_SCALE_COND : NF_OP_SDN_DIV;
            it.slot = (int)out.cond.size();
            CondLayer c;
            c.kind = L.type;
            c.p.assign(p, p + cnt);
            if (L.type == NF_LAYER_SDN5) c.p.resize(23);
            out.cond.push_back(c);
            if (!is_gain_kind(L.type)) out.has_sdn = true;
            break;
        }
        case NF_LAYER_GAIN4:
            if (!(p[0] > 0.0f)) return fail(NF_EINVAL, "layer %d: gain_val must be > 0", li);
            it.type = NF_OP_SCALE;
            it.scale = (double)p[0];
            out.ld_const -= HW * kC * log((double)p[0]);   // AffineCouplingGainEx4.py:114-127
            break;
        }
        items.push_back(it);
    }

    // Fold every gain into a neighbouring 1x1 matrix (NLL: z/g then z@A == z@(A/g)).
    for (size_t i = 0; i < items.size(); ++i) {
        if (items[i].type != NF_OP_SCALE) continue;
        Item *tgt = nullptr;
        if (i + 1 < items.size() && items[i + 1].type == NF_OP_MIX) tgt = &items[i + 1];
        else if (i > 0 && items[i - 1].type == NF_OP_MIX) tgt = &items[i - 1];
        if (!tgt) continue;
        const double g = items[i].scale;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                tgt->A.m[r][c] /= g;
                tgt->Ainv.m[r][c] *= g;
            }
        items[i].type = 0;   // removed
    }

    std::vector<Item *> order;
    for (auto &it : items)
        if (it.type != 0) order.push_back(&it);
    if (direction == 1) std::vector<Item *>(order.rbegin(), order.rend()).swap(order);
    if (order.size() > NF_MAX_OPS) return fail(NF_EINVAL, "too many layers (%zu > %d)", order.size(), NF_MAX_OPS);

    memset(&out.prog, 0, sizeof(out.prog));
    out.prog.width = width ? width : 4;
    out.raw_width = raw_width ? raw_width : 4;
    out.block.clear();
    for (Item *it : order) {
        NfOp &op = out.prog.ops[out.prog.n_ops++];
        op.off = (int32_t)out.block.size();
        switch (it->type) {
        case NF_OP_MIX: {
            op.type = NF_OP_MIX;
            const Mat4 &M = direction == 0 ? it->A : it->Ainv;
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) out.block.push_back((float)M.m[r][c]);
            break;
        }
        case NF_OP_COUPLING_FWD:
            op.type = direction == 0 ? NF_OP_COUPLING_FWD : NF_OP_COUPLING_REV;
            out.block.insert(out.block.end(), it->blk.begin(), it->blk.end());
            break;
        case NF_OP_SDN_DIV:
            op.type = direction == 0 ? NF_OP_SDN_DIV : NF_OP_SDN_MUL;
            op.off = it->slot;
            break;
        case NF_OP_SCALE_COND:
            op.type = NF_OP_SCALE_COND;
            op.off = it->slot;
            break;
        case NF_OP_SCALE: {
            op.type = NF_OP_SCALE;
            const double s = direction == 0 ? 1.0 / it->scale : it->scale;
            out.block.push_back((float)s);
            out.block.push_back(0.f);
            out.block.push_back(0.f);
            out.block.push_back(0.f);
            break;
        }
        }
    }
    if (out.block.empty()) out.block.assign(4, 0.0f);

    // matrix-core re-layout (same op sequence, j-major weights), when the model qualifies
    out.block2.clear();
    memset(&out.prog2, 0, sizeof(out.prog2));
    if (out.prog.width == 4) {
        out.prog2.width = 4;
        for (int i = 0; i < out.prog.n_ops; ++i) {
            const NfOp &src = out.prog.ops[i];
            NfOp &dst = out.prog2.ops[out.prog2.n_ops++];
            dst.type = src.type;
            dst.off = (int32_t)out.block2.size();
            const float *v1 = out.block.data() + src.off;
            if (src.type == NF_OP_MIX) {
                for (int j = 0; j < 4; ++j)
                    for (int c = 0; c < 4; ++c) out.block2.push_back(v1[c * 4 + j]);
            } else if (src.type == NF_OP_COUPLING_FWD || src.type == NF_OP_COUPLING_REV) {
                out.block2.resize(out.block2.size() + NF2_CPL_SIZE);
                relayout_coupling_v2(v1, out.block2.data() + dst.off);
            } else if (src.type == NF_OP_SCALE) {
                out.block2.insert(out.block2.end(), v1, v1 + 4);
            } else {
                dst.off = src.off;   // conditioning slot
            }
        }
        if (out.block2.empty()) out.block2.assign(4, 0.0f);
        if (out.block2.size() > NF2_MAX_FLOATS) out.block2.clear();   // too large for LDS: scalar path only
    }
    out.block4.clear();
    memset(&out.prog4, 0, sizeof(out.prog4));
    // width 32: the product path; width 8 only where the scalar-weight kernel's LDS tile does not fit (patches larger
    // than 48x48): zero-padded to 32 channels.  Width 16 has its own kernel (prog6).
    {
        const size_t tile_px = ((size_t)(th + 2) * (tw + 2) + 1) & ~(size_t)1;
        const bool scalar_fits = sizeof(float) * (tile_px * (2 + (size_t)out.prog.width) + 64) <= 160 * 1024;
        if (out.prog.width == 32 || (out.prog.width == 8 && !scalar_fits)) out.prog4.width = 32;
        // tiled images (nf_device.h): width 8 runs zero-padded on the width-32 kernel; width 16 on its own (fp32; its fp16-CNN
        // mode is the width-32 fp16 kernel's, tiled or not)
        if (out.tiled && out.prog.width == 8) out.prog4.width = 32;
    }
    if (out.prog4.width == 32) {
        for (int i = 0; i < out.prog.n_ops; ++i) {
            const NfOp &src = out.prog.ops[i];
            NfOp &dst = out.prog4.ops[out.prog4.n_ops++];
            dst.type = src.type;
            dst.off = (int32_t)out.block4.size();
            const float *v1 = out.block.data() + src.off;
            if (src.type == NF_OP_MIX) {
                out.block4.insert(out.block4.end(), v1, v1 + 16);
            } else if (src.type == NF_OP_COUPLING_FWD || src.type == NF_OP_COUPLING_REV) {
                out.block4.resize(out.block4.size() + NF4_CPL_SIZE);
                relayout_coupling_wide32(v1, out.prog.width, out.block4.data() + dst.off);
            } else if (src.type == NF_OP_SCALE) {
                out.block4.insert(out.block4.end(), v1, v1 + 4);
            } else {
                dst.off = src.off;   // conditioning slot
            }
        }
        if (out.block4.empty()) out.block4.assign(4, 0.0f);
    }
    out.block6.clear();
    memset(&out.prog6, 0, sizeof(out.prog6));
    if (out.prog.width == 16) {
        out.prog6.width = 16;
        for (int i = 0; i < out.prog.n_ops; ++i) {
            const NfOp &src = out.prog.ops[i];
            NfOp &dst = out.prog6.ops[out.prog6.n_ops++];
            dst.type = src.type;
            dst.off = (int32_t)out.block6.size();
            const float *v1 = out.block.data() + src.off;
            if (src.type == NF_OP_MIX) {
                out.block6.insert(out.block6.end(), v1, v1 + 16);
            } else if (src.type == NF_OP_COUPLING_FWD || src.type == NF_OP_COUPLING_REV) {
                out.block6.resize(out.block6.size() + NF6_CPL_SIZE);
                relayout_coupling_wide16(v1, out.prog.width, out.block6.data() + dst.off);
            } else if (src.type == NF_OP_SCALE) {
                out.block6.insert(out.block6.end(), v1, v1 + 4);
            } else {
                dst.off = src.off;   // conditioning slot
            }
        }
        if (out.block6.empty()) out.block6.assign(4, 0.0f);
    }
    out.block8.clear();
    memset(&out.prog8, 0, sizeof(out.prog8));
    if (out.prog.width > 32 && (cfg->flags & NF_CFG_FP16_CNN)) {
        const int wp = nf7_pad_width(out.prog.width);
        out.prog8.width = wp;
        {   // variant B (weights resident in LDS, pixel tiles per wavefront) where its slabs fit: widths <= 128; NF_GEMM16=a: A/B aid
            const char *e = getenv("NF_GEMM16");
            out.gemm16_b = wp <= 128 && !(e && e[0] == 'a');
        }
        for (int i = 0; i < out.prog.n_ops; ++i) {
            const NfOp &src = out.prog.ops[i];
            NfOp &dst = out.prog8.ops[out.prog8.n_ops++];
            dst.type = src.type;
            dst.off = (int32_t)out.block8.size();
            const float *v1 = out.block.data() + src.off;
            if (src.type == NF_OP_MIX) {
                out.block8.insert(out.block8.end(), v1, v1 + 16);
            } else if (src.type == NF_OP_COUPLING_FWD || src.type == NF_OP_COUPLING_REV) {
                out.block8.resize(out.block8.size() + (out.gemm16_b ? nf9_cpl_size(wp) : nf8_cpl_size(wp)));
                if (out.gemm16_b) relayout_coupling_gemm16b(v1, out.prog.width, wp, out.block8.data() + dst.off);
                else relayout_coupling_gemm16(v1, out.prog.width, wp, out.block8.data() + dst.off);
            } else if (src.type == NF_OP_SCALE) {
                out.block8.insert(out.block8.end(), v1, v1 + 4);
            } else {
                dst.off = src.off;   // conditioning slot
            }
        }
        if (out.block8.empty()) out.block8.assign(4, 0.0f);
    }
    out.block7.clear();
    memset(&out.prog7, 0, sizeof(out.prog7));
    if (out.prog.width > 32 && !(cfg->flags & NF_CFG_FP16_CNN)) {
        const int wp = nf7_pad_width(out.prog.width);
        out.prog7.width = wp;
        {   // variant B where its slabs fit beside the patch's tiles; NF_GEMM=a: A/B aid
            const char *e = getenv("NF_GEMM");
            out.gemm_b = nf_gemmb_shape_ok(wp, th, tw) && !(e && e[0] == 'a');
        }
        for (int i = 0; i < out.prog.n_ops; ++i) {
            const NfOp &src = out.prog.ops[i];
            NfOp &dst = out.prog7.ops[out.prog7.n_ops++];
            dst.type = src.type;
            dst.off = (int32_t)out.block7.size();
            const float *v1 = out.block.data() + src.off;
            if (src.type == NF_OP_MIX) {
                out.block7.insert(out.block7.end(), v1, v1 + 16);
            } else if (src.type == NF_OP_COUPLING_FWD || src.type == NF_OP_COUPLING_REV) {
                out.block7.resize(out.block7.size() + (out.gemm_b ? nf10_cpl_size(wp) : nf7_cpl_size(wp)));
                if (out.gemm_b) relayout_coupling_gemmb(v1, out.prog.width, wp, out.block7.data() + dst.off);
                else relayout_coupling_gemm(v1, out.prog.width, wp, out.block7.data() + dst.off);
            } else if (src.type == NF_OP_SCALE) {
                out.block7.insert(out.block7.end(), v1, v1 + 4);
            } else {
                dst.off = src.off;   // conditioning slot
            }
        }
        if (out.block7.empty()) out.block7.assign(4, 0.0f);
    }
    out.block5.clear();
    memset(&out.prog5, 0, sizeof(out.prog5));
    // width 4 has its own fp16 kernel for the two full shapes (nf_kernels.hip); on any other shape it runs here, zero-padded
    // to 32 channels (exact: same rounding points, a padded channel is identically zero)
    // NF_H16=4x4 (read at nf_create) keeps the v_mfma_f32_4x4x4_16b_f16 formulation of the width-4 fp16 kernel — an A/B aid
    const bool fp16_big = [] { const char *e = getenv("NF_H16"); return !(e && strcmp(e, "4x4") == 0); }();
    // width 4 in fp16-CNN mode has its own kernel for full 32x32 / 64x64 patches — and, on v_mfma_f32_16x16x32_f16, for images
    // tiled into full 64x64 tiles; everything else rides the width-32 fp16 kernel zero-padded
    const bool w4_full = ((th == 32 && tw == 32) || (th == 64 && tw == 64)) && (!out.tiled || (fp16_big && th == 64 && tw == 64));
    if ((cfg->flags & NF_CFG_FP16_CNN) &&
        (out.prog.width == 8 || out.prog.width == 16 || out.prog.width == 32 || (out.prog.width == 4 && !w4_full))) {
        out.prog5.width = 32;
        for (int i = 0; i < out.prog.n_ops; ++i) {
            const NfOp &src = out.prog.ops[i];
            NfOp &dst = out.prog5.ops[out.prog5.n_ops++];
            dst.type = src.type;
            dst.off = (int32_t)out.block5.size();
            const float *v1 = out.block.data() + src.off;
            if (src.type == NF_OP_MIX) {
                out.block5.insert(out.block5.end(), v1, v1 + 16);
            } else if (src.type == NF_OP_COUPLING_FWD || src.type == NF_OP_COUPLING_REV) {
                out.block5.resize(out.block5.size() + NF5_CPL_SIZE);
                relayout_coupling_wide32_fp16(v1, out.prog.width, out.block5.data() + dst.off);
            } else if (src.type == NF_OP_SCALE) {
                out.block5.insert(out.block5.end(), v1, v1 + 4);
            } else {
                dst.off = src.off;   // conditioning slot
            }
        }
        if (out.block5.empty()) out.block5.assign(4, 0.0f);
    }
    out.block3.clear();
    memset(&out.prog3, 0, sizeof(out.prog3));
    if (out.prog.width == 4 && (cfg->flags & NF_CFG_FP16_CNN)) {
        out.prog3.width = 4;
        out.fp16_big = fp16_big;
        for (int i = 0; i < out.prog.n_ops; ++i) {
            const NfOp &src = out.prog.ops[i];
            NfOp &dst = out.prog3.ops[out.prog3.n_ops++];
            dst.type = src.type;
            dst.off = (int32_t)out.block3.size();
            const float *v1 = out.block.data() + src.off;
            if (src.type == NF_OP_MIX) {
                for (int j = 0; j < 4; ++j)
                    for (int c = 0; c < 4; ++c) out.block3.push_back(v1[c * 4 + j]);
            } else if (src.type == NF_OP_COUPLING_FWD || src.type == NF_OP_COUPLING_REV) {
                // 64x64 patches / tiles: l_2's operands for v_mfma_f32_16x16x32_f16 ride along (NF11_CPL_A2)
                const bool with_a2 = th == 64 && tw == 64;
                out.block3.resize(out.block3.size() + (out.fp16_big ? (with_a2 ? NF11_CPL_SIZE_L2 : NF11_CPL_SIZE) : NF3_CPL_SIZE));
                if (out.fp16_big) relayout_coupling_v11(v1, out.block3.data() + dst.off, with_a2);
                else relayout_coupling_v3(v1, out.block3.data() + dst.off);
            } else if (src.type == NF_OP_SCALE) {
                out.block3.insert(out.block3.end(), v1, v1 + 4);
            } else {
                dst.off = src.off;   // conditioning slot
            }
        }
        if (out.block3.empty()) out.block3.assign(4, 0.0f);
        if (out.block3.size() > (size_t)(out.fp16_big ? NF11_MAX_FLOATS : NF2_MAX_FLOATS)) return fail(NF_EINVAL, "model too large for the fp16-CNN LDS image");
    }
    if (out.tiled) {
        // every coupling widens the dependence of a pixel by 2 (3x3, 1x1, 3x3): nf_device.h, "overlapping tiles"
        int rc = plan_tile_segments(out.prog, cfg->height, cfg->width, th, tw, out.segs);
        if (rc != NF_OK) return rc;
    }
    return NF_OK;
}

// NF_KERNEL=valu forces the scalar-weight VALU kernel (A/B testing); default = matrix core
bool use_matrix_core()
{
    static const bool mc = [] {
        const char *e = getenv("NF_KERNEL");
        return !(e && strcmp(e, "valu") == 0);
    }();   // read once: this sits on the per-call path
    return mc;
}

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    int enter(int dev)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return fail_hip(e, "hipGetDevice");
        if (prev != dev) {
            e = hipSetDevice(dev);
            if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
            changed = true;
        }
        return NF_OK;
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

}  // namespace

int nf_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return fail(code, "%s", buf);
}

int nf_fail_hip(hipError_t e, const char *what) { return fail_hip(e, what); }

struct nf_bs_state;
static void nf_bs_destroy(nf_bs_state *s);

struct nf_handle {
    nf_config cfg;
    int device = 0;
    int n_cu = 256;
    Built fwd, rev;
    float *d_fwd = nullptr;
    float *d_rev = nullptr;
    float *d_fwd2 = nullptr;   // matrix-core layout (null when unavailable)
    float *d_rev2 = nullptr;
    float *d_fwd3 = nullptr;   // fp16-CNN layout (NF_CFG_FP16_CNN)
    float *d_rev3 = nullptr;
    float *d_fwd4 = nullptr;   // wide-CNN layout (width 32)
    float *d_rev4 = nullptr;
    float *d_fwd5 = nullptr;   // wide-CNN fp16 layout
    float *d_rev5 = nullptr;
    float *d_fwd6 = nullptr;   // width-16 layout
    float *d_rev6 = nullptr;
    float *d_fwd7 = nullptr;   // GEMM layout (widths 33 .. 512)
    float *d_rev7 = nullptr;
    float *d_fwd8 = nullptr;   // fp16-CNN GEMM layout
    float *d_rev8 = nullptr;
    bool scalar_ok = true;     // the scalar-weight kernel's LDS tiles fit this patch shape / width
    // scratch of the tiled calls (images beyond 64x64): the per-tile sums and the tensors between two segments.  A small set of
    // device buffers owned by the handle, one per call in flight; a call takes a free one (waiting, on ITS stream, for the work
    // that last used it), so nf_nll / nf_sample allocate only when a call needs more than any earlier one did, or when more
    // calls are in flight than ever before — nf_reserve_workspace sizes it up front
    struct Workspace {
        char *p = nullptr;
        size_t bytes = 0;
        hipEvent_t ev = nullptr;   // recorded behind the last launch that used the buffer
        bool busy = false;         // a host thread is enqueueing on it
        bool used = false;         // `ev` was recorded behind real work (a freshly reserved buffer has no prior user to wait for)
    };
    std::mutex ws_mu;
    std::vector<Workspace *> ws;
    // batch-statistics mode (nf_*_batchstats): the raw model and a lazily allocated scratch
    std::vector<nf_layer_desc> layers;
    std::vector<float> raw;
    std::mutex bs_mu;
    struct nf_bs_state *bs = nullptr;
    // cross-rank batch statistics of nf_*_batchstats (nf_set_sync)
    nf_allreduce_fn sync_fn = nullptr;
    void *sync_user = nullptr;
    double *sync_buf = nullptr;
    int sync_world = 1;
};

int nf_handle_geometry(const nf_handle *h, int32_t *H, int32_t *W, int32_t *device)
{
    if (!h) return fail(NF_EINVAL, "handle is NULL");
    *H = h->cfg.height;
    *W = h->cfg.width;
    *device = h->device;
    return NF_OK;
}

extern "C" {

int nf_destroy(nf_handle *h);

int nf_abi_version(void) { return NF_ABI_VERSION; }

const char *nf_last_error(void) { return g_last_error.c_str(); }

int64_t nf_layer_param_count(int32_t type, int32_t width) { return layer_param_count(type, width); }

int nf_fold_params(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params,
                   int32_t direction, int32_t *ops_out, int32_t ops_cap, int32_t *n_ops, float *folded,
                   size_t folded_cap, size_t *n_folded, double *ld_const)
{
    if (direction != 0 && direction != 1) return fail(NF_EINVAL, "direction must be 0 or 1");
    Built b;
    int rc = build_program(cfg, layers, params, n_params, direction, b);
    if (rc != NF_OK) return rc;
    if (n_ops) *n_ops = b.prog.n_ops;
    if (n_folded) *n_folded = b.block.size();
    if (ld_const) *ld_const = b.ld_const;
    if (ops_out) {
        if (ops_cap < b.prog.n_ops) return fail(NF_EINVAL, "ops_cap too small");
        for (int i = 0; i < b.prog.n_ops; ++i) {
            ops_out[2 * i] = b.prog.ops[i].type;
            ops_out[2 * i + 1] = b.prog.ops[i].off;
        }
    }
    if (folded) {
        if (folded_cap < b.block.size()) return fail(NF_EINVAL, "folded_cap too small");
        memcpy(folded, b.block.data(), b.block.size() * sizeof(float));
    }
    return NF_OK;
}

int nf_fold_layout(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params, int32_t direction,
                   int32_t path, int32_t *ops_out, int32_t ops_cap, int32_t *n_ops, int32_t *layout_width, float *folded,
                   size_t folded_cap, size_t *n_folded)
{
    if (direction != 0 && direction != 1) return fail(NF_EINVAL, "direction must be 0 or 1");
    Built b;
    int rc = build_program(cfg, layers, params, n_params, direction, b);
    if (rc != NF_OK) return rc;
    const NfProgram *pg = nullptr;
    const std::vector<float> *blk = nullptr;
    switch (path) {
    case NF_PATH_SCALAR: pg = &b.prog; blk = &b.block; break;
    case NF_PATH_MFMA4: pg = &b.prog2; blk = &b.block2; break;
    case NF_PATH_FP16: pg = &b.prog3; blk = &b.block3; break;
    case NF_PATH_WIDE32: pg = &b.prog4; blk = &b.block4; break;
    case NF_PATH_WIDE32_FP16: pg = &b.prog5; blk = &b.block5; break;
    case NF_PATH_WIDE16: pg = &b.prog6; blk = &b.block6; break;
    case NF_PATH_GEMM: pg = &b.prog7; blk = &b.block7; break;
    case NF_PATH_GEMM_FP16: pg = &b.prog8; blk = &b.block8; break;
    default: return fail(NF_EINVAL, "unknown kernel path %d", path);
    }
    if (blk->empty()) return fail(NF_EINVAL, "this model has no parameter block for kernel path %d", path);
    if (n_ops) *n_ops = pg->n_ops;
    if (layout_width) *layout_width = pg->width;
    if (n_folded) *n_folded = blk->size();
    if (ops_out) {
        if (ops_cap < pg->n_ops) return fail(NF_EINVAL, "ops_cap too small");
        for (int i = 0; i < pg->n_ops; ++i) {
            ops_out[2 * i] = pg->ops[i].type;
            ops_out[2 * i + 1] = pg->ops[i].off;
        }
    }
    if (folded) {
        if (folded_cap < blk->size()) return fail(NF_EINVAL, "folded_cap too small");
        memcpy(folded, blk->data(), blk->size() * sizeof(float));
    }
    return NF_OK;
}

int nf_sdn5_scalars(const float *sdn_params, const nf_cond *cond, double out[2])
{
    if (!sdn_params || !out) return fail(NF_EINVAL, "null argument");
    return sdn5_scalars(sdn_params, cond, out);
}

int nf_create(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params, nf_handle **out)
{
    if (!out) return fail(NF_EINVAL, "out is NULL");
    *out = nullptr;
    nf_handle *h = new (std::nothrow) nf_handle();
    if (!h) return fail(NF_ENOMEM, "out of host memory");
    int rc = build_program(cfg, layers, params, n_params, 0, h->fwd);
    if (rc == NF_OK) rc = build_program(cfg, layers, params, n_params, 1, h->rev);
    if (rc != NF_OK) {
        delete h;
        return rc;
    }
    h->cfg = *cfg;
    h->layers.assign(layers, layers + cfg->n_layers);
    h->raw.assign(params, params + n_params);
    {   // the two LDS tiles of the scalar-weight kernel must fit one CU (160 KiB)
        const size_t tile_px = ((size_t)(h->fwd.tile_h + 2) * (h->fwd.tile_w + 2) + 1) & ~(size_t)1;
        const size_t lds = sizeof(float) * (tile_px * (2 + (size_t)h->fwd.prog.width) + 64);
        h->scalar_ok = lds <= 160 * 1024 && h->fwd.prog.width <= 32;
        if (lds > 160 * 1024 && h->fwd.block2.empty() && h->fwd.block4.empty() && h->fwd.block5.empty() && h->fwd.block6.empty() &&
            h->fwd.block7.empty() && h->fwd.block8.empty()) {
            const int w = h->fwd.prog.width;
            delete h;
            return fail(NF_EINVAL, "a %dx%d patch with coupling width %d needs %zu KiB of LDS (> 160): unsupported",
                        cfg->height, cfg->width, w, lds / 1024);
        }
    }
    hipError_t e;
    if (cfg->device >= 0) {
        h->device = cfg->device;
    } else if ((e = hipGetDevice(&h->device)) != hipSuccess) {
        delete h;
        return fail_hip(e, "hipGetDevice");
    }
    DeviceGuard guard;
    if ((rc = guard.enter(h->device)) != NF_OK) {
        delete h;
        return rc;
    }
    int n_cu = 0;
    if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device)) != hipSuccess) {
        delete h;
        return fail_hip(e, "hipDeviceGetAttribute");
    }
    h->n_cu = n_cu > 0 ? n_cu : 256;
    const size_t nb_f = h->fwd.block.size() * sizeof(float), nb_r = h->rev.block.size() * sizeof(float);
    if ((e = hipMalloc((void **)&h->d_fwd, nb_f)) != hipSuccess || (e = hipMalloc((void **)&h->d_rev, nb_r)) != hipSuccess) {
        if (h->d_fwd) (void)hipFree(h->d_fwd);
        delete h;
        return fail_hip(e, "hipMalloc(params)");
    }
    if ((e = hipMemcpy(h->d_fwd, h->fwd.block.data(), nb_f, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(h->d_rev, h->rev.block.data(), nb_r, hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipFree(h->d_fwd);
        (void)hipFree(h->d_rev);
        delete h;
        return fail_hip(e, "hipMemcpy(params)");
    }
    if (cfg->flags & NF_CFG_FP16_CNN) {
        const int hw = cfg->height * cfg->width;
        const bool w4_ok = !h->fwd.block3.empty() && ((cfg->height == 32 && cfg->width == 32) || (cfg->height == 64 && cfg->width == 64) ||
                                                      (h->fwd.tiled && h->fwd.fp16_big && h->fwd.tile_h == 64 && h->fwd.tile_w == 64));
        if ((!w4_ok && h->fwd.block5.empty() && h->fwd.block8.empty()) || hw == 0) {
            nf_destroy(h);   // the scalar-layout blocks are already on the device
            return fail(NF_EINVAL, "NF_CFG_FP16_CNN: no half-precision kernel for this width / patch shape");
        }
    }
    for (int d = 0; d < 14; ++d) {
        const std::vector<float> &b2 = d == 0 ? h->fwd.block2 : d == 1 ? h->rev.block2 : d == 2 ? h->fwd.block3 : d == 3 ? h->rev.block3
                                       : d == 4 ? h->fwd.block4 : d == 5 ? h->rev.block4 : d == 6 ? h->fwd.block5 : d == 7 ? h->rev.block5
                                       : d == 8 ? h->fwd.block6 : d == 9 ? h->rev.block6 : d == 10 ? h->fwd.block7 : d == 11 ? h->rev.block7
                                       : d == 12 ? h->fwd.block8 : h->rev.block8;
        float **dst = d == 0 ? &h->d_fwd2 : d == 1 ? &h->d_rev2 : d == 2 ? &h->d_fwd3 : d == 3 ? &h->d_rev3 : d == 4 ? &h->d_fwd4
                      : d == 5 ? &h->d_rev4 : d == 6 ? &h->d_fwd5 : d == 7 ? &h->d_rev5 : d == 8 ? &h->d_fwd6 : d == 9 ? &h->d_rev6
                      : d == 10 ? &h->d_fwd7 : d == 11 ? &h->d_rev7 : d == 12 ? &h->d_fwd8 : &h->d_rev8;
        if (b2.empty()) continue;
        if ((e = hipMalloc((void **)dst, b2.size() * sizeof(float))) != hipSuccess ||
            (e = hipMemcpy(*dst, b2.data(), b2.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) {
            nf_destroy(h);
            return fail_hip(e, "hipMalloc/hipMemcpy(matrix-core params)");
        }
    }
    *out = h;
    return NF_OK;
}

int nf_destroy(nf_handle *h)
{
    if (!h) return NF_OK;
    nf_hostpipe_release(h);
    DeviceGuard guard;
    (void)guard.enter(h->device);
    if (h->d_fwd) (void)hipFree(h->d_fwd);
    if (h->d_rev) (void)hipFree(h->d_rev);
    if (h->d_fwd2) (void)hipFree(h->d_fwd2);
    if (h->d_rev2) (void)hipFree(h->d_rev2);
    if (h->d_fwd3) (void)hipFree(h->d_fwd3);
    if (h->d_rev3) (void)hipFree(h->d_rev3);
    if (h->d_fwd4) (void)hipFree(h->d_fwd4);
    if (h->d_rev4) (void)hipFree(h->d_rev4);
    if (h->d_fwd5) (void)hipFree(h->d_fwd5);
    if (h->d_rev5) (void)hipFree(h->d_rev5);
    if (h->d_fwd6) (void)hipFree(h->d_fwd6);
    if (h->d_rev6) (void)hipFree(h->d_rev6);
    if (h->d_fwd7) (void)hipFree(h->d_fwd7);
    if (h->d_rev7) (void)hipFree(h->d_rev7);
    if (h->d_fwd8) (void)hipFree(h->d_fwd8);
    if (h->d_rev8) (void)hipFree(h->d_rev8);
    nf_bs_destroy(h->bs);
    for (nf_handle::Workspace *w : h->ws) {
        if (w->ev) {
            (void)hipEventSynchronize(w->ev);
            (void)hipEventDestroy(w->ev);
        }
        if (w->p) (void)hipFree(w->p);
        delete w;
    }
    delete h;
    return NF_OK;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- argument checking + NfLaunch assembly shared by the resident and the batch-statistics paths ----
static int nll_args(nf_handle *h, const float *x, const float *y, int64_t B, const nf_cond *cond, float *nll_out,
                    float *sd_out, float *logdet_out, float *z_out, double *sums_out, uint32_t flags, NfLaunch &a)
{
    if (!h) return fail(NF_EINVAL, "handle is NULL");
    if (B < 0) return fail(NF_EINVAL, "B must be >= 0");
    if (B > 0 && !x) return fail(NF_EINVAL, "x is NULL");
    if (h->fwd.has_sdn && B > 0 && !y) return fail(NF_EINVAL, "model has a signal-dependent layer but y is NULL");
    if (!aligned16(x) || !aligned16(y) || !aligned16(z_out))
        return fail(NF_EINVAL, "x, y and z_out must be 16-byte aligned (every pixel is one float4 access)");
    float ca[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {1.f, 1.f, 1.f, 1.f};
    double ld_call = 0.0;   // per-call part of the constant log-det (plain `gain` layers)
    for (size_t i = 0; i < h->fwd.cond.size(); ++i) {
        double sc[2];
        int rc = cond_scalars(h->fwd.cond[i], cond, sc);
        if (rc != NF_OK) return rc;
        if (is_gain_kind(h->fwd.cond[i].kind)) {
            ca[i] = (float)(1.0 / sc[0]);            // NLL direction divides
            // AffineCouplingGain / GainEx1 / GainEx3 write -log(scale) once per patch (no H*W*C factor, e.g.
            // AffineCouplingGain.py:113-127); GainEx2 broadcasts the scale first and sums (AffineCouplingGainEx2.py:112-126)
            ld_call -= (h->fwd.cond[i].kind == NF_LAYER_GAIN2 ? (double)h->cfg.height * h->cfg.width * kC : 1.0) * log(sc[0]);
        } else {
            ca[i] = (float)sc[0];
            cb[i] = (float)sc[1];
        }
    }
    memset(&a, 0, sizeof(a));
    a.in = x;
    a.y = y;
    a.out = z_out;
    a.nll_out = nll_out;
    a.sd_out = sd_out;
    a.ld_out = logdet_out;
    a.sums = sums_out;
    a.B = B;
    a.ld_const = ld_call;   // + the model's constant part, added by the launcher
    a.in_scale = 1.0f;
    memcpy(a.cond_a, ca, sizeof(ca));
    memcpy(a.cond_b, cb, sizeof(cb));
    a.H = h->cfg.height;
    a.W = h->cfg.width;
    a.flags = (flags & NF_NO_PRIOR) ? 0u : NF_K_PRIOR;
    if (flags & NF_SUMS_WIDE) a.flags |= NF_K_SUMS_WIDE;
    return NF_OK;
}

static int sample_args(nf_handle *h, const float *y, const float *eps, uint64_t seed, int64_t patch_index_base,
                       float temp, int64_t B, const nf_cond *cond, float *x_out, NfLaunch &a)
{
    if (!h) return fail(NF_EINVAL, "handle is NULL");
    if (B < 0) return fail(NF_EINVAL, "B must be >= 0");
    if (B > 0 && !x_out) return fail(NF_EINVAL, "x_out is NULL");
    if (h->rev.has_sdn && B > 0 && !y) return fail(NF_EINVAL, "model has a signal-dependent layer but y is NULL");
    if (!aligned16(y) || !aligned16(eps) || !aligned16(x_out))
        return fail(NF_EINVAL, "y, eps and x_out must be 16-byte aligned (every pixel is one float4 access)");
    float ca[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {1.f, 1.f, 1.f, 1.f};
    for (size_t i = 0; i < h->rev.cond.size(); ++i) {
        double sc[2];
        int rc = cond_scalars(h->rev.cond[i], cond, sc);
        if (rc != NF_OK) return rc;
        ca[i] = (float)sc[0];                        // sampling direction multiplies
        cb[i] = (float)sc[1];
    }
    memset(&a, 0, sizeof(a));
    a.in = eps;
    a.y = y;
    a.out = x_out;
    a.B = B;
    a.patch_base = patch_index_base;
    a.seed = seed;
    a.in_scale = temp;
    memcpy(a.cond_a, ca, sizeof(ca));
    memcpy(a.cond_b, cb, sizeof(cb));
    a.H = h->cfg.height;
    a.W = h->cfg.width;
    a.flags = eps ? 0u : NF_K_PHILOX_IN;
    return NF_OK;
}

// ---- workspace of the tiled calls -------------------------------------------------------------------------------------
static size_t tiled_workspace_bytes(const nf_handle *h, int direction, int64_t B, bool want_sums)
{
    const Built &b = direction == 0 ? h->fwd : h->rev;
    if (!b.tiled || B <= 0) return 0;
    const int S = (int)b.segs.size();
    size_t part4 = 0;
    for (int s = 0; s < S; ++s) part4 += (size_t)B * b.segs[s].ny * b.segs[s].nx;
    const size_t img = (((size_t)B * h->cfg.height * h->cfg.width * kC * sizeof(float)) + 255) & ~(size_t)255;
    const size_t part = want_sums ? ((part4 * 4 * sizeof(float) + 255) & ~(size_t)255) : 0;
    return part + img * (size_t)(S - 1 < 2 ? (S - 1 < 0 ? 0 : S - 1) : 2);
}

// a buffer of >= bytes that no other call is enqueueing on; work the stream `st` enqueues is ordered behind its last user
static int ws_acquire(nf_handle *h, size_t bytes, hipStream_t st, nf_handle::Workspace **out)
{
    *out = nullptr;
    if (bytes == 0) return NF_OK;
    nf_handle::Workspace *w = nullptr;
    hipError_t e = hipSuccess;
    {
        std::lock_guard<std::mutex> lk(h->ws_mu);
        for (nf_handle::Workspace *c : h->ws)      // smallest free buffer that is large enough, else the largest free one (to grow)
            if (!c->busy && (!w || (c->bytes >= bytes ? (w->bytes < bytes || c->bytes < w->bytes) : (w->bytes < bytes && c->bytes > w->bytes)))) w = c;
        if (!w) {
            w = new (std::nothrow) nf_handle::Workspace();
            if (!w) return fail(NF_ENOMEM, "out of host memory");
            if ((e = hipEventCreateWithFlags(&w->ev, hipEventDisableTiming)) != hipSuccess) {
                delete w;
                return fail_hip(e, "hipEventCreate");
            }
            h->ws.push_back(w);
        }
        w->busy = true;                            // ours from here on: the lock is NOT held across the (device-synchronising) grow below
    }
    if (w->bytes < bytes) {                        // grow: the only allocation a call can make (the first one of its size)
        if (w->p) {
            (void)hipEventSynchronize(w->ev);
            (void)hipFree(w->p);
            w->p = nullptr;
            w->bytes = 0;
        }
        if ((e = hipMalloc((void **)&w->p, bytes)) != hipSuccess) {
            std::lock_guard<std::mutex> lk(h->ws_mu);
            w->busy = false;
            return fail_hip(e, "hipMalloc(tiled workspace)");
        }
        w->bytes = bytes;
    } else if (w->used && (e = hipStreamWaitEvent(st, w->ev, 0)) != hipSuccess) {
        std::lock_guard<std::mutex> lk(h->ws_mu);
        w->busy = false;
        return fail_hip(e, "hipStreamWaitEvent");
    }
    *out = w;
    return NF_OK;
}

static void ws_release(nf_handle *h, nf_handle::Workspace *w, hipStream_t st, bool worked = true)
{
    if (!w) return;
    if (worked) (void)hipEventRecord(w->ev, st);   // (nf_reserve_workspace only sized the buffer: nothing for its first user to wait for)
    std::lock_guard<std::mutex> lk(h->ws_mu);
    if (worked) w->used = true;
    w->busy = false;
}

// Images beyond 64x64 (nf_device.h, "overlapping tiles"): per segment of the program one launch of the fused width-4 kernel
// (or, at widths 8 / 16 / 32 and in fp16-CNN mode, of the width-32 matrix-core kernel) over B x tiles tile-sized "patches" that reads and writes image-shaped tensors in place (the caller's, and between two
// segments a scratch tensor), then — in the NLL direction — a one-wavefront-per-image kernel that adds the tiles' sums up.  Scratch
// comes from the handle's workspace set (above): one buffer per call in flight, so concurrent calls on one handle never share it.
static int launch_tiled(nf_handle *h, int direction, NfLaunch &a, hipStream_t st, const char *what)
{
    const Built &b = direction == 0 ? h->fwd : h->rev;
    float *d1 = direction == 0 ? h->d_fwd : h->d_rev;
    float *d2 = direction == 0 ? h->d_fwd2 : h->d_rev2;
    const int64_t B = a.B;
    const int S = (int)b.segs.size();
    const bool want = direction == 0 && (a.nll_out || a.sd_out || a.ld_out || a.sums);
    NfTileParts tp;
    memset(&tp, 0, sizeof(tp));
    tp.n_seg = S;
    int64_t part4 = 0;   // float4 entries
    for (int s = 0; s < S; ++s) {
        tp.nt[s] = b.segs[s].ny * b.segs[s].nx;
        tp.off[s] = part4;
        if (B > (INT64_MAX / 64) / tp.nt[s]) return fail(NF_EINVAL, "B too large");
        part4 += B * tp.nt[s];
    }
    const size_t img_bytes = (((size_t)B * h->cfg.height * h->cfg.width * kC * sizeof(float)) + 255) & ~(size_t)255;
    float *part = nullptr, *scratch[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    nf_handle::Workspace *wsp = nullptr;
    {
        const int rc = ws_acquire(h, tiled_workspace_bytes(h, direction, B, want), st, &wsp);
        if (rc != NF_OK) return rc;
        char *q = wsp ? wsp->p : nullptr;
        if (want) {
            part = reinterpret_cast<float *>(q);
            q += ((size_t)part4 * 4 * sizeof(float) + 255) & ~(size_t)255;
        }
        for (int i = 0; i < 2 && i < S - 1; ++i, q += img_bytes) scratch[i] = reinterpret_cast<float *>(q);
    }
    // kernel family: fp16-CNN mode and widths 8 / 16 / 32 on the width-32 matrix-core kernel (zero-padded), width 4 in fp32 on
    // the fused width-4 kernels
    float *d4 = direction == 0 ? h->d_fwd4 : h->d_rev4;
    float *d5 = direction == 0 ? h->d_fwd5 : h->d_rev5;
    float *d3 = direction == 0 ? h->d_fwd3 : h->d_rev3;
    float *d7 = direction == 0 ? h->d_fwd7 : h->d_rev7;   // widths 33 .. 512: the GEMM kernels take tiles as they take patches
    float *d8 = direction == 0 ? h->d_fwd8 : h->d_rev8;
    const int gemm = d8 ? 2 : d7 ? 1 : 0;
    float *d6 = direction == 0 ? h->d_fwd6 : h->d_rev6;   // width 16 in fp32: nf_wide16.hip takes tiles as well
    const bool w16 = !gemm && !d5 && d6 && b.prog.width == 16 && (use_matrix_core() || !h->scalar_ok);
    const bool hb = !gemm && d3 && b.fp16_big && !d5;   // width 4, fp16 CNN, full 64x64 tiles: the fused kernel on v_mfma_f32_16x16x32_f16
    const int wide = (hb || gemm || w16) ? 0 : d5 ? 2 : (d4 && b.prog.width > 4) ? 1 : 0;
    const bool mc = hb || (d2 && (use_matrix_core() || !h->scalar_ok));
    const NfProgram &full = w16 ? b.prog6 : gemm == 2 ? b.prog8 : gemm == 1 ? b.prog7 : hb ? b.prog3 : wide == 2 ? b.prog5 : wide == 1 ? b.prog4 : mc ? b.prog2 : b.prog;
    const float *cur = a.in;
    for (int s = 0; s < S && e == hipSuccess; ++s) {
        const Built::TileSeg &g = b.segs[s];
        NfProgram sp;
        memset(&sp, 0, sizeof(sp));
        sp.width = full.width;
        sp.n_ops = g.op1 - g.op0;
        memcpy(sp.ops, full.ops + g.op0, sizeof(NfOp) * (size_t)sp.n_ops);
        NfLaunch t = a;
        t.B = B * tp.nt[s];
        t.H = b.tile_h;
        t.W = b.tile_w;
        t.img_H = h->cfg.height;
        t.img_W = h->cfg.width;
        t.tile_ny = g.ny;
        t.tile_nx = g.nx;
        t.tile_halo = g.halo;
        t.flags |= NF_K_TILED;
        if (s > 0) {   // only the first segment draws / scales the input
            t.flags &= ~(uint32_t)NF_K_PHILOX_IN;
            t.in_scale = 1.0f;
        }
        t.in = cur;
        t.out = s == S - 1 ? a.out : scratch[s & 1];
        t.nll_out = t.sd_out = t.ld_out = nullptr;
        t.sums = nullptr;
        t.tile_part = want ? part + tp.off[s] * 4 : nullptr;
        if (w16) {
            t.params = d6;
            t.n_params = (int32_t)b.block6.size();
            e = nf_launch_wide16(sp, t, h->n_cu, h->device, st);
        } else if (gemm == 2) {
            t.params = d8;
            t.n_params = (int32_t)b.block8.size();
            t.flags |= NF_K_FP16_CNN;
            e = b.gemm16_b ? nf_launch_gemm16b(sp, t, h->n_cu, h->device, st) : nf_launch_gemm16(sp, t, h->n_cu, h->device, st);
        } else if (gemm == 1) {
            t.params = d7;
            t.n_params = (int32_t)b.block7.size();
            e = b.gemm_b ? nf_launch_gemmb(sp, t, h->n_cu, h->device, st) : nf_launch_gemm(sp, t, h->n_cu, h->device, st);
        } else if (wide) {
            t.params = wide == 2 ? d5 : d4;
            t.n_params = (int32_t)(wide == 2 ? b.block5.size() : b.block4.size());
            if (wide == 2) t.flags |= NF_K_FP16_CNN;
            e = nf_launch_wide(sp, t, h->n_cu, h->device, st);
        } else {
            t.params = hb ? d3 : mc ? d2 : d1;
            if (hb) {
                t.n_params = (int32_t)b.block3.size();
                t.flags |= NF_K_FP16_CNN | NF_K_FP16_BIG;
            } else if (mc) {
                t.n_params = (int32_t)b.block2.size();
            }
            e = nf_launch_flow(sp, t, h->n_cu, st, mc);
        }
        cur = t.out;
    }
    if (e == hipSuccess && want)
        e = nf_launch_tile_combine(part, tp, B, (double)h->cfg.height * h->cfg.width * kC, a.ld_const, a.flags, a.nll_out, a.sd_out, a.ld_out,
                                   a.sums, st);
    ws_release(h, wsp, st);
    if (e != hipSuccess) return fail_hip(e, what);
    return NF_OK;
}

// launch on the handle's resident (running-statistics) programs
static int launch_resident(nf_handle *h, int direction, NfLaunch &a, hipStream_t st, const char *what)
{
    const Built &b = direction == 0 ? h->fwd : h->rev;
    float *d1 = direction == 0 ? h->d_fwd : h->d_rev;
    float *d2 = direction == 0 ? h->d_fwd2 : h->d_rev2;
    float *d3 = direction == 0 ? h->d_fwd3 : h->d_rev3;
    float *d4 = direction == 0 ? h->d_fwd4 : h->d_rev4;
    float *d5 = direction == 0 ? h->d_fwd5 : h->d_rev5;
    if (direction == 0) a.ld_const += b.ld_const;
    if (b.tiled) return launch_tiled(h, direction, a, st, what);
    if (d5) {   // NF_CFG_FP16_CNN at width 8 / 16 / 32: v_mfma_f32_32x32x16_f16
        a.params = d5;
        a.n_params = (int32_t)b.block5.size();
        a.flags |= NF_K_FP16_CNN;
        hipError_t e = nf_launch_wide(b.prog5, a, h->n_cu, h->device, st);
        if (e != hipSuccess) return fail_hip(e, what);
        return NF_OK;
    }
    float *d8 = direction == 0 ? h->d_fwd8 : h->d_rev8;
    if (d8) {   // NF_CFG_FP16_CNN at widths 33 .. 512: the GEMM kernel on v_mfma_f32_32x32x16_f16 (nf_gemm16.hip)
        a.params = d8;
        a.n_params = (int32_t)b.block8.size();
        a.flags |= NF_K_FP16_CNN;
        hipError_t e = b.gemm16_b ? nf_launch_gemm16b(b.prog8, a, h->n_cu, h->device, st) : nf_launch_gemm16(b.prog8, a, h->n_cu, h->device, st);
        if (e != hipSuccess) return fail_hip(e, what);
        return NF_OK;
    }
    float *d7 = direction == 0 ? h->d_fwd7 : h->d_rev7;
    if (d7) {   // widths 33 .. 512: LDS-staged GEMMs on v_mfma_f32_32x32x2_f32 (nf_gemm.hip); no other kernel holds these widths
        a.params = d7;
        a.n_params = (int32_t)b.block7.size();
        hipError_t e = b.gemm_b ? nf_launch_gemmb(b.prog7, a, h->n_cu, h->device, st) : nf_launch_gemm(b.prog7, a, h->n_cu, h->device, st);
        if (e != hipSuccess) return fail_hip(e, what);
        return NF_OK;
    }
    // NF_KERNEL=valu (A/B aid) selects the scalar-weight kernel only where that kernel can hold the patch
    const bool mcore = use_matrix_core() || !h->scalar_ok;
    float *d6 = direction == 0 ? h->d_fwd6 : h->d_rev6;
    if (d6 && mcore) {   // width 16: v_mfma_f32_16x16x4_f32 (nf_wide16.hip)
        a.params = d6;
        a.n_params = (int32_t)b.block6.size();
        hipError_t e = nf_launch_wide16(b.prog6, a, h->n_cu, h->device, st);
        if (e != hipSuccess) return fail_hip(e, what);
        return NF_OK;
    }
    if (d4 && mcore) {   // width 32: the three convs on v_mfma_f32_32x32x2_f32 (nf_wide.hip)
        a.params = d4;
        a.n_params = (int32_t)b.block4.size();
        hipError_t e = nf_launch_wide(b.prog4, a, h->n_cu, h->device, st);
        if (e != hipSuccess) return fail_hip(e, what);
        return NF_OK;
    }
    const bool mc = d3 || (d2 && use_matrix_core());
    a.params = d1;
    if (d3) {
        a.params = d3;
        a.n_params = (int32_t)b.block3.size();
        a.flags |= NF_K_FP16_CNN | (b.fp16_big ? NF_K_FP16_BIG : 0u);
    } else if (mc) {
        a.params = d2;
        a.n_params = (int32_t)b.block2.size();
    }
    hipError_t e = nf_launch_flow(d3 ? b.prog3 : mc ? b.prog2 : b.prog, a, h->n_cu, st, mc);
    if (e != hipSuccess) return fail_hip(e, what);
    return NF_OK;
}

int64_t nf_workspace_bytes(const nf_handle *h, int32_t direction, int64_t B)
{
    if (!h || (direction != 0 && direction != 1) || B < 0) return fail(NF_EINVAL, "bad argument");
    return (int64_t)tiled_workspace_bytes(h, direction, B, direction == 0);
}

int nf_reserve_workspace(nf_handle *h, int64_t B, int32_t calls_in_flight)
{
    if (!h || B < 0 || calls_in_flight < 1 || calls_in_flight > 64) return fail(NF_EINVAL, "bad argument");
    const size_t need = std::max(tiled_workspace_bytes(h, 0, B, true), tiled_workspace_bytes(h, 1, B, false));
    if (need == 0) return NF_OK;
    DeviceGuard guard;
    int rc = guard.enter(h->device);
    if (rc != NF_OK) return rc;
    std::vector<nf_handle::Workspace *> got;
    for (int i = 0; i < calls_in_flight && rc == NF_OK; ++i) {
        nf_handle::Workspace *w = nullptr;
        rc = ws_acquire(h, need, nullptr, &w);   // marks it busy, so that the next iteration takes (or makes) another one
        if (w) got.push_back(w);
    }
    for (nf_handle::Workspace *w : got) ws_release(h, w, nullptr, false);
    return rc;
}

int nf_kernel_path(const nf_handle *h, int32_t direction)
{
    if (!h || (direction != 0 && direction != 1)) return fail(NF_EINVAL, "bad argument");
    if (direction == 0 ? h->d_fwd5 : h->d_rev5) return NF_PATH_WIDE32_FP16;
    if (h->fwd.tiled) {   // images beyond 64x64: launch_tiled's choice
        if (direction == 0 ? h->d_fwd8 : h->d_rev8) return NF_PATH_GEMM_FP16;
        if (direction == 0 ? h->d_fwd7 : h->d_rev7) return NF_PATH_GEMM;
        if ((direction == 0 ? h->d_fwd3 : h->d_rev3) && h->fwd.fp16_big) return NF_PATH_FP16;
        if ((direction == 0 ? h->d_fwd6 : h->d_rev6) && h->fwd.prog.width == 16 && (use_matrix_core() || !h->scalar_ok)) return NF_PATH_WIDE16;
        if ((direction == 0 ? h->d_fwd4 : h->d_rev4) && h->fwd.prog.width > 4) return NF_PATH_WIDE32;
        return (direction == 0 ? h->d_fwd2 : h->d_rev2) && (use_matrix_core() || !h->scalar_ok) ? NF_PATH_MFMA4 : NF_PATH_SCALAR;
    }
    if (direction == 0 ? h->d_fwd8 : h->d_rev8) return NF_PATH_GEMM_FP16;
    if (direction == 0 ? h->d_fwd7 : h->d_rev7) return NF_PATH_GEMM;
    const bool mcore = use_matrix_core() || !h->scalar_ok;
    if ((direction == 0 ? h->d_fwd6 : h->d_rev6) && mcore) return NF_PATH_WIDE16;
    if ((direction == 0 ? h->d_fwd4 : h->d_rev4) && mcore) return NF_PATH_WIDE32;
    if (direction == 0 ? h->d_fwd3 : h->d_rev3) return NF_PATH_FP16;
    if ((direction == 0 ? h->d_fwd2 : h->d_rev2) && use_matrix_core()) return NF_PATH_MFMA4;
    return NF_PATH_SCALAR;
}

int nf_nll(nf_handle *h, const float *x, const float *y, int64_t B, const nf_cond *cond, float *nll_out, float *sd_out,
           float *logdet_out, float *z_out, double *sums_out, uint32_t flags, void *stream)
{
    NfLaunch a;
    int rc = nll_args(h, x, y, B, cond, nll_out, sd_out, logdet_out, z_out, sums_out, flags, a);
    if (rc != NF_OK) return rc;
    DeviceGuard guard;
    if ((rc = guard.enter(h->device)) != NF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (sums_out && !(flags & NF_ACCUMULATE)) {
        const size_t nb = (flags & NF_SUMS_WIDE) ? (size_t)NF_SUMS_SLOTS * NF_SUMS_STRIDE : 3;
        hipError_t e = hipMemsetAsync(sums_out, 0, nb * sizeof(double), st);
        if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(sums)");
    }
    if (B == 0) return NF_OK;
    return launch_resident(h, 0, a, st, "nf_nll launch");
}

int nf_sample(nf_handle *h, const float *y, const float *eps, uint64_t seed, int64_t patch_index_base, float temp,
              int64_t B, const nf_cond *cond, float *x_out, void *stream)
{
    NfLaunch a;
    int rc = sample_args(h, y, eps, seed, patch_index_base, temp, B, cond, x_out, a);
    if (rc != NF_OK) return rc;
    if (B == 0) return NF_OK;
    DeviceGuard guard;
    if ((rc = guard.enter(h->device)) != NF_OK) return rc;
    return launch_resident(h, 1, a, (hipStream_t)stream, "nf_sample launch");
}

// ---- batch-statistics mode (is_training=True graphs: layers.py:386-398) ----
// The normalisation of every coupling CNN uses the moments of the CURRENT call's B patches, which couples all
// patches: the moments of coupling c depend on the outputs of the couplings before it under THEIR batch moments.
// The call therefore walks the couplings in execution order with the tensor in front of each one RESIDENT in HBM
// (T_0 = the caller's input, T_c = the output of coupling c-1 in one of two scratch buffers); per coupling c
//   launch A_c : [segment c-1 = the layers after coupling c-2 up to coupling c-1, with its moments; STORE T_c;] segment c, statistics of its l_1
//   launch B_c : the same segment from T_c with BN_1 folded, statistics of its l_2
// each followed by a one-workgroup kernel that turns the sums into moments and re-folds the layer IN PLACE on a
// device-resident working copy of the parameter block (which starts as the identity-normalised model), and finally
// one fused pass over the LAST segment (+ the layers behind the last coupling) from the last resident tensor: every
// earlier segment already ran in its final form in the launch that stored the tensor behind it, and the log-det it
// contributed travels with that tensor (NfLaunch::ld_carry, one float per thread of the patch's workgroup).  2 short launches per coupling instead of
// re-running the whole prefix, no host round trip until the moments are copied out at the end.
struct BsPlan {
    bool ready = false;
    Built ident;                      // the model folded with an identity normalisation in every coupling CNN
    float *d_ident = nullptr;         // ... on the device, scalar layout
    float *d_ident2 = nullptr;        // ... matrix-core layout (null when unavailable)
    std::vector<int> cpl_ops;         // op index of every coupling in ident.prog, execution order
    std::vector<int> cpl_row;         // its row in moments_out (NLL layer order)
    std::vector<float> shift;         // [coupling in NLL layer order][2][width]: the running means the statistics are taken around
    std::vector<NfProgram> progA, progB;
    std::vector<NfProgram> progA2, progB2;   // the same segments over the matrix-core layout (width 4)
    NfProgram progF, progF2;          // the last segment + everything behind it (the final launch when there are >= 2 couplings)
    int32_t *d_pairs = nullptr;       // (dst in matrix-core block, src in scalar block) of the BN-dependent entries
    int n_pairs = 0;
};

struct nf_bs_state {
    BsPlan plan[2];
    float *d_work = nullptr, *d_work2 = nullptr;   // working copies of the parameter block (scalar / matrix-core layout)
    float *d_work2b = nullptr;                      // matrix-core path: the launches ping-pong between two copies
    size_t work_cap = 0, work2_cap = 0, work2b_cap = 0;
    double *d_stats_mc = nullptr;                   // matrix-core path: one [NF_STATS_SLOTS][8] block per statistics pass
    size_t stats_mc_passes = 0;
    double *d_stats = nullptr;
    float *d_mom = nullptr;            // [couplings][4][w]
    float *d_T[2] = {nullptr, nullptr};
    size_t T_cap = 0;                  // floats per scratch tensor
    float *d_carry = nullptr;          // [B][1024] per-thread log-det of the segments behind the resident tensor
    size_t carry_cap = 0;
    nf_trainer *wide = nullptr;        // the layer-by-layer evaluator on the trainer's GEMM path (nf_bs_wide_*), where the passes above do not reach
    ~nf_bs_state()
    {
        if (wide) (void)nf_trainer_destroy(wide);
        for (int d = 0; d < 2; ++d) {
            if (plan[d].d_ident) (void)hipFree(plan[d].d_ident);
            if (plan[d].d_ident2) (void)hipFree(plan[d].d_ident2);
            if (plan[d].d_pairs) (void)hipFree(plan[d].d_pairs);
        }
        if (d_work) (void)hipFree(d_work);
        if (d_work2) (void)hipFree(d_work2);
        if (d_work2b) (void)hipFree(d_work2b);
        if (d_stats_mc) (void)hipFree(d_stats_mc);
        if (d_stats) (void)hipFree(d_stats);
        if (d_mom) (void)hipFree(d_mom);
        if (d_T[0]) (void)hipFree(d_T[0]);
        if (d_T[1]) (void)hipFree(d_T[1]);
        if (d_carry) (void)hipFree(d_carry);
    }
};

static int bs_build_plan(nf_handle *h, int direction, BsPlan &P)
{
    // unit-scale normalisation in every coupling: var 1 - eps -> scale exactly 1.  The MEAN stays the stored running mean:
    // the statistics passes then see activations centred near zero, so that their one-pass sum / sum-of-squares does not
    // cancel when a channel's mean is large against its spread (shipped model: mean^2 / var up to 400); the re-fold
    // B <- (B - mean') * scale is unchanged, and the reported batch mean is mean' + running mean.
    std::vector<float> p = h->raw;
    int n_cpl = 0;
    P.shift.clear();
    for (int i = 0; i < h->cfg.n_layers; ++i) {
        const nf_layer_desc &L = h->layers[i];
        if (L.type != NF_LAYER_COUPLING) continue;
        ++n_cpl;
        const int w = L.width, wk = h->fwd.prog.width;   // width of the variables / of the kernels' layout (zero-padded channels:
                                                          // activation 0 on every pixel, batch moments 0, weights and bias stay 0)
        float *lp = p.data() + L.param_offset;
        float *mv[4] = {lp + 19 * w, lp + 20 * w, lp + 22 * w + w * w, lp + 23 * w + w * w};
        P.shift.insert(P.shift.end(), mv[0], mv[0] + w);
        P.shift.insert(P.shift.end(), (size_t)(wk - w), 0.0f);
        P.shift.insert(P.shift.end(), mv[2], mv[2] + w);
        P.shift.insert(P.shift.end(), (size_t)(wk - w), 0.0f);
        for (int j = 0; j < w; ++j) mv[1][j] = mv[3][j] = (float)(1.0 - kBnEps);
    }
    int rc = build_program(&h->cfg, h->layers.data(), p.data(), p.size(), direction, P.ident);
    if (rc != NF_OK) return rc;
    const NfProgram &prog = P.ident.prog;
    P.cpl_ops.clear();
    for (int i = 0; i < prog.n_ops; ++i)
        if (prog.ops[i].type == NF_OP_COUPLING_FWD || prog.ops[i].type == NF_OP_COUPLING_REV) P.cpl_ops.push_back(i);
    if ((int)P.cpl_ops.size() != n_cpl) return fail(NF_EINVAL, "internal: batch-statistics plan mismatch");
    P.cpl_row.resize(n_cpl);
    for (int c = 0; c < n_cpl; ++c) P.cpl_row[c] = direction == 0 ? c : n_cpl - 1 - c;
    P.progA.assign(n_cpl, NfProgram());
    P.progB.assign(n_cpl, NfProgram());
    for (int c = 0; c < n_cpl; ++c) {
        NfProgram &A = P.progA[c], &B = P.progB[c];
        memset(&A, 0, sizeof(A));
        memset(&B, 0, sizeof(B));
        A.width = B.width = prog.width;
        const int first = c == 0 ? 0 : P.cpl_ops[c - 1] + 1;
        if (c > 0) {   // segment c-1 (the layers after coupling c-2 up to and including coupling c-1), now with its moments
            const int prev_first = c == 1 ? 0 : P.cpl_ops[c - 2] + 1;
            if (P.cpl_ops[c - 1] - prev_first + 2 > NF_MAX_OPS) return fail(NF_EINVAL, "too many layers for the batch-statistics plan");
            for (int i = prev_first; i <= P.cpl_ops[c - 1]; ++i) A.ops[A.n_ops++] = prog.ops[i];
            A.ops[A.n_ops].type = NF_OP_STORE;
            A.ops[A.n_ops++].off = 0;
        }
        if (A.n_ops + (P.cpl_ops[c] - first + 1) > NF_MAX_OPS) return fail(NF_EINVAL, "too many layers for the batch-statistics plan");
        for (int i = first; i <= P.cpl_ops[c]; ++i) {
            A.ops[A.n_ops++] = prog.ops[i];
            B.ops[B.n_ops++] = prog.ops[i];
        }
    }
    memset(&P.progF, 0, sizeof(P.progF));
    P.progF.width = prog.width;
    if (n_cpl > 1) {
        const int first = P.cpl_ops[n_cpl - 2] + 1;
        if (prog.n_ops - first > NF_MAX_OPS) return fail(NF_EINVAL, "too many layers for the batch-statistics plan");
        for (int i = first; i < prog.n_ops; ++i) P.progF.ops[P.progF.n_ops++] = prog.ops[i];
    }
    if (!P.ident.block2.empty()) {   // same op sequences, offsets of the matrix-core layout
        auto to_mc = [&](const NfProgram &src) {
            NfProgram dst = src;
            for (int i = 0; i < src.n_ops; ++i) {
                if (src.ops[i].type == NF_OP_STORE) continue;
                for (int k = 0; k < prog.n_ops; ++k)   // identify the op in the full program by (type, offset)
                    if (prog.ops[k].type == src.ops[i].type && prog.ops[k].off == src.ops[i].off) {
                        dst.ops[i].off = P.ident.prog2.ops[k].off;
                        break;
                    }
            }
            return dst;
        };
        P.progA2.resize(n_cpl);
        P.progB2.resize(n_cpl);
        for (int c = 0; c < n_cpl; ++c) {
            P.progA2[c] = to_mc(P.progA[c]);
            P.progB2[c] = to_mc(P.progB[c]);
        }
        P.progF2 = to_mc(P.progF);
    }
    hipError_t e;
    const size_t nb = P.ident.block.size() * sizeof(float);
    if ((e = hipMalloc((void **)&P.d_ident, nb)) != hipSuccess ||
        (e = hipMemcpy(P.d_ident, P.ident.block.data(), nb, hipMemcpyHostToDevice)) != hipSuccess)
        return fail_hip(e, "batch-statistics plan upload");
    if (!P.ident.block2.empty()) {
        // the normalisation-dependent entries of the matrix-core layout are plain copies of scalar-layout entries
        std::vector<int32_t> pairs;
        const int w = 4;
        for (int i = 0; i < prog.n_ops; ++i) {
            if (prog.ops[i].type != NF_OP_COUPLING_FWD && prog.ops[i].type != NF_OP_COUPLING_REV) continue;
            const int o1 = prog.ops[i].off, o2 = P.ident.prog2.ops[i].off;
            for (int j = 0; j < 4; ++j) {
                pairs.push_back(o2 + NF2_CPL_B1 + j); pairs.push_back(o1 + nf_cpl_off_B1(w) + j);
                pairs.push_back(o2 + NF2_CPL_B2 + j); pairs.push_back(o1 + nf_cpl_off_B2(w) + j);
                for (int di = 0; di < 3; ++di)
                    for (int q = 0; q < 6; ++q) {
                        pairs.push_back(o2 + NF2_CPL_W1T + 24 * j + 8 * di + q);
                        pairs.push_back(o1 + nf_cpl_off_W1(w) + (di * 6 + q) * 4 + j);
                    }
                for (int i2 = 0; i2 < 4; ++i2) {
                    pairs.push_back(o2 + NF2_CPL_W2T + 4 * j + i2);
                    pairs.push_back(o1 + nf_cpl_off_W2(w) + i2 * 4 + j);
                }
            }
        }
        P.n_pairs = (int)(pairs.size() / 2);
        const size_t nb2 = P.ident.block2.size() * sizeof(float);
        if ((e = hipMalloc((void **)&P.d_ident2, nb2)) != hipSuccess ||
            (e = hipMemcpy(P.d_ident2, P.ident.block2.data(), nb2, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMalloc((void **)&P.d_pairs, pairs.size() * sizeof(int32_t))) != hipSuccess ||
            (e = hipMemcpy(P.d_pairs, pairs.data(), pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess)
            return fail_hip(e, "batch-statistics plan upload (matrix-core layout)");
    }
    P.ready = true;
    return NF_OK;
}

// all-reduce the `nvals` slotted sums of one statistics pass over the ranks (no-op without nf_set_sync): the totals come
// back as slot 0, so the consumer — the next launch's prologue or nf_bs_finalize — needs no other change than n x world
static int bs_sync(nf_handle *h, double *stats, int nvals, hipStream_t st)
{
    if (!h->sync_fn || h->sync_world < 2) return NF_OK;
    if (nvals > 64) return fail(NF_EINVAL, "cross-rank batch statistics cover coupling widths up to 32");
    hipError_t e = nf_launch_stats_compact(stats, nvals, h->sync_buf, st);
    if (e != hipSuccess) return fail_hip(e, "batch-statistics compaction");
    const int rc = h->sync_fn(h->sync_user, h->sync_buf, (int64_t)nvals, (void *)st);
    if (rc != 0) return fail(NF_EINVAL, "the all-reduce callback of nf_set_sync failed (status %d)", rc);
    if ((e = nf_launch_stats_scatter(stats, nvals, h->sync_buf, st)) != hipSuccess) return fail_hip(e, "batch-statistics scatter");
    return NF_OK;
}

// Where the statistics passes of the fused kernels do not reach — coupling widths beyond 32 (sidd/ArgParser.py:43 defaults to 512),
// and patches beyond the scalar-weight kernel's two LDS tiles at widths 5 .. 32 (64x64 at the paper's width 32) — the call is a
// layer-by-layer walk on the trainer's matrix-core GEMM path (nf_train.hip: nf_bs_wide_run): same moments, same outputs, one
// resident tensor; the evaluator is created on first use and grows with B.
static int run_batchstats_wide(nf_handle *h, int direction, const NfLaunch &a, const nf_cond *cond, float *moments_out, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(h->bs_mu);   // one evaluator per handle: calls serialise
    if (!h->bs) h->bs = new (std::nothrow) nf_bs_state();
    if (!h->bs) return fail(NF_ENOMEM, "out of host memory");
    nf_bs_state &S = *h->bs;
    int64_t cap = a.B;
    if (S.wide && nf_bs_wide_capacity(S.wide) < a.B) {
        // grow geometrically: a caller that ramps its batch size pays O(log B) re-allocations of the evaluator's tensors, not one per step
        cap = std::max<int64_t>(a.B, 2 * nf_bs_wide_capacity(S.wide));
        (void)nf_trainer_destroy(S.wide);
        S.wide = nullptr;
    }
    if (!S.wide) {
        nf_config cfg = h->cfg;
        cfg.device = h->device;
        int rc = nf_bs_wide_create(&cfg, h->layers.data(), h->raw.data(), h->raw.size(), cap, &S.wide);
        if (rc != NF_OK && cap > a.B) rc = nf_bs_wide_create(&cfg, h->layers.data(), h->raw.data(), h->raw.size(), a.B, &S.wide);   // (no room for the head-room)
        if (rc != NF_OK) return rc;
    }
    nf_bs_wide_args w;
    memset(&w, 0, sizeof(w));
    w.direction = direction;
    w.in = (a.flags & NF_K_PHILOX_IN) ? nullptr : a.in;
    w.y = a.y;
    w.B = a.B;
    w.cond = cond;
    w.seed = a.seed;
    w.patch_base = a.patch_base;
    w.in_scale = a.in_scale;
    w.out = a.out;
    w.nll_out = a.nll_out;
    w.sd_out = a.sd_out;
    w.ld_out = a.ld_out;
    w.sums = a.sums;
    w.prior = (a.flags & NF_K_PRIOR) != 0;
    w.moments_out = moments_out;
    w.sync_fn = h->sync_fn;
    w.sync_user = h->sync_user;
    w.sync_buf = h->sync_buf;
    w.sync_world = h->sync_world;
    return nf_bs_wide_run(S.wide, w, st);
}

static int run_batchstats(nf_handle *h, int direction, NfLaunch a, float *moments_out, hipStream_t st, const nf_cond *cond)
{
    if (h->cfg.flags & NF_CFG_FP16_CNN) return fail(NF_EINVAL, "batch-statistics mode is fp32 only");
    {
        // the generic schedule below runs on the scalar-weight kernel: both LDS tiles of a patch on one CU, one pixel per lane at
        // width 32; the width-4 schedule on the matrix-core kernel takes every size (images beyond 64x64 as overlapping tiles)
        const int pw = h->fwd.prog.width;
        const size_t tile_px = ((size_t)(a.H + 2) * (a.W + 2) + 1) & ~(size_t)1;
        const bool scalar_fits = sizeof(float) * (tile_px * (2 + (size_t)pw) + 64) <= 160 * 1024 && !(pw >= 32 && a.H * a.W > 1024) && h->scalar_ok && !h->fwd.tiled;
        const bool mc4 = pw == 4 && !h->fwd.block2.empty() && use_matrix_core();
        // NF_BS_WIDE=1: A/B aid; read per call on purpose (tests/test_gpu_batchstats.py flips it inside one process), =0 means off
        const char *bw = getenv("NF_BS_WIDE");
        // width 32 on 32x32 patches (the paper's coupling CNN on the training patch size): the evaluator runs the trainer's patch-resident
        // forward stages on the matrix cores (csrc/nf_train_pr.h) — 3 launches per coupling instead of the scalar-weight schedule
        // (1 024 patches: 3.7 -> 1.1 ms, 138: 1.24 -> 0.48).  NF_TRAIN_PR=0 keeps the scalar schedule.
        const char *pe = getenv("NF_TRAIN_PR");   // (read per call, like NF_BS_WIDE: the tests flip it inside one process)
        const bool pr_off = pe && atoi(pe) == 0;
        const bool pr32 = h->fwd.raw_width == 32 && a.H == 32 && a.W == 32 && !h->fwd.tiled && !pr_off;
        if (pw > 32 || pr32 || (!mc4 && !scalar_fits) || (bw && atoi(bw) != 0))
            return run_batchstats_wide(h, direction, a, cond, moments_out, st);
    }
    const int wr = h->fwd.raw_width;   // rows of moments_out are [4][wr]; the kernels' rows [4][prog.width] (zero-padded widths)
    // images beyond 64x64 (nf_device.h, "overlapping tiles"): the width-4 matrix-core schedule below, every launch tiled with
    // ONE halo for the whole call — 3 = the deepest launch (re-run coupling c-1, then l_1 of coupling c for its statistics) —
    // so that the tile grid, and with it the per-thread log-det carry, is the same in every launch
    const bool tiled = h->fwd.tiled;
    if (tiled && !(h->fwd.prog.width == 4 && !h->fwd.block2.empty() && use_matrix_core()))
        return fail(NF_EINVAL, "batch-statistics mode beyond 64x64 pixels (%dx%d given) runs on the width-4 matrix-core kernels only",
                    h->cfg.height, h->cfg.width);
    const int bs_halo = 3;
    const int bs_ny = tiled ? nf_tile_count(h->cfg.height, h->fwd.tile_h, bs_halo) : 1;
    const int bs_nx = tiled ? nf_tile_count(h->cfg.width, h->fwd.tile_w, bs_halo) : 1;
    const int bs_nt = bs_ny * bs_nx;
    std::lock_guard<std::mutex> lock(h->bs_mu);   // one scratch per handle: calls serialise
    if (!h->bs) h->bs = new (std::nothrow) nf_bs_state();
    if (!h->bs) return fail(NF_ENOMEM, "out of host memory");
    nf_bs_state &S = *h->bs;
    BsPlan &P = S.plan[direction];
    hipError_t e;
    if (!P.ready) {
        int rc = bs_build_plan(h, direction, P);
        if (rc != NF_OK) return rc;
    }
    const int n_cpl = (int)P.cpl_ops.size();
    const int w = P.ident.prog.width;
    const size_t nw1 = P.ident.block.size(), nw2 = P.ident.block2.size();
    auto grow = [&](float *&ptr, size_t &cap, size_t need) -> hipError_t {
        if (cap >= need) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t er = hipMalloc((void **)&ptr, need * sizeof(float));
        if (er == hipSuccess) cap = need;
        return er;
    };
    if ((e = grow(S.d_work, S.work_cap, nw1)) != hipSuccess || (nw2 && (e = grow(S.d_work2, S.work2_cap, nw2)) != hipSuccess))
        return fail_hip(e, "hipMalloc(batch-statistics parameters)");
    if (!S.d_stats && (e = hipMalloc((void **)&S.d_stats, NF_STATS_SLOTS * 64 * sizeof(double))) != hipSuccess) {
        S.d_stats = nullptr;
        return fail_hip(e, "hipMalloc(batch-statistics accumulators)");
    }
    if (!S.d_mom && (e = hipMalloc((void **)&S.d_mom, (size_t)std::max(n_cpl, 1) * 4 * 32 * sizeof(float))) != hipSuccess) {
        S.d_mom = nullptr;
        return fail_hip(e, "hipMalloc(batch-statistics moments)");
    }
    const size_t tensor = (size_t)a.B * a.H * a.W * 4;
    if (n_cpl > 1 && S.T_cap < tensor) {
        for (int i = 0; i < 2; ++i) {
            if (S.d_T[i]) (void)hipFree(S.d_T[i]);
            S.d_T[i] = nullptr;
        }
        S.T_cap = 0;
        if ((e = hipMalloc((void **)&S.d_T[0], tensor * sizeof(float))) != hipSuccess ||
            (e = hipMalloc((void **)&S.d_T[1], tensor * sizeof(float))) != hipSuccess)
            return fail_hip(e, "hipMalloc(batch-statistics scratch tensors)");
        S.T_cap = tensor;
    }
    // the log-det only matters to the outputs that contain it
    const bool carry = n_cpl > 1 && (a.nll_out || a.ld_out || a.sums);
    if (carry && (e = grow(S.d_carry, S.carry_cap, (size_t)a.B * bs_nt * 1024)) != hipSuccess) return fail_hip(e, "hipMalloc(batch-statistics log-det carry)");
    // the final launch: from the last resident tensor when there is one
    auto final_args = [&](NfLaunch &f) {
        if (n_cpl < 2) return;
        f.in = S.d_T[(n_cpl - 1) & 1];
        f.in_scale = 1.0f;
        f.flags &= ~(uint32_t)NF_K_PHILOX_IN;
        if (carry) {
            f.flags |= NF_K_CARRY_IN;
            f.ld_carry = S.d_carry;
        }
    };
    if (nw2 && P.ident.prog.width == 4 && use_matrix_core()) {
        // ---- width 4: every launch on the matrix-core kernel, the re-fold fused into the consumer's prologue ----
        // launch k gathers the sums of pass k into its own block of `d_stats_mc`; launch k+1 (every workgroup, in LDS)
        // turns them into moments and rescales the layer, workgroup 0 writes the patched image to the OTHER working
        // copy, which launch k+2 reads.  2 launches per coupling + the final fused pass, nothing else on the stream.
        const size_t passes = 2 * (size_t)n_cpl;
        if ((e = grow(S.d_work2b, S.work2b_cap, nw2)) != hipSuccess) return fail_hip(e, "hipMalloc(batch-statistics parameters)");
        if (S.stats_mc_passes < passes) {
            if (S.d_stats_mc) (void)hipFree(S.d_stats_mc);
            S.d_stats_mc = nullptr;
            S.stats_mc_passes = 0;
            if ((e = hipMalloc((void **)&S.d_stats_mc, passes * NF_STATS_SLOTS * 8 * sizeof(double))) != hipSuccess)
                return fail_hip(e, "hipMalloc(batch-statistics accumulators)");
            S.stats_mc_passes = passes;
        }
        if ((e = hipMemsetAsync(S.d_stats_mc, 0, passes * NF_STATS_SLOTS * 8 * sizeof(double), st)) != hipSuccess)
            return fail_hip(e, "batch-statistics set-up");
        const double n = (double)a.B * a.H * a.W * h->sync_world;   // with nf_set_sync the moments are over all ranks' patches
        const float *cur = P.d_ident2;          // parameter block the next launch reads
        float *bufs[2] = {S.d_work2, S.d_work2b};
        int nbuf = 0;
        // the fix a launch has to apply = the pass of the launch before it
        const double *pend_stats = nullptr;
        int pend_off = 0, pend_stage = 0;
        float *pend_mom = nullptr;
        auto launch = [&](const NfProgram &prog, NfLaunch &s) -> hipError_t {
            s.params = cur;
            s.n_params = (int32_t)nw2;
            s.flags |= NF_K_BATCHSTATS;
            if (tiled) {   // every "patch" of the launch is one tile of an image (tensors stay image-shaped)
                s.flags |= NF_K_TILED;
                s.img_H = h->cfg.height;
                s.img_W = h->cfg.width;
                s.H = h->fwd.tile_h;
                s.W = h->fwd.tile_w;
                s.tile_ny = bs_ny;
                s.tile_nx = bs_nx;
                s.tile_halo = bs_halo;
                s.B = a.B * bs_nt;
            }
            s.fix_stats = pend_stats;
            s.fix_n = n;
            s.fix_off = pend_off;
            s.fix_stage = pend_stage;
            s.fix_mom_out = pend_mom;
            s.fix_params_out = pend_stats ? bufs[nbuf] : nullptr;
            hipError_t er = nf_launch_flow(prog, s, h->n_cu, st, true);
            if (pend_stats) {
                cur = bufs[nbuf];
                nbuf ^= 1;
            }
            return er;
        };
        const float *const in0 = a.in;
        const float in_scale0 = a.in_scale;
        const uint32_t flags0 = a.flags;
        for (int c = 0; c < n_cpl; ++c) {
            float *mom = S.d_mom + (size_t)P.cpl_row[c] * 16;
            for (int stage = 1; stage <= 2; ++stage) {
                const NfProgram &prog = stage == 1 ? P.progA2[c] : P.progB2[c];
                NfLaunch s = a;
                s.nll_out = s.sd_out = s.ld_out = nullptr;
                s.sums = nullptr;
                s.stats = S.d_stats_mc + (size_t)(2 * c + stage - 1) * NF_STATS_SLOTS * 8;
                s.stats_op = prog.n_ops - 1;
                s.stats_stage = stage;
                const int tin = stage == 1 ? (c > 0 ? c - 1 : 0) : c;
                const bool from_input = tin == 0;
                s.in = from_input ? in0 : S.d_T[tin & 1];
                s.in_scale = from_input ? in_scale0 : 1.0f;
                s.flags = from_input ? flags0 : (flags0 & ~(uint32_t)NF_K_PHILOX_IN);
                s.out = (stage == 1 && c > 0) ? S.d_T[c & 1] : nullptr;
                if (carry && s.out) {
                    s.flags |= NF_K_CARRY_OUT | (c > 1 ? NF_K_CARRY_ADD : 0u);
                    s.ld_carry = S.d_carry;
                }
                if ((e = launch(prog, s)) != hipSuccess) return fail_hip(e, "batch-statistics launch");
                {
                    const int rc = bs_sync(h, s.stats, 8, st);
                    if (rc != NF_OK) return rc;
                }
                pend_stats = s.stats;
                pend_off = P.ident.prog2.ops[P.cpl_ops[c]].off;
                pend_stage = stage;
                pend_mom = mom + (stage == 1 ? 0 : 8);
            }
        }
        a.ld_const = a.ld_const + (direction == 0 ? P.ident.ld_const : 0.0);
        final_args(a);
        if (tiled) {
            // the final launch leaves per-tile sums; one more kernel forms the per-image results (as launch_tiled)
            const bool want = direction == 0 && (a.nll_out || a.sd_out || a.ld_out || a.sums);
            NfLaunch f = a;
            f.nll_out = f.sd_out = f.ld_out = nullptr;
            f.sums = nullptr;
            float *part = nullptr;
            if (want && (e = hipMallocAsync((void **)&part, (size_t)a.B * bs_nt * 4 * sizeof(float), st)) != hipSuccess)
                return fail_hip(e, "hipMallocAsync(tile sums)");
            f.tile_part = part;
            e = launch(n_cpl > 1 ? P.progF2 : P.ident.prog2, f);
            if (e == hipSuccess && want) {
                NfTileParts tp;
                memset(&tp, 0, sizeof(tp));
                tp.n_seg = 1;
                tp.nt[0] = bs_nt;
                e = nf_launch_tile_combine(part, tp, a.B, (double)h->cfg.height * h->cfg.width * kC, a.ld_const, a.flags, a.nll_out, a.sd_out,
                                           a.ld_out, a.sums, st);
            }
            if (part) {
                hipError_t e2 = hipFreeAsync(part, st);
                if (e == hipSuccess) e = e2;
            }
            if (e != hipSuccess) return fail_hip(e, "batch-statistics final launch");
        } else if ((e = launch(n_cpl > 1 ? P.progF2 : P.ident.prog2, a)) != hipSuccess) return fail_hip(e, "batch-statistics final launch");
        std::vector<float> mom_h((size_t)std::max(n_cpl, 1) * 16);
        if (moments_out && n_cpl &&
            (e = hipMemcpyAsync(mom_h.data(), S.d_mom, (size_t)n_cpl * 16 * sizeof(float), hipMemcpyDeviceToHost, st)) != hipSuccess) {
            (void)hipStreamSynchronize(st);
            return fail_hip(e, "batch-statistics moments readback");
        }
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail_hip(e, "batch-statistics final sync");   // the scratch is reused
        if (moments_out && n_cpl) {
            for (int c = 0; c < n_cpl; ++c)          // rows are in NLL layer order, like P.shift
                for (int j = 0; j < 4; ++j) {
                    mom_h[(size_t)c * 16 + j] += P.shift[(size_t)c * 8 + j];
                    mom_h[(size_t)c * 16 + 8 + j] += P.shift[(size_t)c * 8 + 4 + j];
                }
            for (int c = 0; c < n_cpl; ++c)          // the caller's rows hold the model's own channels
                for (int q = 0; q < 4; ++q) memcpy(moments_out + ((size_t)c * 4 + q) * wr, mom_h.data() + (size_t)c * 16 + q * 4, (size_t)wr * sizeof(float));
        }
        return NF_OK;
    }
    {   // the generic schedule runs on the scalar-weight kernel: both LDS tiles of a patch on one CU, one pixel per lane at width 32
        const size_t tile_px = ((size_t)(a.H + 2) * (a.W + 2) + 1) & ~(size_t)1;
        const size_t lds = sizeof(float) * (tile_px * (2 + (size_t)w) + 64);
        if (lds > 160 * 1024 || (w >= 32 && a.H * a.W > 1024))
            return fail(NF_EINVAL, "batch-statistics mode at coupling width %d covers patches of up to %s pixels (%dx%d given); "
                        "evaluation mode has no such limit", w, w >= 32 ? "1024" : w == 16 ? "~2270 (e.g. 45x48)" : "~4090 (e.g. 62x62)",
                        a.H, a.W);
    }
    if ((e = hipMemcpyAsync(S.d_work, P.d_ident, nw1 * sizeof(float), hipMemcpyDeviceToDevice, st)) != hipSuccess ||
        (nw2 && (e = hipMemcpyAsync(S.d_work2, P.d_ident2, nw2 * sizeof(float), hipMemcpyDeviceToDevice, st)) != hipSuccess) ||
        (e = hipMemsetAsync(S.d_stats, 0, NF_STATS_SLOTS * 2 * (size_t)w * sizeof(double), st)) != hipSuccess)
        return fail_hip(e, "batch-statistics set-up");

    const double n = (double)a.B * a.H * a.W * h->sync_world;
    const double ld_call = a.ld_const;
    const float *const in0 = a.in;
    const float in_scale0 = a.in_scale;
    const uint32_t flags0 = a.flags;
    for (int c = 0; c < n_cpl; ++c) {
        const int blk = P.ident.prog.ops[P.cpl_ops[c]].off;
        float *mom = S.d_mom + (size_t)P.cpl_row[c] * 4 * w;
        for (int stage = 1; stage <= 2; ++stage) {
            const NfProgram &prog = stage == 1 ? P.progA[c] : P.progB[c];
            NfLaunch s = a;
            s.params = S.d_work;
            s.n_params = 0;
            s.nll_out = s.sd_out = s.ld_out = nullptr;
            s.sums = nullptr;
            s.stats = S.d_stats;
            s.stats_op = prog.n_ops - 1;
            s.stats_stage = stage;
            // launch A_c reads T_{c-1} (it re-runs coupling c-1), launch B_c reads T_c;  T_0 = the caller's input
            const int tin = stage == 1 ? (c > 0 ? c - 1 : 0) : c;
            const bool from_input = tin == 0;
            s.in = from_input ? in0 : S.d_T[tin & 1];
            s.in_scale = from_input ? in_scale0 : 1.0f;
            s.flags = from_input ? flags0 : (flags0 & ~(uint32_t)NF_K_PHILOX_IN);
            s.out = (stage == 1 && c > 0) ? S.d_T[c & 1] : nullptr;
            if (carry && s.out) {
                s.flags |= NF_K_CARRY_OUT | (c > 1 ? NF_K_CARRY_ADD : 0u);
                s.ld_carry = S.d_carry;
            }
            if ((e = nf_launch_flow(prog, s, h->n_cu, st, false)) != hipSuccess) return fail_hip(e, "batch-statistics launch");
            {
                const int rc = bs_sync(h, S.d_stats, 2 * w, st);
                if (rc != NF_OK) return rc;
            }
            float *Wm = S.d_work + blk + (stage == 1 ? nf_cpl_off_W1(w) : nf_cpl_off_W2(w));
            float *Bv = S.d_work + blk + (stage == 1 ? nf_cpl_off_B1(w) : nf_cpl_off_B2(w));
            if ((e = nf_launch_bs_finalize(S.d_stats, w, n, Wm, stage == 1 ? 18 : w, Bv, mom + (stage == 1 ? 0 : 2 * w),
                                           mom + (stage == 1 ? w : 3 * w), st)) != hipSuccess)
                return fail_hip(e, "batch-statistics finalise");
        }
    }

    // final pass with the batch moments folded in
    const bool mc = nw2 != 0 && use_matrix_core();
    if (mc && (e = nf_launch_gather(S.d_work2, S.d_work, P.d_pairs, P.n_pairs, st)) != hipSuccess)
        return fail_hip(e, "batch-statistics re-layout");
    a.params = mc ? S.d_work2 : S.d_work;
    a.n_params = mc ? (int32_t)nw2 : 0;
    a.ld_const = ld_call + (direction == 0 ? P.ident.ld_const : 0.0);
    if (!mc) final_args(a);   // (a final pass on the matrix-core kernel maps pixels to threads differently: whole stack, no carry)
    if ((e = nf_launch_flow(mc ? P.ident.prog2 : n_cpl > 1 ? P.progF : P.ident.prog, a, h->n_cu, st, mc)) != hipSuccess)
        return fail_hip(e, "batch-statistics final launch");
    std::vector<float> mom_h((size_t)std::max(n_cpl, 1) * 4 * w);
    if (moments_out && n_cpl &&
        (e = hipMemcpyAsync(mom_h.data(), S.d_mom, (size_t)n_cpl * 4 * w * sizeof(float), hipMemcpyDeviceToHost, st)) != hipSuccess) {
        (void)hipStreamSynchronize(st);
        return fail_hip(e, "batch-statistics moments readback");
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail_hip(e, "batch-statistics final sync");   // the scratch is reused
    if (moments_out && n_cpl) {
        for (int c = 0; c < n_cpl; ++c)              // [mean1, var1, mean2, var2][w] per coupling, NLL layer order
            for (int j = 0; j < w; ++j) {
                mom_h[(size_t)c * 4 * w + j] += P.shift[(size_t)c * 2 * w + j];
                mom_h[(size_t)c * 4 * w + 2 * w + j] += P.shift[(size_t)c * 2 * w + w + j];
            }
        for (int c = 0; c < n_cpl; ++c)              // the caller's rows hold the model's own channels
            for (int q = 0; q < 4; ++q)
                memcpy(moments_out + ((size_t)c * 4 + q) * wr, mom_h.data() + ((size_t)c * 4 + q) * w, (size_t)wr * sizeof(float));
    }
    return NF_OK;
}

static void nf_bs_destroy(nf_bs_state *s) { delete s; }

int nf_nll_batchstats(nf_handle *h, const float *x, const float *y, int64_t B, const nf_cond *cond, float *nll_out,
                      float *sd_out, float *logdet_out, float *z_out, double *sums_out, uint32_t flags,
                      float *moments_out, void *stream)
{
    NfLaunch a;
    int rc = nll_args(h, x, y, B, cond, nll_out, sd_out, logdet_out, z_out, sums_out, flags, a);
    if (rc != NF_OK) return rc;
    if (B == 0) return fail(NF_EINVAL, "batch statistics of an empty batch are undefined");
    DeviceGuard guard;
    if ((rc = guard.enter(h->device)) != NF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (sums_out && !(flags & NF_ACCUMULATE)) {
        const size_t nb = (flags & NF_SUMS_WIDE) ? (size_t)NF_SUMS_SLOTS * NF_SUMS_STRIDE : 3;
        hipError_t e = hipMemsetAsync(sums_out, 0, nb * sizeof(double), st);
        if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(sums)");
    }
    return run_batchstats(h, 0, a, moments_out, st, cond);
}

int nf_sample_batchstats(nf_handle *h, const float *y, const float *eps, uint64_t seed, int64_t patch_index_base,
                         float temp, int64_t B, const nf_cond *cond, float *x_out, float *moments_out, void *stream)
{
    NfLaunch a;
    int rc = sample_args(h, y, eps, seed, patch_index_base, temp, B, cond, x_out, a);
    if (rc != NF_OK) return rc;
    if (B == 0) return fail(NF_EINVAL, "batch statistics of an empty batch are undefined");
    DeviceGuard guard;
    if ((rc = guard.enter(h->device)) != NF_OK) return rc;
    return run_batchstats(h, 1, a, moments_out, (hipStream_t)stream, cond);
}

int nf_set_sync(nf_handle *h, nf_allreduce_fn fn, void *user, double *sync_buf, int32_t world_size)
{
    if (!h) return fail(NF_EINVAL, "handle is NULL");
    if (fn && (!sync_buf || world_size < 1)) return fail(NF_EINVAL, "nf_set_sync needs a device buffer of 64 doubles and world_size >= 1");
    std::lock_guard<std::mutex> lock(h->bs_mu);
    h->sync_fn = fn;
    h->sync_user = user;
    h->sync_buf = fn ? sync_buf : nullptr;
    h->sync_world = fn ? world_size : 1;
    return NF_OK;
}

int nf_sums_reduce(const double *wide, double *out3, uint32_t flags, void *stream)
{
    if (!wide || !out3) return fail(NF_EINVAL, "null argument");
    hipError_t e = nf_launch_sums_reduce(wide, out3, (flags & NF_ACCUMULATE) != 0, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "nf_sums_reduce launch");
    return NF_OK;
}

int nf_tile_segments(const nf_config *cfg, const nf_layer_desc *layers, const float *params, size_t n_params, int32_t direction,
                     int32_t *out5, int32_t cap)
{
    if (direction != 0 && direction != 1) return fail(NF_EINVAL, "direction must be 0 or 1");
    Built b;
    int rc = build_program(cfg, layers, params, n_params, direction, b);
    if (rc != NF_OK) return rc;
    if (!b.tiled) return 0;
    for (int i = 0; i < (int)b.segs.size() && i < cap && out5; ++i) {
        const Built::TileSeg &g = b.segs[i];
        const int32_t v[5] = {g.op0, g.op1, g.halo, g.ny, g.nx};
        memcpy(out5 + 5 * i, v, sizeof(v));
    }
    return (int)b.segs.size();
}

int nf_tile_plan(int32_t size, int32_t tile, int32_t halo, int32_t *origin, int32_t *core0, int32_t *core1, int32_t cap)
{
    if (size < 1 || tile < 1 || halo < 0 || tile > size) return fail(NF_EINVAL, "bad tile plan (size %d, tile %d, halo %d)", size, tile, halo);
    if (size > tile && tile - 2 * halo < 1) return fail(NF_EINVAL, "a %d-pixel tile has no core with a halo of %d", tile, halo);
    const int n = nf_tile_count(size, tile, halo);
    for (int i = 0; i < n && i < cap; ++i) {
        if (origin) origin[i] = nf_tile_origin(i, size, tile, halo);
        if (core0) core0[i] = nf_tile_core0(i, size, tile, halo);
        if (core1) core1[i] = nf_tile_core1(i, n, size, tile, halo);
    }
    return n;
}

int nf_sample_eps(uint64_t seed, int64_t patch_index_base, int64_t B, int32_t height, int32_t width, float *eps_out, void *stream)
{
    if (B < 0 || height < 1 || width < 1) return fail(NF_EINVAL, "bad shape");
    if (!eps_out || !aligned16(eps_out)) return fail(NF_EINVAL, "eps_out must be a 16-byte aligned device pointer");
    hipError_t e = nf_launch_eps(seed, patch_index_base, B, height * width, eps_out, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "nf_sample_eps launch");
    return NF_OK;
}

int nf_synth_patches(uint64_t seed, int64_t patch_index_base, int64_t B, int32_t height, int32_t width, float beta1,
                     float beta2, float *y_out, float *x_out, void *stream)
{
    if (B < 0 || height < 1 || width < 1) return fail(NF_EINVAL, "bad shape");
    if (!y_out && !x_out) return fail(NF_EINVAL, "both outputs are NULL");
    hipError_t e = nf_launch_synth(seed, patch_index_base, B, height * width, beta1, beta2, y_out, x_out, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "nf_synth_patches launch");
    return NF_OK;
}

}  // extern "C"
