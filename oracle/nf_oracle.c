/*
 * CPU ORACLE in plain C — TEST INFRASTRUCTURE ONLY (see oracle/nf_oracle.py for the
 * numpy fp64 restatement this file is pinned to, and for the "parity unpinned" note).
 *
 * A second, independent restatement of the reference's bijector stack in the reference's
 * own op order and precision (fp32 tensors, one op at a time, nothing folded), written so
 * that it can (a) check the numpy oracle, (b) check the HIP path on FULL batches in
 * seconds, (c) serve as a multi-core CPU baseline (OpenMP over patches).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Reference (paths under /root/reference):
 *   conv1x1          borealisflows/layers.py:108-130        (A / A^-1 supplied by the caller)
 *   coupling + CNN   borealisflows/layers.py:275-291, 355-375, 463-497, 555-613, 651-674
 *   batch_norm eval  borealisflows/layers.py:378-401
 *   sdn-type scale   noise_flow_layers/cond_utils.py:41-52, 178-239 (scalars supplied)
 *   gain4 / gain     noise_flow_layers/AffineCouplingGainEx4.py:114-127, AffineCouplingGain.py:113-127
 *   prior, sd_z      borealisflows/noise_flow_model.py:458-497, 525-541
 *
 * Layer stream (floats), one record per layer, NLL order:
 *   type 1 CONV1X1 : A[16] (row-major [c][k]), A_inv[16], log_abs_det
 *   type 2 COUPLING: width w, then l_1/W[3][3][2][w], l_1/b[w], bn1_mean[w], bn1_var[w],
 *                    l_2/W[w][w], l_2/b[w], bn2_mean[w], bn2_var[w],
 *                    l_last/W[3][3][w+1][4], l_last/b[4], l_last/logs[4], rescaling_scale
 *   type 3 SDN     : a, b          scale = sqrt(a*y + b)         (per-call scalars)
 *   type 4 SCALE   : g, ld         z /= g ; objective += ld      (gain4: ld = -H*W*C*log g;
 *                                                                 plain gain: ld = -log g)
 * Each record starts with its type as a float.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define C 4
#define C2 2
#define BN_EPS 1e-4f

typedef struct {
    int type, w;
    const float *p;
} layer_t;

static int parse(const float *s, int n_layers, layer_t *L)
{
    const float *q = s;
    for (int i = 0; i < n_layers; ++i) {
        L[i].type = (int)q[0];
        L[i].w = 0;
        q += 1;
        switch (L[i].type) {
        case 1: L[i].p = q; q += 33; break;
        case 2: {
            const int w = (int)q[0];
            L[i].w = w;
            L[i].p = q + 1;
            q += 1 + 18 * w + 3 * w + w * w + 3 * w + 36 * (w + 1) + 9;
            break;
        }
        case 3: L[i].p = q; q += 2; break;
        case 4: L[i].p = q; q += 2; break;
        default: return -1;
        }
    }
    return 0;
}

/* shift/raw of the coupling CNN on z0 = z[..., :2]; out[H*W][4]; scratch h1,h2: [H*W][w] */
static void coupling_cnn(const float *z, int H, int W, const layer_t *L, float *out, float *h1, float *h2)
{
    const int w = L->w;
    const float *W1 = L->p, *b1 = W1 + 18 * w, *m1 = b1 + w, *v1 = m1 + w;
    const float *W2 = v1 + w, *b2 = W2 + w * w, *m2 = b2 + w, *v2 = m2 + w;
    const float *W3 = v2 + w, *b3 = W3 + 36 * (w + 1), *logs = b3 + 4;
    /* l_1: 3x3 SAME on 2 channels, + b, BN, ReLU */
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float *h = h1 + (size_t)(i * W + j) * w;
            for (int k = 0; k < w; ++k) h[k] = 0.0f;
            for (int di = 0; di < 3; ++di)
                for (int dj = 0; dj < 3; ++dj) {
                    const int r = i + di - 1, c = j + dj - 1;
                    if (r < 0 || r >= H || c < 0 || c >= W) continue;
                    const float *zz = z + (size_t)(r * W + c) * C;
                    const float *ww = W1 + ((di * 3 + dj) * 2) * w;
                    for (int k = 0; k < w; ++k) h[k] += zz[0] * ww[k] + zz[1] * ww[w + k];
                }
            for (int k = 0; k < w; ++k) {
                float t = h[k] + b1[k];
                t = (t - m1[k]) / sqrtf(v1[k] + BN_EPS);
                h[k] = t > 0.0f ? t : 0.0f;
            }
        }
    /* l_2: 1x1, + b, BN, ReLU */
    for (int p = 0; p < H * W; ++p) {
        const float *a = h1 + (size_t)p * w;
        float *h = h2 + (size_t)p * w;
        for (int k = 0; k < w; ++k) {
            float t = 0.0f;
            for (int i = 0; i < w; ++i) t += a[i] * W2[i * w + k];
            t += b2[k];
            t = (t - m2[k]) / sqrtf(v2[k] + BN_EPS);
            h[k] = t > 0.0f ? t : 0.0f;
        }
    }
    /* l_last: zero pad 1 + edge-indicator channel, 3x3 VALID, + b, * exp(3 logs) */
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int di = 0; di < 3; ++di)
                for (int dj = 0; dj < 3; ++dj) {
                    const int r = i + di - 1, c = j + dj - 1;   /* un-padded coordinates of the tap */
                    const float *ww = W3 + ((di * 3 + dj) * (w + 1)) * 4;
                    if (r < 0 || r >= H || c < 0 || c >= W) {   /* on the ring: data 0, indicator 1 */
                        for (int k = 0; k < 4; ++k) o[k] += ww[w * 4 + k];
                    } else {
                        const float *a = h2 + (size_t)(r * W + c) * w;
                        for (int q = 0; q < w; ++q)
                            for (int k = 0; k < 4; ++k) o[k] += a[q] * ww[q * 4 + k];
                    }
                }
            float *dst = out + (size_t)(i * W + j) * 4;
            for (int k = 0; k < 4; ++k) dst[k] = (o[k] + b3[k]) * expf(logs[k] * 3.0f);
        }
}

static int max_width(const layer_t *L, int n)
{
    int w = 1;
    for (int i = 0; i < n; ++i)
        if (L[i].w > w) w = L[i].w;
    return w;
}

/* NLL direction: NoiseFlow._loss.  Returns 0, -1 on a malformed stream. */
int nfo_nll(const float *stream, int n_layers, int H, int W, const float *x, const float *y, long B,
            float *nll_out, float *sd_out, float *z_out)
{
    layer_t L[128];
    if (n_layers > 128 || parse(stream, n_layers, L)) return -1;
    const int HW = H * W, wmax = max_width(L, n_layers);
    int err = 0;
#pragma omp parallel
    {
        float *z = (float *)malloc(sizeof(float) * HW * C);
        float *o = (float *)malloc(sizeof(float) * HW * 4);
        float *h1 = (float *)malloc(sizeof(float) * HW * wmax);
        float *h2 = (float *)malloc(sizeof(float) * HW * wmax);
#pragma omp for schedule(static)
        for (long b = 0; b < B; ++b) {
            memcpy(z, x + (size_t)b * HW * C, sizeof(float) * HW * C);
            const float *yy = y ? y + (size_t)b * HW * C : NULL;
            double obj = 0.0;
            for (int l = 0; l < n_layers; ++l) {
                const float *p = L[l].p;
                if (L[l].type == 1) {
                    for (int q = 0; q < HW; ++q) {
                        float in[4], *zz = z + (size_t)q * C;
                        memcpy(in, zz, sizeof(in));
                        for (int k = 0; k < 4; ++k) zz[k] = in[0] * p[k] + in[1] * p[4 + k] + in[2] * p[8 + k] + in[3] * p[12 + k];
                    }
                    obj += (double)p[32] * HW;
                } else if (L[l].type == 2) {
                    coupling_cnn(z, H, W, &L[l], o, h1, h2);
                    const float s = p[18 * L[l].w + 3 * L[l].w + L[l].w * L[l].w + 3 * L[l].w + 36 * (L[l].w + 1) + 8];
                    float ld = 0.0f;
                    for (int q = 0; q < HW; ++q)
                        for (int c = 0; c < C2; ++c) {
                            const float ls = s * tanhf(o[q * 4 + 2 + c]);
                            z[q * C + 2 + c] = z[q * C + 2 + c] * expf(ls) + o[q * 4 + c];
                            ld += ls;
                        }
                    obj += ld;
                } else if (L[l].type == 3) {
                    if (!yy) { err = 1; continue; }
                    float ld = 0.0f;
                    for (int q = 0; q < HW * C; ++q) {
                        const float sc = sqrtf(p[0] * yy[q] + p[1]);
                        z[q] = z[q] / sc;
                        ld += logf(sc);
                    }
                    obj -= ld;
                } else {
                    for (int q = 0; q < HW * C; ++q) z[q] = z[q] / p[0];
                    obj += p[1];
                }
            }
            double s1 = 0.0, s2 = 0.0;
            for (int q = 0; q < HW * C; ++q) {
                s1 += z[q];
                s2 += (double)z[q] * z[q];
            }
            const double n = (double)HW * C;
            obj += -0.5 * (n * 1.8378770664093453 + s2);
            const double mean = s1 / n;
            double var = s2 / n - mean * mean;
            if (var < 0) var = 0;
            if (nll_out) nll_out[b] = (float)(-obj);
            if (sd_out) sd_out[b] = (float)sqrt(var);
            if (z_out) memcpy(z_out + (size_t)b * HW * C, z, sizeof(float) * HW * C);
        }
        free(z); free(o); free(h1); free(h2);
    }
    return err ? -2 : 0;
}

/* sampling direction: NoiseFlow.sample / forward with caller-supplied eps */
int nfo_sample(const float *stream, int n_layers, int H, int W, const float *eps, float temp, const float *y,
               long B, float *x_out)
{
    layer_t L[128];
    if (n_layers > 128 || parse(stream, n_layers, L)) return -1;
    const int HW = H * W, wmax = max_width(L, n_layers);
    int err = 0;
#pragma omp parallel
    {
        float *z = (float *)malloc(sizeof(float) * HW * C);
        float *o = (float *)malloc(sizeof(float) * HW * 4);
        float *h1 = (float *)malloc(sizeof(float) * HW * wmax);
        float *h2 = (float *)malloc(sizeof(float) * HW * wmax);
#pragma omp for schedule(static)
        for (long b = 0; b < B; ++b) {
            const float *yy = y ? y + (size_t)b * HW * C : NULL;
            for (int q = 0; q < HW * C; ++q) z[q] = eps[(size_t)b * HW * C + q] * temp;
            for (int l = n_layers - 1; l >= 0; --l) {
                const float *p = L[l].p;
                if (L[l].type == 1) {
                    for (int q = 0; q < HW; ++q) {
                        float in[4], *zz = z + (size_t)q * C;
                        memcpy(in, zz, sizeof(in));
                        for (int k = 0; k < 4; ++k)
                            zz[k] = in[0] * p[16 + k] + in[1] * p[20 + k] + in[2] * p[24 + k] + in[3] * p[28 + k];
                    }
                } else if (L[l].type == 2) {
                    coupling_cnn(z, H, W, &L[l], o, h1, h2);
                    const float s = p[18 * L[l].w + 3 * L[l].w + L[l].w * L[l].w + 3 * L[l].w + 36 * (L[l].w + 1) + 8];
                    for (int q = 0; q < HW; ++q)
                        for (int c = 0; c < C2; ++c) {
                            const float ls = s * tanhf(o[q * 4 + 2 + c]);
                            z[q * C + 2 + c] = (z[q * C + 2 + c] - o[q * 4 + c]) * expf(-ls);
                        }
                } else if (L[l].type == 3) {
                    if (!yy) { err = 1; continue; }
                    for (int q = 0; q < HW * C; ++q) z[q] = z[q] * sqrtf(p[0] * yy[q] + p[1]);
                } else {
                    for (int q = 0; q < HW * C; ++q) z[q] = z[q] * p[0];
                }
            }
            memcpy(x_out + (size_t)b * HW * C, z, sizeof(float) * HW * C);
        }
        free(z); free(o); free(h1); free(h2);
    }
    return err ? -2 : 0;
}

void nfo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int nfo_threads(void)
{
    int n = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp master
        n = omp_get_num_threads();
    }
#endif
    return n;
}
