// Training step at coupling widths beyond 32 (part of nf_train.hip).
//
// sidd/ArgParser.py:43 defaults --width to Glow's 512 and train_noise_flow.py:50-77,187-198 trains at whatever width is set.
// The layer kernels of nf_train.hip keep (w+1)*4 accumulators per thread and the matrix-core stages of nf_train_wide.h hold one
// 32-channel tile per operand: neither exists beyond width 32.  At these widths a coupling IS GEMM-shaped — l_2 is
// [pixels x w] . [w x w], 90 % of the arithmetic at 512 — so the dense products of the step run as plain library GEMMs
// (rocBLAS sgemm, exact fp32; loaded with dlopen when the first wide trainer is created, so the library itself does not depend
// on it) and everything that is not a plain GEMM is a hand-written kernel over [pixel][w] tensors of run-time width:
//
//   forward   Z18 = the 3x3 x 2-channel windows of the pass-through half            k_g_gather18
//             h1 = Z18 . W1 (K = 18)                                                sgemm
//             batch sums of h1 + b1 (slotted; h1 stays without its bias in memory)  k_g_bias_stats   -> k_bn_fin (moments, EMA)
//             a1 = relu(bn1(h1 + b1))                                               k_g_bn_relu
//             h2 = a1 . W2                                                          sgemm
//             batch sums of h2 + b2; a2 = relu(bn2(h2 + b2))                        k_g_bias_stats, k_bn_fin, k_g_bn_relu
//             P = a2 . W3r  (W3r[i][tap*4+k] = l_last/W[tap][i][k], 36 columns)     k_g_pack_w3, sgemm
//             u = gather of the 9 taps of P + edge channel + b3; affine transform   k_g_c3_fwd
//   backward  affine / tanh / exp(3 logs) backward -> gu, d b3, d logs, d scale     k_g_c3_bwd
//             G36[p][tap*4+k] = gu[p - tap][k]; d edge-channel weights              k_g_gather36
//             d l_last/W = a2^T . G36 ;  g_a2 = G36 . W3r^T                         sgemm x 2
//             the two batch sums of BN2's backward (mask from h2, read-only)        k_g_mask_stats   -> k_bnb_fin
//             g_h2 = BN2 backward of the masked g_a2; d b2                          k_g_bn_bwd
//             d l_2/W = a1^T . g_h2 ;  g_a1 = g_h2 . W2^T                           sgemm x 2
//             mask + sums, g_h1 = BN1 backward, d b1                                k_g_mask_stats, k_bnb_fin, k_g_bn_bwd
//             d l_1/W = Z18^T . g_h1 ;  Q = g_h1 . W1^T (18 columns)                sgemm x 2
//             d z0 += gather of Q; the folded Conv2d1x1 backward                    k_g_c1_dz
//
// Gradients of the three filters come out of the GEMMs whole and go straight into the fp64 gradient vector (k_g_store_grad);
// every other reduction of the step keeps the trainer's slotted partial sums, so the batch statistics can be synchronised
// across ranks exactly as at the other widths.  One stream, no side work: a wide step is tens of milliseconds of GEMMs.
//
// Replaces (reference, /root/reference): train_noise_flow.py:64-66 with borealisflows/layers.py:452-498 at hps.width > 32.
#pragma once
#include <dlfcn.h>
#include <rocblas/rocblas.h>   // types and enumerators only: the functions are resolved with dlsym

namespace {

// the widths with kernels of their own (nf_train.hip, nf_train_tiled.h, nf_train_wide.h); every other width 1 .. 512 runs here
// (NF_TRAIN_GEMM=1 in the environment sends those four here as well: an A/B switch for tests and profiles)
inline bool gemm_width(int w)
{
    static const bool all = [] { const char *e = getenv("NF_TRAIN_GEMM"); return e && atoi(e) != 0; }();
    return all || (w != 4 && w != 8 && w != 16 && w != 32);
}

struct RocBlas {
    void *lib = nullptr;
    rocblas_status (*create)(rocblas_handle *) = nullptr;
    rocblas_status (*destroy)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*set_atomics)(rocblas_handle, rocblas_atomics_mode) = nullptr;   // optional
    rocblas_status (*sgemm)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const float *,
                            const float *, rocblas_int, const float *, rocblas_int, const float *, float *, rocblas_int) = nullptr;
    rocblas_status (*sgemm_sb)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const float *,
                               const float *, rocblas_int, rocblas_stride, const float *, rocblas_int, rocblas_stride, const float *, float *,
                               rocblas_int, rocblas_stride, rocblas_int) = nullptr;
};

// process-wide, loaded once (never unloaded)
inline const RocBlas *rocblas_api()
{
    static const RocBlas api = [] {
        RocBlas r;
        for (const char *name : {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.create = reinterpret_cast<decltype(r.create)>(dlsym(r.lib, "rocblas_create_handle"));
            r.destroy = reinterpret_cast<decltype(r.destroy)>(dlsym(r.lib, "rocblas_destroy_handle"));
            r.set_stream = reinterpret_cast<decltype(r.set_stream)>(dlsym(r.lib, "rocblas_set_stream"));
            r.set_atomics = reinterpret_cast<decltype(r.set_atomics)>(dlsym(r.lib, "rocblas_set_atomics_mode"));
            r.sgemm = reinterpret_cast<decltype(r.sgemm)>(dlsym(r.lib, "rocblas_sgemm"));
            r.sgemm_sb = reinterpret_cast<decltype(r.sgemm_sb)>(dlsym(r.lib, "rocblas_sgemm_strided_batched"));
            if (!r.create || !r.destroy || !r.set_stream || !r.sgemm || !r.sgemm_sb) r.lib = nullptr;
        }
        return r;
    }();
    return api.lib ? &api : nullptr;
}

// Row-major C[M x N] = op(A) . op(B) (+ beta C): rocBLAS is column-major, so the call computes C^T = op(B)^T . op(A)^T.
// ta / tb: the row-major operand is stored transposed ([K x M] / [N x K]).
inline bool gemm_rm(nf_trainer *t, hipStream_t st, bool ta, bool tb, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                    const float *B, int64_t ldb, float *C, int64_t ldc)
{
    const RocBlas *rb = rocblas_api();
    const float one = 1.0f, zero = 0.0f;
    if (!rb || !t->blas) return false;
    rocblas_handle h = (rocblas_handle)t->blas;
    if (rb->set_stream(h, st) != rocblas_status_success) return false;
    return rb->sgemm(h, tb ? rocblas_operation_transpose : rocblas_operation_none, ta ? rocblas_operation_transpose : rocblas_operation_none,
                     (rocblas_int)N, (rocblas_int)M, (rocblas_int)K, &one, B, (rocblas_int)ldb, A, (rocblas_int)lda, &zero, C,
                     (rocblas_int)ldc) == rocblas_status_success;
}

// Filter gradients: row-major C[M x N] = A^T . B with A [K x M], B [K x N] and K = the PIXELS of the minibatch (1e5 .. 1e6) against
// M, N of 18 .. 512 — one GEMM with a handful of output tiles, i.e. a handful of workgroups on 256 CUs (measured: the library
// picks a 9-way split at M = N = 64 and the call takes milliseconds).  So the pixels are cut into `S` chunks here, one strided-
// batched call computes the S partial products (S x tiles workgroups), and k_g_store_grad adds them up in fp64 on their way into
// the gradient vector.  Returns the number of partials left in `part` ([S][M][N]), 0 on failure.
constexpr int64_t kGradPartFloats = (int64_t)1 << 22;   // 16 MiB of partial products per filter
inline int gemm_atb_split(nf_trainer *t, hipStream_t st, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B,
                          int64_t ldb, float *part)
{
    const RocBlas *rb = rocblas_api();
    const float one = 1.0f, zero = 0.0f;
    if (!rb || !t->blas) return 0;
    rocblas_handle h = (rocblas_handle)t->blas;
    if (rb->set_stream(h, st) != rocblas_status_success) return 0;
    int64_t S = std::max<int64_t>(1, std::min<int64_t>(256, kGradPartFloats / (M * N)));
    S = std::min<int64_t>(S, std::max<int64_t>(1, K / 512));          // chunks of at least 512 pixels
    const int64_t Kc = K / S, rem = K - Kc * S;                         // S equal chunks, the remainder as one more partial
    // column-major view: C^T [N x M] = B^T-chunk [N x Kc] . A-chunk [Kc x M]  ->  op(B) = none (ld = ldb), op(A) = transpose (ld = lda)
    if (rb->sgemm_sb(h, rocblas_operation_none, rocblas_operation_transpose, (rocblas_int)N, (rocblas_int)M, (rocblas_int)Kc, &one, B,
                     (rocblas_int)ldb, (rocblas_stride)(Kc * ldb), A, (rocblas_int)lda, (rocblas_stride)(Kc * lda), &zero, part, (rocblas_int)N,
                     (rocblas_stride)(M * N), (rocblas_int)S) != rocblas_status_success)
        return 0;
    if (rem > 0) {
        if (rb->sgemm(h, rocblas_operation_none, rocblas_operation_transpose, (rocblas_int)N, (rocblas_int)M, (rocblas_int)rem, &one,
                      B + Kc * S * ldb, (rocblas_int)ldb, A + Kc * S * lda, (rocblas_int)lda, &zero, part + S * M * N,
                      (rocblas_int)N) != rocblas_status_success)
            return 0;
        return (int)S + 1;
    }
    return (int)S;
}

// ---- kernels over [pixel][w] tensors of run-time width ------------------------------------------------------------------

// Z18[p][tap*2 + c] = z0 of pixel p + tap (zero outside the patch): the im2col of l_1 (layers.py:586-613, 'SAME')
__global__ void k_g_gather18(Geo g, const float *__restrict__ z, float *__restrict__ Z18)
{
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            float *o = Z18 + p * 18;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int rr = r + tap / 3 - 1, cc = c + tap % 3 - 1;
                float2 v = make_float2(0.f, 0.f);
                if (rr >= 0 && rr < g.H && cc >= 0 && cc < g.W) v = *reinterpret_cast<const float2 *>(z + ((int64_t)b * g.HW + rr * g.W + cc) * 4);
                o[2 * tap] = v.x;
                o[2 * tap + 1] = v.y;
            }
        }
    }
}

// ---- column reductions over [pixel][w] tensors --------------------------------------------------------------------------
// "Flat" walk (widths that are a multiple of 4): the tensor as a stream of float4s, Q = w / 4 of them per pixel; the threads
// of a workgroup that take part (TBq = the largest multiple of Q that fits 256) step through it with a stride that is a multiple
// of Q, so every thread keeps its 4 channels and a wavefront reads 1 KiB of consecutive memory per load.  At the end the threads
// that share a channel group meet in LDS and the workgroup leaves ONE partial per channel in its slot — the layout k_bn_fin /
// k_bnb_fin / k_reduce add up.  Launched with exactly g.nslot workgroups (nobody's slot stays stale).
// Widths that are not a multiple of 4 take the plain per-channel loops below (k_g_*_slow).
struct FlatWalk {
    int Q, TBq;
    int64_t total, stride;
};
__device__ __forceinline__ FlatWalk flat_walk(int64_t npix, int w)
{
    FlatWalk f;
    f.Q = w >> 2;
    f.TBq = (256 / f.Q) * f.Q;
    f.total = npix * f.Q;
    f.stride = (int64_t)gridDim.x * f.TBq;
    return f;
}
// every thread of the workgroup calls this: v = this thread's sums for its 4 channels (threads beyond TBq pass zeros)
__device__ __forceinline__ void flat_store(const float (&v)[4], float *red, const FlatWalk &f, int w, Acc dst)
{
    const int t = threadIdx.x;
    __syncthreads();                       // the previous use of `red` is over
#pragma unroll
    for (int k = 0; k < 4; ++k) red[t * 4 + k] = v[k];
    __syncthreads();
    for (int c = t; c < w; c += 256) {
        float tot = 0.0f;
        for (int u = c >> 2; u < f.TBq; u += f.Q) tot += red[u * 4 + (c & 3)];
        (dst + c).p[blockIdx.x] = tot;
    }
}

// The GEMM's output h stays WITHOUT its bias in memory: every consumer adds it on the fly (g_xhat), which saves this kernel the
// write pass over the tensor.
// normalised activation of one value (layers.py:378-401 with the batch moments): the expression every kernel below shares, so
// that the ReLU mask the backward pass re-derives from h is bit for bit the forward's
__device__ __forceinline__ float g_xhat(float h, float b, float m, float rs) { return ((h + b) - m) * rs; }

// the slotted batch sums of h + bias (sum, sum of squares per channel)
__global__ __launch_bounds__(256) void k_g_bias_stats(Geo g, int w, const float *__restrict__ h, const float *__restrict__ bias, Acc stats)
{
    __shared__ float red[256 * 4];
    const FlatWalk f = flat_walk(g.npix, w);
    const int t = threadIdx.x, cg = t % f.Q;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < f.TBq) {
        const float b4[4] = {bias[4 * cg], bias[4 * cg + 1], bias[4 * cg + 2], bias[4 * cg + 3]};
        for (int64_t e = (int64_t)blockIdx.x * f.TBq + t; e < f.total; e += f.stride) {
            float4 v = reinterpret_cast<const float4 *>(h)[e];
            v.x += b4[0]; v.y += b4[1]; v.z += b4[2]; v.w += b4[3];
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] = fmaf(v.x, v.x, q[0]); q[1] = fmaf(v.y, v.y, q[1]); q[2] = fmaf(v.z, v.z, q[2]); q[3] = fmaf(v.w, v.w, q[3]);
        }
    }
    flat_store(s, red, f, w, stats);
    flat_store(q, red, f, w, stats + w);
}
// l_1 itself where the width allows the flat walk (a multiple of 4): h1 = Z18 . W1 has K = 18 — less arithmetic than the write
// of its own result — so the library GEMM, its read-back for the statistics and their launches collapse into ONE pass: every
// thread keeps the 18 x 4 filter entries of its four channels in registers, reads a pixel's 18 gathered inputs (the lanes that
// share a pixel share the cache line), writes its float4 of h1 (without the bias, as everywhere here) and accumulates the batch
// sums of h1 + bias.
__global__ __launch_bounds__(256) void k_g_c1_fwd(Geo g, int w, const float *__restrict__ Z18, const float *__restrict__ W1,
                                                  const float *__restrict__ bias, float *__restrict__ h, Acc stats)
{
    __shared__ float red[256 * 4];
    const FlatWalk f = flat_walk(g.npix, w);
    const int t = threadIdx.x, cg = t % f.Q;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < f.TBq) {
        float wk[18][4], b4[4];
#pragma unroll
        for (int k = 0; k < 18; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) wk[k][j] = W1[k * w + 4 * cg + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[j] = bias[4 * cg + j];
        for (int64_t e = (int64_t)blockIdx.x * f.TBq + t; e < f.total; e += f.stride) {
            const int64_t p = e / f.Q;
            const float2 *zr = reinterpret_cast<const float2 *>(Z18 + p * 18);   // rows of 72 bytes: 8-byte aligned
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k2 = 0; k2 < 9; ++k2) {
                const float2 zv = zr[k2];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = fmaf(zv.y, wk[2 * k2 + 1][j], fmaf(zv.x, wk[2 * k2][j], a[j]));
            }
            reinterpret_cast<float4 *>(h)[e] = make_float4(a[0], a[1], a[2], a[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = a[j] + b4[j];
                s[j] += v;
                q[j] = fmaf(v, v, q[j]);
            }
        }
    }
    flat_store(s, red, f, w, stats);
    flat_store(q, red, f, w, stats + w);
}
__global__ __launch_bounds__(256) void k_g_bias_stats_slow(Geo g, int w, const float *__restrict__ h, const float *__restrict__ bias, Acc stats)
{
    const int64_t per = (g.npix + gridDim.x - 1) / gridDim.x, p0 = per * blockIdx.x, p1 = p0 + per < g.npix ? p0 + per : g.npix;
    for (int j = threadIdx.x; j < w; j += 256) {
        const float bj = bias[j];
        double s = 0.0, q = 0.0;   // one thread walks the whole run: a float sum of `per` values would cost the mean its last bits
        for (int64_t p = p0; p < p1; ++p) {
            const float v = h[p * w + j] + bj;
            s += (double)v;
            q += (double)v * (double)v;
        }
        (stats + j).p[blockIdx.x] = (float)s;
        (stats + (w + j)).p[blockIdx.x] = (float)q;
    }
}

// a = relu((h + bias - mean) * rstd)   (layers.py:378-401 with the batch moments, then :478 / :489).  V = 4: four channels per
// thread (widths that are a multiple of 4), else one
template <int V>
__global__ void k_g_bn_relu(int64_t nv, int wv, const float *__restrict__ h, const float *__restrict__ bias, const float *__restrict__ bn,
                            float *__restrict__ a)
{
    const int w = V * wv;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nv; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e % wv) * V;
        if constexpr (V == 4) {
            const float4 v = reinterpret_cast<const float4 *>(h)[e];
            reinterpret_cast<float4 *>(a)[e] =
                make_float4(fmaxf(g_xhat(v.x, bias[j], bn[j], bn[w + j]), 0.f), fmaxf(g_xhat(v.y, bias[j + 1], bn[j + 1], bn[w + j + 1]), 0.f),
                            fmaxf(g_xhat(v.z, bias[j + 2], bn[j + 2], bn[w + j + 2]), 0.f), fmaxf(g_xhat(v.w, bias[j + 3], bn[j + 3], bn[w + j + 3]), 0.f));
        } else {
            a[e] = fmaxf(g_xhat(h[e], bias[j], bn[j], bn[w + j]), 0.f);
        }
    }
}

// W3r[i][tap*4 + k] = l_last/W[tap][i][k], i < w (the edge-indicator row i = w is handled by the gather kernels)
__global__ void k_g_pack_w3(int w, const float *__restrict__ W3, float *__restrict__ W3r)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= w * 36) return;
    const int i = e / 36, col = e - i * 36, tap = col >> 2, k = col & 3;
    W3r[e] = W3[(tap * (w + 1) + i) * 4 + k];
}

// u = sum of the 9 taps of P (+ the edge channel's weight where the tap falls on the padding ring, + b3); the affine transform
// of the second half (layers.py:355-375, 555-583, 651-674); keeps u for the backward pass
__global__ void k_g_c3_fwd(Geo g, int w, const float *__restrict__ zin, const float *__restrict__ P36, const float *__restrict__ Pw,
                           int off_w3, float *__restrict__ zout, Acc ldacc, float *__restrict__ u_out)
{
    const float *W3 = Pw + off_w3, *b3 = W3 + 36 * (w + 1), *logs = b3 + 4;
    const float sc = logs[4];
    const float e30 = expf(kLogscale * logs[0]), e31 = expf(kLogscale * logs[1]), e32 = expf(kLogscale * logs[2]),
                e33 = expf(kLogscale * logs[3]);
    float l = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            float u[4] = {b3[0], b3[1], b3[2], b3[3]};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int rr = r + tap / 3 - 1, cc = c + tap % 3 - 1;
                float4 v;
                if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {   // (the parameter block is only 4-byte aligned)
                    const float *e = W3 + (tap * (w + 1) + w) * 4;
                    v = make_float4(e[0], e[1], e[2], e[3]);
                } else {
                    v = *reinterpret_cast<const float4 *>(P36 + ((int64_t)b * g.HW + rr * g.W + cc) * 36 + tap * 4);
                }
                u[0] += v.x; u[1] += v.y; u[2] += v.z; u[3] += v.w;
            }
            reinterpret_cast<float4 *>(u_out)[p] = make_float4(u[0], u[1], u[2], u[3]);
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

// k_c3_bwd at run-time width (l_last's output was kept): through the affine transform, tanh, exp(3 logs); leaves d loss / d u in
// `gu`, d loss / d z1 in dz[2:4]; accumulates d l_last/b, d logs, d rescaling_scale
__global__ void k_g_c3_bwd(Geo g, int w, const float *__restrict__ zin, const float *__restrict__ P, int off_w3, float invB,
                           float *__restrict__ dz, float *__restrict__ gu, Acc G, const float *__restrict__ zlat,
                           const float *__restrict__ u_in)
{
    const float *W3 = P + off_w3, *b3 = W3 + 36 * (w + 1), *logs = b3 + 4;
    const float sc = logs[4];
    const int off_b3 = off_w3 + 36 * (w + 1);
    float e3[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e3[k] = expf(kLogscale * logs[k]);
    float g_s = 0.0f, g_logs[4] = {0.f, 0.f, 0.f, 0.f}, g_b3[4] = {0.f, 0.f, 0.f, 0.f};
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 uv = reinterpret_cast<const float4 *>(u_in)[p];
            const float u[4] = {uv.x, uv.y, uv.z, uv.w};
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
                const float th = tanhf(o[2 + k]), E = expf(sc * th);
                gz1[k] = gx1[k] * E;
                const float gls = gx1[k] * z1[k] * E - invB;   // loss = mean(-(sum ls + ...))
                g_s = fmaf(gls, th, g_s);
                go[k] = gx1[k];
                go[2 + k] = gls * sc * (1.0f - th * th);
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
    acc_add_n<9>(G + off_b3, tail, g.nslot);
}

// G36[q][tap*4 + k] = gu[q - tap][k], i.e. the gradient that reaches pixel q's activation through filter tap `tap`
// (q = p + tap  <=>  p = q - tap; zero where p falls outside the patch) — the operand of both d l_last/W = a2^T . G36 and
// g_a2 = G36 . W3r^T; and the edge channel: d W3[tap][w][k] = sum of gu[p][k] over the pixels whose tap falls on the padding ring
__global__ void k_g_gather36(Geo g, int w, const float *__restrict__ gu, float *__restrict__ G36, int off_w3, Acc G)
{
    float edge[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) edge[i] = 0.0f;
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
            const float4 own = reinterpret_cast<const float4 *>(gu)[p];
            float *o = G36 + p * 36;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int di = tap / 3 - 1, dj = tap % 3 - 1;
                const int pr = r - di, pc = c - dj;          // the output pixel whose tap `tap` reads this pixel
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pr >= 0 && pr < g.H && pc >= 0 && pc < g.W) v = reinterpret_cast<const float4 *>(gu)[(int64_t)b * g.HW + pr * g.W + pc];
                *reinterpret_cast<float4 *>(o + tap * 4) = v;
                const int rr = r + di, cc = c + dj;          // this pixel's own tap: on the ring?
                if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {
                    edge[tap * 4 + 0] += own.x; edge[tap * 4 + 1] += own.y; edge[tap * 4 + 2] += own.z; edge[tap * 4 + 3] += own.w;
                }
            }
        }
    }
    // 9 groups of 4 adjacent values at l_last/W[tap][w][0..3]
    for (int tap = 0; tap < 9; ++tap) {
        const float v4[4] = {edge[tap * 4], edge[tap * 4 + 1], edge[tap * 4 + 2], edge[tap * 4 + 3]};
        acc_add_n<4>(G + off_w3 + (tap * (w + 1) + w) * 4, v4, g.nslot);
    }
}

// The two batch sums BN's backward needs — sum gx, sum gx * xhat with gx = g_a where the activation is positive — from a
// read-only pass: the mask is re-derived from h (xhat > 0 <=> the forward's relu kept the value; same expression, same bits) and
// applied again by k_g_bn_bwd, so the masked gradient is never written.
__global__ __launch_bounds__(256) void k_g_mask_stats(Geo g, int w, const float *__restrict__ ga, const float *__restrict__ h,
                                                      const float *__restrict__ bias, const float *__restrict__ bn, Acc bstats)
{
    __shared__ float red[256 * 4];
    const FlatWalk f = flat_walk(g.npix, w);
    const int t = threadIdx.x, cg = t % f.Q;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < f.TBq) {
        float b[4], m[4], rs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = bias[4 * cg + k];
            m[k] = bn[4 * cg + k];
            rs[k] = bn[w + 4 * cg + k];
        }
        for (int64_t e = (int64_t)blockIdx.x * f.TBq + t; e < f.total; e += f.stride) {
            const float4 hv4 = reinterpret_cast<const float4 *>(h)[e], gv4 = reinterpret_cast<const float4 *>(ga)[e];
            const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w}, gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = g_xhat(hv[k], b[k], m[k], rs[k]);
                const float gx = xh > 0.f ? gv[k] : 0.f;
                s[k] += gx;
                q[k] = fmaf(gx, xh, q[k]);
            }
        }
    }
    flat_store(s, red, f, w, bstats);
    flat_store(q, red, f, w, bstats + w);
}
__global__ __launch_bounds__(256) void k_g_mask_stats_slow(Geo g, int w, const float *__restrict__ ga, const float *__restrict__ h,
                                                           const float *__restrict__ bias, const float *__restrict__ bn, Acc bstats)
{
    const int64_t per = (g.npix + gridDim.x - 1) / gridDim.x, p0 = per * blockIdx.x, p1 = p0 + per < g.npix ? p0 + per : g.npix;
    for (int j = threadIdx.x; j < w; j += 256) {
        const float b = bias[j], m = bn[j], rs = bn[w + j];
        double s = 0.0, q = 0.0;
        for (int64_t p = p0; p < p1; ++p) {
            const float xh = g_xhat(h[p * w + j], b, m, rs);
            const float gx = xh > 0.0f ? ga[p * w + j] : 0.0f;
            s += (double)gx;
            q += (double)gx * (double)xh;
        }
        (bstats + j).p[blockIdx.x] = (float)s;
        (bstats + (w + j)).p[blockIdx.x] = (float)q;
    }
}

// BN backward (in place, on the UNMASKED g_a): gx = g_a where xhat > 0, g_h = rstd * (gx - mean(gx) - xhat * mean(gx * xhat));
// and d bias = sum of g_h
__global__ __launch_bounds__(256) void k_g_bn_bwd(Geo g, int w, float *__restrict__ gx, const float *__restrict__ h, const float *__restrict__ bias,
                                                  const float *__restrict__ bn, const float *__restrict__ bb, Acc Gb)
{
    __shared__ float red[256 * 4];
    const FlatWalk f = flat_walk(g.npix, w);
    const int t = threadIdx.x, cg = t % f.Q;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < f.TBq) {
        float b[4], m[4], rs[4], ba[4], bq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = bias[4 * cg + k];
            m[k] = bn[4 * cg + k];
            rs[k] = bn[w + 4 * cg + k];
            ba[k] = bb[4 * cg + k];
            bq[k] = bb[w + 4 * cg + k];
        }
        for (int64_t e = (int64_t)blockIdx.x * f.TBq + t; e < f.total; e += f.stride) {
            const float4 hv4 = reinterpret_cast<const float4 *>(h)[e], gv4 = reinterpret_cast<const float4 *>(gx)[e];
            const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w}, gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = g_xhat(hv[k], b[k], m[k], rs[k]);
                o[k] = rs[k] * ((xh > 0.f ? gv[k] : 0.f) - ba[k] - xh * bq[k]);
                s[k] += o[k];
            }
            reinterpret_cast<float4 *>(gx)[e] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    flat_store(s, red, f, w, Gb);
}
__global__ __launch_bounds__(256) void k_g_bn_bwd_slow(Geo g, int w, float *__restrict__ gx, const float *__restrict__ h,
                                                       const float *__restrict__ bias, const float *__restrict__ bn,
                                                       const float *__restrict__ bb, Acc Gb)
{
    const int64_t per = (g.npix + gridDim.x - 1) / gridDim.x, p0 = per * blockIdx.x, p1 = p0 + per < g.npix ? p0 + per : g.npix;
    for (int j = threadIdx.x; j < w; j += 256) {
        const float b = bias[j], m = bn[j], rs = bn[w + j], ba = bb[j], bq = bb[w + j];
        double s = 0.0;
        for (int64_t p = p0; p < p1; ++p) {
            const float xh = g_xhat(h[p * w + j], b, m, rs);
            const float o = rs * ((xh > 0.0f ? gx[p * w + j] : 0.0f) - ba - xh * bq);
            gx[p * w + j] = o;
            s += (double)o;
        }
        (Gb + j).p[blockIdx.x] = (float)s;
    }
}

// transposed l_1 from Q[p][tap*2 + c] = sum_j g_h1[p][j] W1[tap][c][j]: d z0[q][c] += sum over taps of Q[q - tap][tap][c];
// MIX: the backward of the preceding Conv2d1x1 folded in (per pixel: dA += z_in^T d, d <- d A^T), as k_c1_dz
template <bool MIX>
__global__ void k_g_c1_dz(Geo g, const float *__restrict__ Q18, float *__restrict__ dz, const float *__restrict__ zmix_in,
                          const float *__restrict__ A, Acc dA)
{
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
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int qr = r - (tap / 3 - 1), qc = c - (tap % 3 - 1);
                if (qr < 0 || qr >= g.H || qc < 0 || qc >= g.W) continue;
                const float2 v = *reinterpret_cast<const float2 *>(Q18 + ((int64_t)b * g.HW + qr * g.W + qc) * 18 + tap * 2);
                a0 += v.x;
                a1 += v.y;
            }
            const float4 dv = reinterpret_cast<const float4 *>(dz)[p];
            if (MIX) {
                const float4 zv = reinterpret_cast<const float4 *>(zmix_in)[p];
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
                reinterpret_cast<float4 *>(dz)[p] = make_float4(dv.x + a0, dv.y + a1, dv.z, dv.w);
            }
        }
    }
    if (MIX) acc_add_n<16>(dA, acc, g.nslot);
}

// a filter gradient as the split GEMM left it — `nparts` partial products [nparts][n] — added up in fp64 into the gradient vector
// (after k_reduce has written the slotted values).
// mode 0: G[dst + e];  mode 1 (l_last/W): the partials are [w][36] = (i, tap*4 + k) -> G[dst + (tap*(w+1) + i)*4 + k]
constexpr int kStoreY = 16;   // threads that share one gradient entry's partial products
__global__ void __launch_bounds__(64 * kStoreY) k_g_store_grad(int n, int w, int mode, const float *__restrict__ part, int nparts,
                                                                double *__restrict__ G, int dst)
{
    __shared__ double acc[kStoreY][64];
    const int e = blockIdx.x * 64 + threadIdx.x;
    double s = 0.0;
    if (e < n)
        for (int q = threadIdx.y; q < nparts; q += kStoreY) s += (double)part[(size_t)q * n + e];
    acc[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y != 0 || e >= n) return;
    for (int y = 1; y < kStoreY; ++y) s += acc[y][threadIdx.x];
    if (mode == 0) {
        G[dst + e] = s;
    } else {
        const int i = e / 36, col = e - i * 36, tap = col >> 2, k = col & 3;
        G[dst + (tap * (w + 1) + i) * 4 + k] = s;
    }
}
inline void store_grad(hipStream_t st, int n, int w, int mode, const float *part, int nparts, double *G, int dst)
{
    hipLaunchKernelGGL(k_g_store_grad, dim3((n + 63) / 64), dim3(64, kStoreY), 0, st, n, w, mode, part, nparts, G, dst);
}

// ---- host side -------------------------------------------------------------------------------------------------------------

inline unsigned gemm_grid(const Geo &g) { return (unsigned)g.nslot; }   // one workgroup per slot: nobody's slot stays stale
// floats of partial products one filter gradient may leave (up to 257 partials of at most kGradPartFloats / 256 ... w*w floats)
inline size_t gemm_part_floats(int w) { return (size_t)kGradPartFloats + 2 * (size_t)w * w; }

bool coupling_forward_gemm(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float *zout, Acc ldacc, const float *zpre,
                           const float *A, hipStream_t st)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(g.npix), ns = gemm_grid(g);
    const int w = L.width, off_w1 = L.off, off_b1 = L.off + 18 * w, off_m1 = L.off + 19 * w, off_w2 = L.off + 21 * w,
              off_b2 = off_w2 + w * w, off_m2 = L.off + 22 * w + w * w, off_w3 = L.off + 24 * w + w * w;
    const double n = (double)g.npix * t->sync_world;
    const float *P = t->d_params;
    const int V = w % 4 == 0 ? 4 : 1;
    const int64_t nv = g.npix * (w / V);
    const unsigned ne = (unsigned)std::min<int64_t>((nv + 255) / 256, 256 * 32);
    auto bn_relu = [&](const float *h, const float *bias, const float *bn, float *a) {
        if (V == 4) hipLaunchKernelGGL(k_g_bn_relu<4>, dim3(ne), dim3(256), 0, st, nv, w / 4, h, bias, bn, a);
        else hipLaunchKernelGGL(k_g_bn_relu<1>, dim3(ne), dim3(256), 0, st, nv, w, h, bias, bn, a);
    };
    if (zpre) hipLaunchKernelGGL(k_mix_fwd, dim3(nb), dim3(TB), 0, st, g, zpre, A, const_cast<float *>(zin));
    hipLaunchKernelGGL(k_g_gather18, dim3(nb), dim3(TB), 0, st, g, zin, t->gz18);
    bool ok = true;
    if (V == 4 && t->gemm_c1_fused) {
        hipLaunchKernelGGL(k_g_c1_fwd, dim3(ns), dim3(256), 0, st, g, w, (const float *)t->gz18, P + off_w1, P + off_b1, c.h1, t->acc(c.d_st1));
    } else {
        ok = gemm_rm(t, st, false, false, g.npix, w, 18, t->gz18, 18, P + off_w1, w, c.h1, w);
        if (V == 4) hipLaunchKernelGGL(k_g_bias_stats, dim3(ns), dim3(256), 0, st, g, w, (const float *)c.h1, P + off_b1, t->acc(c.d_st1));
        else hipLaunchKernelGGL(k_g_bias_stats_slow, dim3(ns), dim3(256), 0, st, g, w, (const float *)c.h1, P + off_b1, t->acc(c.d_st1));
    }
    sync_slots(t, t->acc(c.d_st1), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_bn_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_st1), w, g.nslot, n, t->d_params, off_m1, off_m1 + w, t->d_flt + c.f_bn1);
    bn_relu(c.h1, P + off_b1, t->d_flt + c.f_bn1, c.a1);
    ok = ok && gemm_rm(t, st, false, false, g.npix, w, w, c.a1, w, P + off_w2, w, c.h2, w);
    if (V == 4) hipLaunchKernelGGL(k_g_bias_stats, dim3(ns), dim3(256), 0, st, g, w, (const float *)c.h2, P + off_b2, t->acc(c.d_st2));
    else hipLaunchKernelGGL(k_g_bias_stats_slow, dim3(ns), dim3(256), 0, st, g, w, (const float *)c.h2, P + off_b2, t->acc(c.d_st2));
    sync_slots(t, t->acc(c.d_st2), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_bn_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_st2), w, g.nslot, n, t->d_params, off_m2, off_m2 + w, t->d_flt + c.f_bn2);
    bn_relu(c.h2, P + off_b2, t->d_flt + c.f_bn2, c.a2);
    hipLaunchKernelGGL(k_g_pack_w3, dim3((w * 36 + 255) / 256), dim3(256), 0, st, w, P + off_w3, t->gw3r);
    ok = ok && gemm_rm(t, st, false, false, g.npix, 36, w, c.a2, w, t->gw3r, 36, t->gp36, 36);
    hipLaunchKernelGGL(k_g_c3_fwd, dim3(nb), dim3(TB), 0, st, g, w, zin, (const float *)t->gp36, P, off_w3, zout, ldacc, c.u);
    return ok;
}

bool coupling_backward_gemm(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float invB, const float *zmix_in,
                            const float *A, Acc dA, hipStream_t st, const float *zlat)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(g.npix), ns = gemm_grid(g);
    const int w = L.width, off_w1 = L.off, off_b1 = L.off + 18 * w, off_w2 = L.off + 21 * w, off_b2 = off_w2 + w * w,
              off_w3 = L.off + 24 * w + w * w;
    const double n = (double)g.npix * t->sync_world;
    const float *P = t->d_params, *bn1 = t->d_flt + c.f_bn1, *bn2 = t->d_flt + c.f_bn2;
    const Acc G = t->acc(0);
    float *t1 = t->t1[0], *t2 = t->t2[0], *gu = t->gu[0];
    // this coupling's filter gradients: the partial products of the split GEMMs, summed when the step's gradients are assembled
    float *dW1 = t->gdw + (size_t)(3 * L.aux) * gemm_part_floats(w), *dW2 = dW1 + gemm_part_floats(w), *dW3r = dW2 + gemm_part_floats(w);
    int *np = t->gnp + 3 * L.aux;
    hipLaunchKernelGGL(k_g_c3_bwd, dim3(nb), dim3(TB), 0, st, g, w, zin, P, off_w3, invB, t->dz, gu, G, zlat, (const float *)c.u);
    hipLaunchKernelGGL(k_g_gather36, dim3(nb), dim3(TB), 0, st, g, w, (const float *)gu, t->gp36, off_w3, G);
    hipLaunchKernelGGL(k_g_pack_w3, dim3((w * 36 + 255) / 256), dim3(256), 0, st, w, P + off_w3, t->gw3r);
    bool ok = (np[2] = gemm_atb_split(t, st, w, 36, g.npix, c.a2, w, t->gp36, 36, dW3r)) > 0;         // d l_last/W = a2^T . G36
    ok = ok && gemm_rm(t, st, false, true, g.npix, w, 36, t->gp36, 36, t->gw3r, 36, t1, w);          // g_a2 = G36 . W3r^T
    const bool flat = w % 4 == 0;
    hipLaunchKernelGGL(flat ? k_g_mask_stats : k_g_mask_stats_slow, dim3(ns), dim3(256), 0, st, g, w, (const float *)t1, (const float *)c.h2,
                       P + off_b2, bn2, t->acc(c.d_bs2));
    sync_slots(t, t->acc(c.d_bs2), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_bnb_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_bs2), w, g.nslot, n, t->d_flt + c.f_bb2);
    hipLaunchKernelGGL(flat ? k_g_bn_bwd : k_g_bn_bwd_slow, dim3(ns), dim3(256), 0, st, g, w, t1, (const float *)c.h2, P + off_b2, bn2,
                       (const float *)(t->d_flt + c.f_bb2), G + off_b2);
    ok = ok && (np[1] = gemm_atb_split(t, st, w, w, g.npix, c.a1, w, t1, w, dW2)) > 0;                 // d l_2/W = a1^T . g_h2
    ok = ok && gemm_rm(t, st, false, true, g.npix, w, w, t1, w, P + off_w2, w, t2, w);               // g_a1 = g_h2 . W2^T
    hipLaunchKernelGGL(flat ? k_g_mask_stats : k_g_mask_stats_slow, dim3(ns), dim3(256), 0, st, g, w, (const float *)t2, (const float *)c.h1,
                       P + off_b1, bn1, t->acc(c.d_bs1));
    sync_slots(t, t->acc(c.d_bs1), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_bnb_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_bs1), w, g.nslot, n, t->d_flt + c.f_bb1);
    hipLaunchKernelGGL(flat ? k_g_bn_bwd : k_g_bn_bwd_slow, dim3(ns), dim3(256), 0, st, g, w, t2, (const float *)c.h1, P + off_b1, bn1,
                       (const float *)(t->d_flt + c.f_bb1), G + off_b1);
    hipLaunchKernelGGL(k_g_gather18, dim3(nb), dim3(TB), 0, st, g, zin, t->gz18);
    ok = ok && (np[0] = gemm_atb_split(t, st, 18, w, g.npix, t->gz18, 18, t2, w, dW1)) > 0;            // d l_1/W = Z18^T . g_h1
    ok = ok && gemm_rm(t, st, false, true, g.npix, 18, w, t2, w, P + off_w1, w, t->gq18, 18);        // Q = g_h1 . W1^T
    if (zmix_in)
        hipLaunchKernelGGL(k_g_c1_dz<true>, dim3(nb), dim3(TB), 0, st, g, (const float *)t->gq18, t->dz, zmix_in, A, dA);
    else
        hipLaunchKernelGGL(k_g_c1_dz<false>, dim3(nb), dim3(TB), 0, st, g, (const float *)t->gq18, t->dz, (const float *)nullptr,
                           (const float *)nullptr, dA);
    return ok;
}

}  // namespace
