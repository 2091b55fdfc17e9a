// Training step at coupling widths without stage kernels of their own (part of nf_train.hip): 33 .. 512 and 1 .. 31 except 4 / 8 / 16.
//
// sidd/ArgParser.py:43 defaults --width to Glow's 512 and train_noise_flow.py:50-77,187-198 trains at whatever width is set.
// The layer kernels of nf_train.hip keep (w+1)*4 accumulators per thread and the matrix-core stages of nf_train_wide.h hold one
// 32-channel tile per operand: neither exists beyond width 32.  At these widths a coupling IS GEMM-shaped — l_2 is
// [pixels x w] . [w x w], 90 % of the arithmetic at 512 — so the dense products of the step are the hand-written fp32
// matrix-core GEMMs of nf_train_mm.h (k_mm_pix: pixels on M; k_mm_kpix: pixels on K), with batch normalisation + ReLU fused into
// their operand staging and the batch sums into their epilogues; what is not a dense product is a kernel over [pixel][w]
// tensors of run-time width below:
//
//   forward   the coupling's weights in the GEMMs' packed layouts (one launch)     mm::k_mm_pack_all
//             Z18 = the 3x3 x 2-channel windows of the pass-through half            k_g_gather18
//             h1 = Z18 . W1 (K = 18) + batch sums of h1 + b1                        k_mm_pix <EPI 1> (NF_TRAIN_GEMM_C1=1: k_g_c1_fwd) -> k_bn_fin
//             h2 = relu(bn1(h1 + b1)) . W2 + batch sums of h2 + b2                  k_mm_pix <APRO 1, EPI 1>                 -> k_bn_fin
//             P = relu(bn2(h2 + b2)) . W3r  (W3r[i][tap*4+k] = l_last/W[tap][i][k]) k_mm_pix <APRO 1>
//             u = gather of the 9 taps of P + edge channel + b3; affine transform   k_g_c3_fwd
//   backward  affine / tanh / exp(3 logs) backward -> gu, d b3, d logs, d scale     k_g_c3_bwd
//             G36[p][tap*4+k] = gu[p - tap][k]; d edge-channel weights              k_g_gather36
//             d l_last/W = relu(bn2(h2 + b2))^T . G36  (w >= 256: and mask2^T . G36)   k_mm_kpix <APRO 1 / 3> (one pass over h2)
//             the two batch sums of BN2's backward: those products . W3r            k_g_bnb_from_parts                       -> k_bnb_fin
//                                    (w < 256: a sums-only pass over G36 . W3r^T)   k_mm_pix <EPI 4>
//             g_h2 = BN2 backward of the masked g_a2 = G36 . W3r^T (K = 36); d b2   k_mm_pix <EPI 3>
//             d l_2/W = relu(bn1(h1 + b1))^T . g_h2                                 k_mm_kpix <APRO 1>
//             g_a1 = g_h2 . W2^T + the two batch sums of BN1's backward             k_mm_pix <EPI 2>                         -> k_bnb_fin
//             d l_1/W = Z18^T . g_h1, g_h1 = BN1 backward of the masked g_a1; d b1  k_mm_kpix <BPRO 2>
//             Q = g_h1 . W1^T (18 columns), g_h1 formed again                       k_mm_pix <APRO 2>
//             d z0 += gather of Q; the folded Conv2d1x1 backward                    k_g_c1_dz
//
// The pre-BN activations h1 / h2 stay WITHOUT their bias in memory and are the only [pixel][w] tensors a coupling keeps: the
// normalised activations are re-formed from them wherever they are an operand (same expression, same bits: mm::xhat).
// Gradients of the three filters come out of k_mm_kpix as partial products per pixel chunk and go into the fp64 gradient vector
// (k_g_store_grad); every other reduction of the step keeps the trainer's slotted partial sums, so the batch statistics can be
// synchronised across ranks exactly as at the other widths.  One stream, no side work: a wide step is milliseconds of GEMMs.
//
// Replaces (reference, /root/reference): train_noise_flow.py:64-66 with borealisflows/layers.py:452-498 at hps.width > 32.
#pragma once
#include "nf_train_mm.h"

namespace {

static_assert(mm::kSlotStride == NSLOT, "nf_train_mm.h writes the trainer's slotted accumulators");

// the widths with kernels of their own (nf_train.hip, nf_train_tiled.h, nf_train_wide.h); every other width 1 .. 512 runs here
// (NF_TRAIN_GEMM=1 in the environment sends those four here as well: an A/B switch for tests and profiles)
inline bool gemm_width(int w)
{
    static const bool all = [] { const char *e = getenv("NF_TRAIN_GEMM"); return e && atoi(e) != 0; }();
    return all || (w != 4 && w != 8 && w != 16 && w != 32);
}

// ---- kernels over [pixel][w] tensors of run-time width ------------------------------------------------------------------

// Z18[p][tap*2 + c] = z0 of pixel p + tap (zero outside the patch): the im2col of l_1 (layers.py:586-613, 'SAME').  Rows of
// kZ18 = 20 floats (16-byte aligned; the two spare columns are zero, so a GEMM may run K = 20)
constexpr int kZ18 = 20;
// one thread per 16-byte piece (2 taps) of a row: a wavefront writes 1 KiB of consecutive memory (one thread per ROW wrote 5 pieces
// 80 bytes apart: 0.8 TB/s, 113 us per coupling at 1 024 patches)
__global__ void k_g_gather18(Geo g, const float *__restrict__ z, float *__restrict__ Z18)
{
    const int64_t total = g.npix * 5;
    for (int64_t it = (int64_t)blockIdx.x * TB + threadIdx.x; it < total; it += (int64_t)gridDim.x * TB) {
        const int64_t p = it / 5;
        const int q = (int)(it - p * 5);
        const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
        float2 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tap = 2 * q + h;
            const int rr = r + tap / 3 - 1, cc = c + tap % 3 - 1;
            v[h] = make_float2(0.f, 0.f);
            if (tap < 9 && rr >= 0 && rr < g.H && cc >= 0 && cc < g.W) v[h] = *reinterpret_cast<const float2 *>(z + ((int64_t)b * g.HW + rr * g.W + cc) * 4);
        }
        reinterpret_cast<float4 *>(Z18)[it] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
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

// The pre-BN activations h stay WITHOUT their bias in memory: every consumer adds it on the fly (mm::xhat).

// l_1 itself where the width allows the flat walk (a multiple of 4): h1 = Z18 . W1 has K = 18 — less arithmetic than the write
// of its own result — so the product, the read-back for the statistics and their launches collapse into ONE pass: every
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
            const float2 *zr = reinterpret_cast<const float2 *>(Z18 + p * kZ18);
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
// 288 threads = 32 pixels x 9 taps: a thread keeps its tap, so a wavefront writes 1 KiB of consecutive memory and the ring sums of
// a tap stay in 4 registers (one thread per PIXEL wrote 9 pieces 144 bytes apart: 1.2 TB/s)
constexpr int kG36T = 288;
__global__ __launch_bounds__(kG36T) void k_g_gather36(Geo g, int w, const float *__restrict__ gu, float *__restrict__ G36, int off_w3, Acc G)
{
    __shared__ float red[kG36T][4];
    const int t = threadIdx.x, tap = t % 9, lp = t / 9;
    const int di = tap / 3 - 1, dj = tap % 3 - 1;
    const float4 *const gu4 = reinterpret_cast<const float4 *>(gu);
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ngroups = (g.npix + 31) / 32;
    for (int64_t gi = blockIdx.x; gi < ngroups; gi += gridDim.x) {
        const int64_t p = gi * 32 + lp;
        if (p >= g.npix) continue;
        const int b = (int)(p / g.HW), rem = (int)(p - (int64_t)b * g.HW), r = rem / g.W, c = rem - r * g.W;
        const int pr = r - di, pc = c - dj;          // the output pixel whose tap `tap` reads this pixel
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr >= 0 && pr < g.H && pc >= 0 && pc < g.W) v = gu4[(int64_t)b * g.HW + pr * g.W + pc];
        reinterpret_cast<float4 *>(G36)[p * 9 + tap] = v;
        const int rr = r + di, cc = c + dj;          // this pixel's own tap: on the ring?
        if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {
            const float4 own = gu4[p];
            e.x += own.x; e.y += own.y; e.z += own.z; e.w += own.w;
        }
    }
    red[t][0] = e.x; red[t][1] = e.y; red[t][2] = e.z; red[t][3] = e.w;
    __syncthreads();
    if (t < 36) {   // 9 groups of 4 adjacent values at l_last/W[tap][w][0..3]: this workgroup's partial into its slot
        const int tp = t >> 2, k = t & 3;
        float sum = 0.0f;
        for (int i = 0; i < 32; ++i) sum += red[i * 9 + tp][k];
        float *d = (G + off_w3 + (tp * (w + 1) + w) * 4 + k).p;
        d[blockIdx.x] = sum;
        for (int q = blockIdx.x + gridDim.x; q < g.nslot; q += gridDim.x) d[q] = 0.0f;
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

// Filter gradients as the split GEMMs left them — `nparts` partial products [nparts][n] each — added up in fp64 into the gradient
// vector, every filter of the step in ONE launch (3 per coupling: at width 64 the 24 separate launches were 6 % of the step).
// mode 0: G[dst + e];  mode 1 (l_last/W): the partials are [w][36] = (i, tap*4 + k) -> G[dst + (tap*(w+1) + i)*4 + k]
constexpr int kStoreY = 16;   // threads that share one gradient entry's partial products
struct StoreJob {
    const float *part;
    int n, w, mode, nparts, dst, pstride;   // pstride: floats between two partial products
};
struct StoreJobs {
    static constexpr int kMax = 24;
    int count = 0;
    int first[kMax + 1] = {0};   // first workgroup of job i (64 entries per workgroup)
    StoreJob j[kMax];
};
// A workgroup = 64 x kStoreY threads: x owns kStoreV = 4 consecutive entries (one 16-byte load per partial product; jobs whose
// partial products are not 16-byte aligned fall back to one entry per thread of a 4 x shorter row), y strides over the partials.
constexpr int kStoreV = 4;
__device__ __forceinline__ void store_entry(const StoreJob &q, double *__restrict__ G, int e, double s)
{
    if (q.mode == 0) {
        G[q.dst + e] = s;
    } else {
        const int i = e / 36, col = e - i * 36, tap = col >> 2, k = col & 3;
        G[q.dst + (tap * (q.w + 1) + i) * 4 + k] = s;
    }
}
__global__ void __launch_bounds__(64 * kStoreY) k_g_store_grads(const StoreJobs J, double *__restrict__ G)
{
    __shared__ double acc[kStoreY][64][kStoreV];
    int job = 0;
    while (job + 1 < J.count && (int)blockIdx.x >= J.first[job + 1]) ++job;   // workgroup-uniform
    const StoreJob q = J.j[job];
    const int e0 = (((int)blockIdx.x - J.first[job]) * 64 + threadIdx.x) * kStoreV;
    const bool vec = (q.n % kStoreV == 0) && (q.pstride % kStoreV == 0) && ((reinterpret_cast<uintptr_t>(q.part) & 15) == 0);   // workgroup-uniform
    double s[kStoreV] = {0.0, 0.0, 0.0, 0.0};
    if (e0 < q.n) {
        if (vec) {
            for (int k = threadIdx.y; k < q.nparts; k += kStoreY) {
                const float4 v = *reinterpret_cast<const float4 *>(q.part + (size_t)k * q.pstride + e0);
                s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
            }
        } else {
            for (int k = threadIdx.y; k < q.nparts; k += kStoreY)
#pragma unroll
                for (int i = 0; i < kStoreV; ++i)
                    if (e0 + i < q.n) s[i] += (double)q.part[(size_t)k * q.pstride + e0 + i];
        }
    }
#pragma unroll
    for (int i = 0; i < kStoreV; ++i) acc[threadIdx.y][threadIdx.x][i] = s[i];
    __syncthreads();
    if (threadIdx.y != 0 || e0 >= q.n) return;
#pragma unroll
    for (int i = 0; i < kStoreV; ++i) {
        for (int y = 1; y < kStoreY; ++y) s[i] += acc[y][threadIdx.x][i];
        if (e0 + i < q.n) store_entry(q, G, e0 + i, s[i]);
    }
}
inline void store_grads_flush(hipStream_t st, StoreJobs &J, double *G)
{
    if (J.count) hipLaunchKernelGGL(k_g_store_grads, dim3((unsigned)J.first[J.count]), dim3(64, kStoreY), 0, st, J, G);
    J = StoreJobs{};
}
inline void store_grad(hipStream_t st, StoreJobs &J, int n, int w, int mode, const float *part, int nparts, double *G, int dst, int pstride = 0)
{
    if (J.count == StoreJobs::kMax) store_grads_flush(st, J, G);
    J.j[J.count] = StoreJob{part, n, w, mode, nparts, dst, pstride ? pstride : n};
    J.first[J.count + 1] = J.first[J.count] + (n + 64 * kStoreV - 1) / (64 * kStoreV);
    ++J.count;
}

// The slotted sums of runs of values (everything but the filters the GEMMs deliver whole) added up in fp64, all runs in ONE launch
struct ReduceRuns {
    static constexpr int kMax = 32;
    int count = 0;
    int first[kMax + 1] = {0};   // first workgroup (= value) of run i
    const float *part[kMax];     // the run's first row of slots
    int dst[kMax];
};
__global__ void __launch_bounds__(64) k_g_reduce_runs(const ReduceRuns R, int nslot, double *__restrict__ G)
{
    int run = 0;
    while (run + 1 < R.count && (int)blockIdx.x >= R.first[run + 1]) ++run;
    const int i = (int)blockIdx.x - R.first[run];
    const float *row = R.part[run] + (size_t)i * NSLOT;
    double s = 0.0;
    for (int k = threadIdx.x; k < nslot; k += 64) s += (double)row[k];
    s = wsum(s);
    if (threadIdx.x == 0) G[R.dst[run] + i] = s;
}
inline void reduce_runs_flush(hipStream_t st, ReduceRuns &R, int nslot, double *G)
{
    if (R.count) hipLaunchKernelGGL(k_g_reduce_runs, dim3((unsigned)R.first[R.count]), dim3(64), 0, st, R, nslot, G);
    R = ReduceRuns{};
}
inline void reduce_run(hipStream_t st, ReduceRuns &R, int nslot, double *G, const float *part, int a, int b)
{
    if (b <= a) return;
    if (R.count == ReduceRuns::kMax) reduce_runs_flush(st, R, nslot, G);
    R.part[R.count] = part;
    R.dst[R.count] = a;
    R.first[R.count + 1] = R.first[R.count] + (b - a);
    ++R.count;
}

// The two batch sums of BN2's backward from the two products of k_mm_kpix <APRO 3> (part[s][0] = relu(xhat)^T . G36 = d l_last/W as
// [w][36], part[s][1] = mask^T . G36), contracted with the filter W3r [w][36]:
//   sum_p gx xhat = sum_col W3r[i][col] (relu(xhat)^T G36)[i][col]      (gx = mask g_a2, g_a2 = G36 . W3r^T, relu(xhat) = mask xhat)
//   sum_p gx      = sum_col W3r[i][col] (mask^T G36)[i][col]
// in fp64 over the chunks, left as slot 0 (hi) + slot 1 (lo) of the slotted sums k_bnb_fin / sync_slots read.  One wavefront per channel.
__global__ __launch_bounds__(64) void k_g_bnb_from_parts(int w, int S, const float *__restrict__ part, const float *__restrict__ W3r, Acc bstats,
                                                          int nslot)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    double sx = 0.0, sg = 0.0;
    for (int e = lane; e < S * 36; e += 64) {
        const int sidx = e / 36, col = e - sidx * 36;
        const float wv = W3r[i * 36 + col];
        const float *ps = part + ((size_t)sidx * 2 * w + i) * 36 + col;
        sx += (double)wv * (double)ps[0];
        sg += (double)wv * (double)ps[(size_t)w * 36];
    }
    sx = wsum(sx);
    sg = wsum(sg);
    float *pg = (bstats + i).p, *px = (bstats + (w + i)).p;
    const float gh = (float)sg, xh = (float)sx;
    for (int k = lane; k < nslot; k += 64) {
        pg[k] = k == 0 ? gh : k == 1 ? (float)(sg - (double)gh) : 0.0f;
        px[k] = k == 0 ? xh : k == 1 ? (float)(sx - (double)xh) : 0.0f;
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------

inline unsigned gemm_grid(const Geo &g) { return (unsigned)g.nslot; }   // one workgroup per slot: nobody's slot stays stale
// floats of partial products one filter gradient may leave (up to 256 partials of M x N — 512 at widths <= 128 —, at least one)
// — sized from what mm_kpix_launch can ask for at this width, not a flat 64 MiB: S <= 512 partials (x 2 for the dual product) of at most
// max(w x w, 36 x w) floats (17 couplings at width 12 reserved 3.2 GiB and used a few MiB); the launches are handed this capacity
// (KpixArgs::part_cap) and cut their chunk count to it
inline size_t gemm_part_cap(int w)
{
    const size_t mn = std::max((size_t)w * w, (size_t)64 * std::max(w, 32));
    return std::min<size_t>((size_t)mm::kGradPartFloats, 1024 * mn);
}
inline size_t gemm_part_floats(int w) { return gemm_part_cap(w) + 2 * (size_t)w * w; }
// floats of packed weights one coupling's GEMMs read (nf_train_mm.h: pack_layout)
inline size_t gemm_pack_floats(int w) { return (mm::pack_layout(w).total + 3) & ~(size_t)3; }

// batch moments of one BN of an evaluation under batch statistics (nf_bs_wide_run): as k_bn_fin, but the running statistics stay
// where they are (nf_*_batchstats reports the moments; applying the EMA is the caller's business) and mean / variance go to `mom`
__global__ __launch_bounds__(64) void k_bn_fin_eval(Acc stats, int W, int nslot, double n, float *__restrict__ bn_out, float *__restrict__ mom)
{
    const int j = blockIdx.x;
    double sm, sq;
    acc_total2(stats + j, stats + W + j, nslot, sm, sq);
    const double m = sm / n;
    double v = sq / n - m * m;
    if (v < 0.0) v = 0.0;
    if (threadIdx.x == 0) {
        bn_out[j] = (float)m;
        bn_out[W + j] = (float)(1.0 / sqrt(v + (double)kBnEps));
        mom[j] = (float)m;
        mom[W + j] = (float)v;
    }
}

// The coupling CNN up to the 36 columns of l_last's transposed evaluation (t->gp36), batch statistics formed on the way.
// mom = nullptr: a training step (running statistics move); else the [4][w] moments of an evaluation call are left there.
// the packed weights of every coupling of a training step (they do not change between its forward and backward pass)
void gemm_pack_step(nf_trainer *t, hipStream_t st)
{
    const int w = t->width ? t->width : 4;
    const mm::PackAll pl = mm::pack_layout(w);
    mm::PackCpl pc{};
    int n = 0, first = 0;
    auto flush = [&] {
        if (n)
            hipLaunchKernelGGL(mm::k_mm_pack_all, dim3((unsigned)((pl.total + 255) / 256), (unsigned)n), dim3(256), 0, st, w, pl,
                               (const float *)t->d_params, pc, t->gpack + (size_t)first * gemm_pack_floats(w), gemm_pack_floats(w));
        first += n;
        n = 0;
    };
    for (int l = 0; l < t->tl.n; ++l) {
        const TLayer &L = t->tl.l[l];
        if (L.type != NF_LAYER_COUPLING) continue;
        if (n == mm::PackCpl::kMax) flush();
        pc.off[n++] = L.off;   // couplings in layer order = aux order
    }
    flush();
}

bool coupling_cnn_gemm(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, hipStream_t st, float *mom)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(t, g.npix), ns = gemm_grid(g);
    const int w = L.width, off_w1 = L.off, off_b1 = L.off + 18 * w, off_m1 = L.off + 19 * w, off_w2 = L.off + 21 * w,
              off_b2 = off_w2 + w * w, off_m2 = L.off + 22 * w + w * w;
    const double n = (double)g.npix * t->sync_world;
    const float *P = t->d_params;
    const bool v4 = w % 4 == 0;
    const mm::PackAll pl = mm::pack_layout(w);
    // this coupling's packed weights: written here, read again by the backward pass (an evaluator keeps one coupling's at a time)
    float *pk = t->gpack + (t->eval_only ? 0 : (size_t)L.aux * gemm_pack_floats(w));
    if (t->eval_only) {   // (a trainer packs every coupling's weights in one launch at the start of the step: gemm_pack_step)
        mm::PackCpl pc{};
        pc.off[0] = off_w1;
        hipLaunchKernelGGL(mm::k_mm_pack_all, dim3((unsigned)((pl.total + 255) / 256), 1), dim3(256), 0, st, w, pl, P, pc, pk, (size_t)0);
    }
    float *const z18 = t->gz18 + (size_t)L.aux * t->gz18_stride;   // kept for the backward pass (a trainer; an evaluator: stride 0)
    hipLaunchKernelGGL(k_g_gather18, dim3(nb), dim3(TB), 0, st, g, zin, z18);
    const mm::Ctx cx{t->n_cu, t->device};
    bool ok = true;
    mm::PixArgs a{};
    a.P = g.npix;
    a.nslot = g.nslot;
    auto bn_fin = [&](int d_st, int off_m, int f_bn, float *mo) {
        sync_slots(t, t->acc(d_st), 2 * w, g.nslot, st);
        if (mom) hipLaunchKernelGGL(k_bn_fin_eval, dim3(w), dim3(64), 0, st, t->acc(d_st), w, g.nslot, n, t->d_flt + f_bn, mo);
        else hipLaunchKernelGGL(k_bn_fin, dim3(w), dim3(64), 0, st, t->acc(d_st), w, g.nslot, n, t->d_params, off_m, off_m + w, t->d_flt + f_bn);
    };
    // ---- l_1: h1 = Z18 . W1 and the batch sums of h1 + b1 ----
    if (v4 && t->gemm_c1_fused) {
        hipLaunchKernelGGL(k_g_c1_fwd, dim3(ns), dim3(256), 0, st, g, w, (const float *)z18, P + off_w1, P + off_b1, c.h1, t->acc(c.d_st1));
    } else {
        a.N = w; a.K = kZ18;                                   // the two spare columns of Z18 and of the packed W1^T are zero
        a.A = z18; a.lda = kZ18;
        a.Bt = pk + pl.o_w1t; a.ldb = 20;
        a.C = c.h1; a.ldc = w;
        a.ebias = P + off_b1; a.stats = t->acc(c.d_st1).p;
        ok = mm::mm_pix<0, 1, 4>(cx, st, a) && ok;
    }
    bn_fin(c.d_st1, off_m1, c.f_bn1, mom);
    // ---- l_2: h2 = relu(bn1(h1 + b1)) . W2 and the batch sums of h2 + b2 ----
    a.N = w; a.K = w;
    a.A = c.h1; a.lda = w;
    a.Bt = pk + pl.o_w2t; a.ldb = pl.w4;
    a.C = c.h2; a.ldc = w;
    a.abias = P + off_b1; a.abn = t->d_flt + c.f_bn1;
    a.ebias = P + off_b2; a.stats = t->acc(c.d_st2).p;
    ok = (v4 ? mm::mm_pix<1, 1, 4>(cx, st, a) : mm::mm_pix<1, 1, 1>(cx, st, a)) && ok;
    bn_fin(c.d_st2, off_m2, c.f_bn2, mom ? mom + 2 * w : nullptr);
    // ---- l_last, transposed: P36 = relu(bn2(h2 + b2)) . W3r ----
    a.N = 36; a.K = w;
    a.A = c.h2; a.lda = w;
    a.Bt = pk + pl.o_w3a; a.ldb = pl.w4;
    a.C = t->gp36; a.ldc = 36;
    a.abias = P + off_b2; a.abn = t->d_flt + c.f_bn2;
    a.ebias = nullptr; a.stats = nullptr;
    ok = (v4 ? mm::mm_pix<1, 0, 4>(cx, st, a) : mm::mm_pix<1, 0, 1>(cx, st, a)) && ok;
    return ok;
}

bool coupling_forward_gemm(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float *zout, Acc ldacc, const float *zpre,
                           const float *A, hipStream_t st)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(t, g.npix);
    const int w = L.width, off_w3 = L.off + 24 * w + w * w;
    if (zpre) hipLaunchKernelGGL(k_mix_fwd, dim3(nb), dim3(TB), 0, st, g, zpre, A, const_cast<float *>(zin));
    const bool ok = coupling_cnn_gemm(t, g, L, zin, st, nullptr);
    hipLaunchKernelGGL(k_g_c3_fwd, dim3(nb), dim3(TB), 0, st, g, w, zin, (const float *)t->gp36, t->d_params, off_w3, zout, ldacc, c.u);
    return ok;
}

bool coupling_backward_gemm(nf_trainer *t, const Geo &g, const TLayer &L, const float *zin, float invB, const float *zmix_in,
                            const float *A, Acc dA, hipStream_t st, const float *zlat)
{
    const Cpl &c = t->cpl[L.aux];
    const unsigned nb = blocks_for(t, g.npix);
    const int w = L.width, off_w1 = L.off, off_b1 = L.off + 18 * w, off_w2 = L.off + 21 * w, off_b2 = off_w2 + w * w,
              off_w3 = L.off + 24 * w + w * w;
    (void)off_w1; (void)off_w2;
    const double n = (double)g.npix * t->sync_world;
    const float *P = t->d_params, *bn1 = t->d_flt + c.f_bn1, *bn2 = t->d_flt + c.f_bn2, *bb1 = t->d_flt + c.f_bb1, *bb2 = t->d_flt + c.f_bb2;
    // in-kernel offsets from G are all behind this coupling's l_2/W (the rows before it are shifted by the holes: nf_trainer::acc)
    const Acc G = Acc{t->acc(off_w3).p - (size_t)off_w3 * NSLOT};
    float *t1 = t->t1[0], *t2 = t->t2[0], *gu = t->gu[0];
    const bool v4 = w % 4 == 0;
    const mm::PackAll pl = mm::pack_layout(w);
    const float *pk = t->gpack + (size_t)L.aux * gemm_pack_floats(w);   // packed by this step's forward pass
    // this coupling's filter gradients: the partial products of the pixel-K GEMMs, summed when the step's gradients are assembled
    float *dW1 = t->gdw + (size_t)(3 * L.aux) * gemm_part_floats(w), *dW2 = dW1 + gemm_part_floats(w), *dW3r = dW2 + gemm_part_floats(w);
    int *np = t->gnp + 3 * L.aux;
    hipLaunchKernelGGL(k_g_c3_bwd, dim3(nb), dim3(TB), 0, st, g, w, zin, P, off_w3, invB, t->dz, gu, G, zlat, (const float *)c.u);
    hipLaunchKernelGGL(k_g_gather36, dim3(nb), dim3(kG36T), 0, st, g, w, (const float *)gu, t->gp36, off_w3, G);
    const mm::Ctx cx{t->n_cu, t->device};
    bool ok = true;
    mm::KpixArgs k{};
    k.npix = g.npix;
    k.nslot = g.nslot;
    k.part_cap = (int64_t)gemm_part_cap(w);
    mm::PixArgs a{};
    a.P = g.npix;
    a.nslot = g.nslot;
    // ---- d l_last/W = relu(bn2(h2 + b2))^T . G36 — and the two batch sums of BN2's backward over g_a2 = G36 . W3r^T.  At wide
    //      couplings the same kernel also forms mask2^T . G36: contracted with the filter the two products ARE those sums
    //      (k_g_bnb_from_parts) and no pass over a [pixel][w] tensor is spent on them (width 512: 228 -> 186 us per coupling); below
    //      256 channels the sums-only pass over the K = 36 product is the cheaper way (width 64: 3.7 against 3.9 ms per step) ----
    const bool dual = w >= 256;
    k.M = w; k.N = 36;
    k.A = c.h2; k.lda = w; k.B = t->gp36; k.ldb = 36; k.part = dW3r;
    k.abias = P + off_b2; k.abn = bn2;
    if (dual) np[2] = v4 ? mm::mm_kpix_launch<2, 2, 1, 3, 4, 4>(cx, st, k) : mm::mm_kpix_launch<2, 2, 1, 3, 1, 4>(cx, st, k);
    else if (w > 64) np[2] = v4 ? mm::mm_kpix_launch<2, 2, 1, 1, 4, 4>(cx, st, k) : mm::mm_kpix_launch<2, 2, 1, 1, 1, 4>(cx, st, k);
    else np[2] = v4 ? mm::mm_kpix_launch<2, 1, 1, 1, 4, 4>(cx, st, k) : mm::mm_kpix_launch<2, 1, 1, 1, 1, 4>(cx, st, k);
    ok = ok && np[2] > 0;
    t->gdual[L.aux] = dual;
    a.N = w; a.K = 36;
    a.A = t->gp36; a.lda = 36;
    a.Bt = pk + pl.o_w3b; a.ldb = 36;
    a.C = t1; a.ldc = w;
    a.ebias = P + off_b2; a.ebn = bn2; a.eh = c.h2; a.ldh = w;
    if (dual) {
        hipLaunchKernelGGL(k_g_bnb_from_parts, dim3(w), dim3(64), 0, st, w, np[2], (const float *)dW3r, pk + pl.o_w3b, t->acc(c.d_bs2), g.nslot);
    } else {
        a.stats = t->acc(c.d_bs2).p;
        ok = mm::mm_pix<0, 4, 4>(cx, st, a) && ok;
    }
    sync_slots(t, t->acc(c.d_bs2), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_bnb_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_bs2), w, g.nslot, n, t->d_flt + c.f_bb2);
    // ---- g_h2 = BN2 backward of the masked g_a2 = G36 . W3r^T (K = 36), stored once; d b2 = its column sums ----
    a.ebb = bb2; a.stats = t->acc(off_b2).p;
    ok = mm::mm_pix<0, 3, 4>(cx, st, a) && ok;
    // ---- d l_2/W = relu(bn1(h1 + b1))^T . g_h2 ----
    k.M = w; k.N = w;
    k.A = c.h1; k.lda = w; k.B = t1; k.ldb = w; k.part = dW2;
    k.abias = P + off_b1; k.abn = bn1;
    if (w > 64) np[1] = v4 ? mm::mm_kpix_launch<2, 2, 2, 1, 4, 4>(cx, st, k) : mm::mm_kpix_launch<2, 2, 2, 1, 1, 1>(cx, st, k);
    else np[1] = v4 ? mm::mm_kpix_launch<2, 1, 1, 1, 4, 4>(cx, st, k) : mm::mm_kpix_launch<2, 1, 1, 1, 1, 1>(cx, st, k);
    ok = ok && np[1] > 0;
    // ---- g_a1 = g_h2 . W2^T and the two batch sums of BN1's backward ----
    a.N = w; a.K = w;
    a.A = t1; a.lda = w;
    a.Bt = pk + pl.o_w2; a.ldb = pl.w4;
    a.C = t2; a.ldc = w;
    a.ebias = P + off_b1; a.ebn = bn1; a.eh = c.h1; a.ldh = w; a.ebb = nullptr; a.stats = t->acc(c.d_bs1).p;
    ok = (v4 ? mm::mm_pix<0, 2, 4>(cx, st, a) : mm::mm_pix<0, 2, 1>(cx, st, a)) && ok;
    sync_slots(t, t->acc(c.d_bs1), 2 * w, g.nslot, st);
    hipLaunchKernelGGL(k_bnb_fin, dim3(w), dim3(64), 0, st, t->acc(c.d_bs1), w, g.nslot, n, t->d_flt + c.f_bb1);
    // ---- g_h1 = BN1 backward of the masked g_a1 is never stored: both of its consumers form it while they stage their tiles ----
    // d l_1/W = Z18^T . g_h1 (and d b1 = its column sums) ;  Q = g_h1 . W1^T
    k.M = 18; k.N = w;
    k.A = t->gz18 + (size_t)L.aux * t->gz18_stride; k.lda = kZ18;   // the forward pass's windows of this coupling's input
    k.B = t2; k.ldb = w; k.part = dW1;
    k.abias = nullptr; k.abn = nullptr;
    k.B2 = c.h1; k.bbias = P + off_b1; k.bbn = bn1; k.bbb = bb1; k.dbias = t->acc(off_b1).p;
    // channel tile of the workgroup: 256 / 128 / 64 (at width 64 on the 256-wide tile three of four wavefronts multiplied zeros
    // and the staged B tile was three quarters padding: 425 us per coupling at 1 024 patches)
    if (w > 128) np[0] = v4 ? mm::mm_kpix_launch<1, 1, 2, 0, 1, 4, 2>(cx, st, k) : mm::mm_kpix_launch<1, 1, 2, 0, 1, 1, 2>(cx, st, k);
    else if (w > 64) np[0] = v4 ? mm::mm_kpix_launch<1, 1, 1, 0, 1, 4, 2>(cx, st, k) : mm::mm_kpix_launch<1, 1, 1, 0, 1, 1, 2>(cx, st, k);
    else np[0] = v4 ? mm::mm_kpix_launch<2, 1, 1, 0, 1, 4, 2>(cx, st, k) : mm::mm_kpix_launch<2, 1, 1, 0, 1, 1, 2>(cx, st, k);
    ok = ok && np[0] > 0;
    a.N = 18; a.K = w;
    a.A = t2; a.lda = w; a.A2 = c.h1;
    a.abias = P + off_b1; a.abn = bn1; a.abb = bb1;
    a.Bt = pk + pl.o_w1; a.ldb = pl.w4;
    a.C = t->gq18; a.ldc = 18;
    a.ebias = nullptr; a.ebn = nullptr; a.eh = nullptr; a.stats = nullptr;
    ok = (v4 ? mm::mm_pix<2, 0, 4>(cx, st, a) : mm::mm_pix<2, 0, 1>(cx, st, a)) && ok;
    if (zmix_in)
        hipLaunchKernelGGL(k_g_c1_dz<true>, dim3(nb), dim3(TB), 0, st, g, (const float *)t->gq18, t->dz, zmix_in, A, dA);
    else
        hipLaunchKernelGGL(k_g_c1_dz<false>, dim3(nb), dim3(TB), 0, st, g, (const float *)t->gq18, t->dz, (const float *)nullptr,
                           (const float *)nullptr, dA);
    return ok;
}

// ---- evaluation under batch statistics on this path (nf_bs_wide_run in nf_train.hip) --------------------------------------------
// nf_*_batchstats (the reference's is_training=True graphs, layers.py:386-398, what NoiseFlowWrapper.py:86 runs) at the coupling
// widths and patch sizes the fused kernels' statistics passes do not take: the layers are walked one by one over ONE resident
// [B,H,W,4] tensor, the coupling CNN runs as in a training step's forward pass (coupling_cnn_gemm: batch sums in the GEMM
// epilogues, slotted, cross-rank hook), and what a training step keeps only as batch totals is kept PER PATCH here — the
// kernels below give every patch one workgroup, so its log-det share is one sum in a fixed order, no atomics.

// one value per workgroup: fp64 sum of the threads' partials, returned to thread 0
__device__ __forceinline__ double wg_sum(double v, double *sh)
{
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < TB / 64; ++i) tot += sh[i];
    return tot;
}

// input of the walk: in * scale, or (in = NULL) the Philox draw of the fused kernels (same key: seed, patch, pixel, stream)
__global__ void k_e_input(Geo g, const float *__restrict__ in, float scale, uint64_t seed, int64_t patch_base, float *__restrict__ out)
{
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            float v[4];
            if (in) {
                const float4 t = reinterpret_cast<const float4 *>(in)[p];
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                const int64_t b = p / g.HW;
                philox_normal4(seed, patch_base + b, (uint32_t)(p - b * g.HW), NF_STREAM_SAMP, v);
            }
            reinterpret_cast<float4 *>(out)[p] = make_float4(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
        }
    }
}

// AffineCouplingSdnEx5 and relatives (cond_utils.py:205-239), in place.  NLL direction: z / scale, log-det -= sum log(scale);
// sampling: z * scale.  One workgroup per patch.
template <bool INV>
__global__ __launch_bounds__(TB) void k_e_sdn(int HW, float *__restrict__ z, const float *__restrict__ y, const float *__restrict__ ab,
                                               double *__restrict__ ldp)
{
    __shared__ double sh[TB / 64];
    const float a = ab[0], b = ab[1];
    const int64_t base = (int64_t)blockIdx.x * HW;
    double l = 0.0;
    for (int i = threadIdx.x; i < HW; i += TB) {
        const float4 zv = reinterpret_cast<const float4 *>(z)[base + i], yv = reinterpret_cast<const float4 *>(y)[base + i];
        const float s0 = sqrtf(fmaf(a, yv.x, b)), s1 = sqrtf(fmaf(a, yv.y, b)), s2 = sqrtf(fmaf(a, yv.z, b)), s3 = sqrtf(fmaf(a, yv.w, b));
        if (INV) {
            reinterpret_cast<float4 *>(z)[base + i] = make_float4(zv.x * s0, zv.y * s1, zv.z * s2, zv.w * s3);
        } else {
            reinterpret_cast<float4 *>(z)[base + i] = make_float4(zv.x / s0, zv.y / s1, zv.z / s2, zv.w / s3);
            l -= (double)(logf(s0) + logf(s1)) + (double)(logf(s2) + logf(s3));
        }
    }
    if (!INV) {
        const double tot = wg_sum(l, sh);
        if (threadIdx.x == 0) ldp[blockIdx.x] += tot;
    }
}

// the gain family in the sampling direction: z * gain (k_scale_fwd divides)
__global__ void k_e_scale_mul(Geo g, float *__restrict__ z, const float *__restrict__ gain)
{
    const float s = gain[0];
    NF_PIXEL_LOOP(g, p) {
        if (p < g.npix) {
            const float4 v = reinterpret_cast<const float4 *>(z)[p];
            reinterpret_cast<float4 *>(z)[p] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
        }
    }
}

// inverse of every Conv2d1x1 matrix k_prep formed (layers.py:108-115: sampling multiplies by A^-1): Gauss-Jordan with partial
// pivoting in fp64, one thread per matrix
__global__ void k_e_inv4(int n, const float *__restrict__ A, float *__restrict__ Ainv)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = A[m * 16 + i * 4 + j];
            a[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        for (int j = 0; j < 8; ++j) {
            const double t = a[c][j];
            a[c][j] = a[piv][j];
            a[piv][j] = t;
        }
        const double d = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) Ainv[m * 16 + i * 4 + j] = (float)a[i][4 + j];
}

// the affine half of a coupling behind its CNN, in place (layers.py:275-291, 355-375, 555-583, 651-674): u = the 9 taps of P36 +
// the edge channel's weight where a tap falls on the padding ring + b3; NLL direction z1 = z1 exp(ls) + shift, log-det += sum ls;
// sampling z1 = (z1 - shift) exp(-ls).  One workgroup per patch.
template <bool INV>
__global__ __launch_bounds__(TB) void k_e_c3(int H, int W, int w, float *__restrict__ z, const float *__restrict__ P36, const float *__restrict__ Pw,
                                              int off_w3, double *__restrict__ ldp)
{
    __shared__ double sh[TB / 64];
    const float *W3 = Pw + off_w3, *b3 = W3 + 36 * (w + 1), *logs = b3 + 4;
    const float sc = logs[4];
    const float e30 = expf(kLogscale * logs[0]), e31 = expf(kLogscale * logs[1]), e32 = expf(kLogscale * logs[2]), e33 = expf(kLogscale * logs[3]);
    const int HW = H * W;
    const int64_t base = (int64_t)blockIdx.x * HW;
    double l = 0.0;
    for (int i = threadIdx.x; i < HW; i += TB) {
        const int r = i / W, c = i - r * W;
        float u[4] = {b3[0], b3[1], b3[2], b3[3]};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int rr = r + tap / 3 - 1, cc = c + tap % 3 - 1;
            float4 v;
            if (rr < 0 || rr >= H || cc < 0 || cc >= W) {
                const float *e = W3 + (tap * (w + 1) + w) * 4;
                v = make_float4(e[0], e[1], e[2], e[3]);
            } else {
                v = *reinterpret_cast<const float4 *>(P36 + (base + rr * W + cc) * 36 + tap * 4);
            }
            u[0] += v.x; u[1] += v.y; u[2] += v.z; u[3] += v.w;
        }
        const float4 zi = reinterpret_cast<const float4 *>(z)[base + i];
        const float sh0 = u[0] * e30, sh1 = u[1] * e31;
        const float ls0 = sc * tanhf(u[2] * e32), ls1 = sc * tanhf(u[3] * e33);
        if (INV) {
            reinterpret_cast<float4 *>(z)[base + i] = make_float4(zi.x, zi.y, (zi.z - sh0) * expf(-ls0), (zi.w - sh1) * expf(-ls1));
        } else {
            reinterpret_cast<float4 *>(z)[base + i] = make_float4(zi.x, zi.y, fmaf(zi.z, expf(ls0), sh0), fmaf(zi.w, expf(ls1), sh1));
            l += (double)ls0 + (double)ls1;
        }
    }
    if (!INV) {
        const double tot = wg_sum(l, sh);
        if (threadIdx.x == 0) ldp[blockIdx.x] += tot;
    }
}

// per-patch outputs of the NLL direction (noise_flow_model.py:394-428, 477-478, 537-539, as gemm_epilogue of the fused kernels):
// log-det = data part + constant, nll = -log-det (+ prior), sd_z; the call's sums (sum nll, sum sd, B).  One workgroup per patch.
__global__ __launch_bounds__(TB) void k_e_finish(int HW, const float *__restrict__ z, const double *__restrict__ ldp, const double *__restrict__ ldc,
                                                  int prior, float *__restrict__ nll_out, float *__restrict__ sd_out, float *__restrict__ ld_out,
                                                  double *__restrict__ sums)
{
    __shared__ double sh[TB / 64];
    const int64_t base = (int64_t)blockIdx.x * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < HW; i += TB) {
        const float4 v = reinterpret_cast<const float4 *>(z)[base + i];
        s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    const double t1 = wg_sum(s1, sh), t2 = wg_sum(s2, sh);
    if (threadIdx.x == 0) {
        const double npx = (double)HW * 4.0, logdet = ldp[blockIdx.x] + ldc[0];
        double nll = -logdet;
        if (prior) nll += 0.5 * npx * 1.8378770664093453 + 0.5 * t2;
        const double mean = t1 / npx;
        double var = t2 / npx - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double sd = sqrt(var);
        if (nll_out) nll_out[blockIdx.x] = (float)nll;
        if (sd_out) sd_out[blockIdx.x] = (float)sd;
        if (ld_out) ld_out[blockIdx.x] = (float)logdet;
        if (sums) {
            atomicAdd(&sums[0], (double)(float)nll);
            atomicAdd(&sums[1], (double)(float)sd);
            atomicAdd(&sums[2], 1.0);
        }
    }
}

}  // namespace
