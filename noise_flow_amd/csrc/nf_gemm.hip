// Fused Noise Flow stack for WIDE coupling CNNs (hps.width 33 .. 512) as LDS-staged GEMMs on the f32 matrix cores of gfx950.
//
// sidd/ArgParser.py:43 defaults --width to Glow's 512 and layers.py:452-498 accepts any width.  At w = 512 a coupling CNN is
// 290 kMAC per pixel, 90 % of it the 1x1 convolution l_2: [pixels x w] . [w x w] — the one place this model is a real GEMM.
// Same program, same I/O, same per-patch workgroup and epilogue as nf_kernels.hip / nf_wide.hip; what changes is where the
// hidden activations live.  A 32x32 patch at w = 512 has 2 MiB of them per layer: neither registers nor LDS hold a patch,
// so the CNN is evaluated BAND by band:
//
//  * a band = NB = 32768 / WP consecutive pixels (row-major), WP = w zero-padded to 64 / 128 / 256 / 512 (exact: a padded
//    channel has zero weights and bias, hence zero activation).  h1 of the band — WP x NB floats = 128 KiB — sits in LDS in
//    MFMA B-operand order; 512 threads = 8 wavefronts, one workgroup per CU.
//  * tiles as in nf_wide.hip: 32 pixels on the N axis of v_mfma_f32_32x32x2_f32 (exact fp32), channels on M; D register v of
//    lane half g = channel c(v, g) of its tile, which is register by register the B operand of the next layer's K step v.
//  * l_1 (K = 18): B operands from a zero-bordered LDS tile of the pass-through half; its 32 output tiles are dealt 4 per
//    wavefront; ReLU'd results are parked in LDS as [K chunk of 8 channels][lane half][pixel][4] — the lane that will need
//    them as B operands of l_2 reads back one ds_read_b128 per 4 K steps.
//  * l_2: every wavefront owns 2 output-channel tiles x 2 pixel tiles (4 accumulator tiles, 64 VGPRs) over the whole K.  Its
//    A operands (the w x w weights, 1 MiB at w = 512) are NOT staged in LDS: each weight is consumed by exactly one wavefront
//    of the workgroup, so that wavefront streams them straight from L2 in fetch order (one coalesced 16-byte load per lane
//    per 4 K steps, software-pipelined one chunk ahead).  Per workgroup and band the whole weight set crosses L2 -> CU
//    once; all workgroups of an XCD walk the couplings in step, so the 1.2 MiB of a coupling stay L2-resident.
//  * l_last is evaluated transposed, as in nf_wide.hip: P[pixel][tap][j] = sum_c h2[pixel][c] W3[tap][c][j] straight from
//    the ReLU'd l_2 accumulators (h2 never exists in memory).  A wavefront only holds its own 64 channels, so the WM = WP/64
//    partial P tiles of a pixel meet in LDS (the h1 region is dead by then), and every thread then gathers the 9 taps of the
//    output pixels it owns:  o[r][c] += sum_taps P[r+di-1][c+dj-1][tap] over the pixels of THIS band.  'SAME' padding falls
//    out of the bounds checks, the edge-indicator channel is the 16-entry border table of the other kernels.
//
// Replaces (reference, /root/reference): layers.py:251-375 (AffineCoupling), :452-498 (real_nvp_conv_template), :555-613,
// :651-674 (conv2d / add_edge_padding / conv2d_zeros) at hps.width > 32 (sidd/ArgParser.py:43: default 512).
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>
#include "../../include/noiseflow_hip.h"   // NF_SUMS_SLOTS / NF_SUMS_STRIDE
#include "nf_device.h"
#include "nf_gemm_layout.h"
#include "nf_dev_util.h"
#include "nf_gemm_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int GT = 512;          // threads per workgroup
constexpr int GW = GT / 64;      // wavefronts

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

//   WP      padded coupling width: 64, 128, 256, 512
//   PHILOX  input = in-kernel Philox/Box-Muller draw
//   OWN     pixels per thread: 2 (patches <= 1024 pixels), 4 (<= 2048) or 8 (<= 4096: 64x64)
//   BRD     the pass-through tile carries a zero border ('SAME' padding by construction).  Beside the 128 KiB band a bordered
//           64x64 tile does not fit the 160 KiB of a CU by 2 KiB: !BRD keeps the bare H x W planes and masks the taps instead
template <int WP, bool PHILOX, int OWN, bool BRD>
__global__ __launch_bounds__(GT) void nf_gemm_kernel(const NfProgram prog, const NfLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MT = WP / 32;            // channel tiles
    constexpr int NB = NF7_BAND_FLOATS / WP;   // pixels per band
    constexpr int NT = NB / 32;            // pixel tiles per band
    constexpr int WM = MT / 2;             // wavefronts along the channel axis of l_2 (2 tiles each)
    constexpr int WN = GW / WM;            // wavefronts along the pixel axis (2 tiles each)
    constexpr int KC = WP / 8;             // chunks of 4 K steps (8 input channels)
    static_assert(MT * NT == 32 && WM * WN == GW && NT == 2 * WN && KC % 2 == 0, "tile split");
    static_assert(WM * NB * NF7_P_STRIDE <= NF7_BAND_FLOATS, "the partial P tiles reuse the h1 region");
    const int H = a.H, W = a.W, HW = H * W;
    const int Wp = BRD ? W + 2 : W;
    const int PL = ((BRD ? (H + 2) * Wp : HW) + 3) & ~3;   // one channel plane of the z0 tile
    float *const h1 = smem;                             // [KC][2][NB][4]; later the partial P tiles [WM][NB][NF7_P_STRIDE]
    float *const z0s = smem + NF7_BAND_FLOATS;          // [2][PL]
    float *const red = BRD ? z0s + 2 * PL : h1;         // [3][GW]  (!BRD: every byte counts — the band region is idle in the epilogue)
    constexpr int ZB = BRD ? 1 : 0;                     // tile coordinates = pixel coordinates + ZB

    const int t = threadIdx.x;
    const int wv = t >> 6, lane = t & 63, n = lane & 31, g = lane >> 5;
    const int wm = wv / WN, wn = wv % WN;

    for (int i = t; i < 2 * PL; i += GT) z0s[i] = 0.0f;
    __syncthreads();

    // the pixels this thread owns: p = t + GT m
    int pr[OWN], pc[OWN];
    bool act[OWN];
#pragma unroll
    for (int m = 0; m < OWN; ++m) {
        const int p = t + GT * m;
        act[m] = p < HW;
        pr[m] = act[m] ? p / W : 0;
        pc[m] = act[m] ? p - pr[m] * W : 0;
    }

    const int n_ops = prog.n_ops;
    const int n_bands = (HW + NB - 1) / NB;
    double acc_nll = 0.0, acc_sd = 0.0;   // thread 0 only

    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        const GemmTile T = gemm_tile(a, b, H, W);
        float z[OWN][4];
        gemm_input<OWN, PHILOX>(a, T, pr, pc, act, z);

        float ld = 0.0f, ld2 = 0.0f;   // natural-log / log2 parts of this thread's log-det share

        for (int op = 0; op < n_ops; ++op) {
            const int type = prog.ops[op].type;
            const cfloat_p P = (cfloat_p)(a.params + prog.ops[op].off);   // wave-uniform, scalar loads

            if (type == NF_OP_MIX) {
                gemm_mix<OWN>(P, z);
            } else if (type == NF_OP_COUPLING_FWD || type == NF_OP_COUPLING_REV) {
                const float *const img = a.params + prog.ops[op].off + NF7_CPL_IMG;
                // ---- publish the pass-through half ----
#pragma unroll
                for (int m = 0; m < OWN; ++m)
                    if (act[m]) {
                        z0s[(pr[m] + ZB) * Wp + pc[m] + ZB] = z[m][0];
                        z0s[PL + (pr[m] + ZB) * Wp + pc[m] + ZB] = z[m][1];
                    }
                float o[OWN][4];
#pragma unroll
                for (int m = 0; m < OWN; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[m][j] = 0.0f;
                __syncthreads();

                for (int band = 0; band < n_bands; ++band) {
                    const int p0 = band * NB;
                    // ---- l_1: 32 tiles of h1 = relu(W1 z0 + b1), 4 per wavefront, into LDS in B-operand order ----
#pragma unroll 1
                    for (int i = 0; i < 4; ++i) {
                        const int tt = wv * 4 + i, m = tt / NT, nt = tt % NT;
                        int p = p0 + 32 * nt + n;
                        p = p < HW ? p : HW - 1;   // columns past the patch: never gathered
                        const int r = p / W, c = p - r * W;
                        const float *zb = z0s + g * PL + r * Wp + c;   // BRD: tap (di,dj) at + di*Wp + dj
                        [[maybe_unused]] const bool rok[3] = {r > 0, true, r + 1 < H}, cok[3] = {c > 0, true, c + 1 < W};
                        v16f d;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = ldg4(img + nf7_img_B1(WP) + m * 32 + g * 16 + 4 * q);
                            d[4 * q + 0] = bb.x; d[4 * q + 1] = bb.y; d[4 * q + 2] = bb.z; d[4 * q + 3] = bb.w;
                        }
#pragma unroll
                        for (int grp = 0; grp < 3; ++grp) {
                            const float4 aw = ldg4(img + nf7_img_A1(WP) + ((m * 3 + grp) * 64 + lane) * 4);
                            const float as[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const int tap = grp * 4 + s;
                                if (tap < 9) {
                                    float zv;
                                    if constexpr (BRD) {
                                        zv = zb[(tap / 3) * Wp + tap % 3];
                                    } else {   // bare planes: the taps outside the patch are the zeros of 'SAME' padding
                                        const bool ok = rok[tap / 3] && cok[tap % 3];
                                        zv = zb[ok ? (tap / 3 - 1) * Wp + tap % 3 - 1 : 0];
                                        zv = ok ? zv : 0.0f;
                                    }
                                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], zv, d, 0, 0, 0);
                                }
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4 *>(h1 + ((((m * 4 + q) * 2 + g) * NB) + 32 * nt + n) * 4) =
                                make_float4(nf_relu(d[4 * q + 0]), nf_relu(d[4 * q + 1]), nf_relu(d[4 * q + 2]), nf_relu(d[4 * q + 3]));
                    }
                    __syncthreads();

                    // ---- l_2: 2 x 2 accumulator tiles per wavefront over the whole K; weights streamed from L2 ----
                    v16f acc[2][2];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = ldg4(img + nf7_img_B2(WP) + (2 * wm + mi) * 32 + g * 16 + 4 * q);
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                acc[mi][ni][4 * q + 0] = bb.x; acc[mi][ni][4 * q + 1] = bb.y;
                                acc[mi][ni][4 * q + 2] = bb.z; acc[mi][ni][4 * q + 3] = bb.w;
                            }
                        }
                    }
                    {
                        const float *ap0 = img + nf7_img_A2(WP) + ((size_t)(2 * wm + 0) * KC * 64 + lane) * 4;
                        const float *ap1 = img + nf7_img_A2(WP) + ((size_t)(2 * wm + 1) * KC * 64 + lane) * 4;
                        const float *bp0 = h1 + (g * NB + 32 * (2 * wn + 0) + n) * 4;
                        const float *bp1 = h1 + (g * NB + 32 * (2 * wn + 1) + n) * 4;
                        // Software pipeline over K chunks of 4 steps, two operand sets in ping-pong: the loads of chunk kc + 1 are
                        // ISSUED before the 16 MFMAs of chunk kc.  The scheduling barriers keep it that way — left alone, the
                        // machine scheduler sinks every load to just before its first use (measured: 0.72 of peak with the loads
                        // 4 MFMAs ahead of their use, an L2 hit being ~20 MFMAs away).
                        float4 xa0 = ldg4(ap0), xa1 = ldg4(ap1);
                        float4 xb0 = *reinterpret_cast<const float4 *>(bp0), xb1 = *reinterpret_cast<const float4 *>(bp1);
                        float4 ya0, ya1, yb0, yb1;
#define NF_GEMM_CHUNK(A0, A1, B0, B1)                                                                              \
    {                                                                                                              \
        const float as0[4] = {A0.x, A0.y, A0.z, A0.w}, as1[4] = {A1.x, A1.y, A1.z, A1.w};                          \
        const float bs0[4] = {B0.x, B0.y, B0.z, B0.w}, bs1[4] = {B1.x, B1.y, B1.z, B1.w};                          \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                              \
        {                                                                                                          \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(as0[s], bs0[s], acc[0][0], 0, 0, 0);                  \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(as0[s], bs1[s], acc[0][1], 0, 0, 0);                  \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(as1[s], bs0[s], acc[1][0], 0, 0, 0);                  \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(as1[s], bs1[s], acc[1][1], 0, 0, 0);                  \
        }                                                                                                          \
    }
#pragma unroll 1
                        for (int kc = 0; kc < KC; kc += 2) {
                            const int k1 = kc + 1, k2 = kc + 2 < KC ? kc + 2 : KC - 1;
                            ya0 = ldg4(ap0 + (size_t)k1 * 256);
                            ya1 = ldg4(ap1 + (size_t)k1 * 256);
                            yb0 = *reinterpret_cast<const float4 *>(bp0 + (size_t)k1 * 2 * NB * 4);
                            yb1 = *reinterpret_cast<const float4 *>(bp1 + (size_t)k1 * 2 * NB * 4);
                            __builtin_amdgcn_sched_barrier(0);
                            NF_GEMM_CHUNK(xa0, xa1, xb0, xb1)
                            __builtin_amdgcn_sched_barrier(0);
                            xa0 = ldg4(ap0 + (size_t)k2 * 256);
                            xa1 = ldg4(ap1 + (size_t)k2 * 256);
                            xb0 = *reinterpret_cast<const float4 *>(bp0 + (size_t)k2 * 2 * NB * 4);
                            xb1 = *reinterpret_cast<const float4 *>(bp1 + (size_t)k2 * 2 * NB * 4);
                            __builtin_amdgcn_sched_barrier(0);
                            NF_GEMM_CHUNK(ya0, ya1, yb0, yb1)
                            __builtin_amdgcn_sched_barrier(0);
                        }
#undef NF_GEMM_CHUNK
                    }
                    // ---- P = W3^T relu(h2): this wavefront's 64 channels; taps 0 .. 7 as one 32-row tile, tap 8 on 4x4x1 ----
                    v16f pa[2];
                    v4f p8[2];
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                        for (int v = 0; v < 16; ++v) pa[ni][v] = 0.0f;
                        p8[ni] = v4f{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                        for (int grp = 0; grp < 4; ++grp) {
                            const float4 w0 = ldg4(img + nf7_img_A3(WP) + ((((2 * wm + mi) * 4 + grp) * 64) + lane) * 4);
                            const float4 wc = ldg4(img + nf7_img_A3C(WP) + ((((2 * wm + mi) * 4 + grp) * 8) + g * 4 + (lane & 3)) * 4);
                            const float ws0[4] = {w0.x, w0.y, w0.z, w0.w}, wcs[4] = {wc.x, wc.y, wc.z, wc.w};
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const float hA = nf_relu(acc[mi][0][grp * 4 + s]), hB = nf_relu(acc[mi][1][grp * 4 + s]);
                                pa[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws0[s], hA, pa[0], 0, 0, 0);
                                pa[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws0[s], hB, pa[1], 0, 0, 0);
                                p8[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wcs[s], hA, p8[0], 0, 0, 0);
                                p8[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wcs[s], hB, p8[1], 0, 0, 0);
                            }
                        }
                    }
                    __syncthreads();   // every wavefront is done with h1: the region becomes the partial P tiles

                    // per pixel [tap 0..7][j] (register group a of lane half g holds tap 2 a + g), then tap 8 of lane half 0 / 1
                    float *const pp = h1;   // [WM][NB][NF7_P_STRIDE]
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        float *dst = pp + ((size_t)(wm * NB + 32 * (2 * wn + ni) + n)) * NF7_P_STRIDE;
#pragma unroll
                        for (int aa = 0; aa < 4; ++aa)
                            *reinterpret_cast<float4 *>(dst + (2 * aa + g) * 4) =
                                make_float4(pa[ni][4 * aa + 0], pa[ni][4 * aa + 1], pa[ni][4 * aa + 2], pa[ni][4 * aa + 3]);
                        *reinterpret_cast<float4 *>(dst + 32 + 4 * g) = make_float4(p8[ni][0], p8[ni][1], p8[ni][2], p8[ni][3]);
                    }
                    __syncthreads();

                    // ---- gather: the taps of this band's pixels that fall on the output pixels this thread owns ----
#pragma unroll
                    for (int m = 0; m < OWN; ++m) {
                        const int q = t + GT * m;
                        if (!act[m] || q + W + 1 < p0 || q >= p0 + NB + W + 1) continue;
#pragma unroll
                        for (int di = 0; di < 3; ++di) {
                            const int rr = pr[m] + di - 1;
                            if (rr < 0 || rr >= H) continue;
#pragma unroll
                            for (int dj = 0; dj < 3; ++dj) {
                                const int cc = pc[m] + dj - 1;
                                const int src = rr * W + cc - p0;
                                if (cc < 0 || cc >= W || src < 0 || src >= NB) continue;
#pragma unroll
                                for (int k = 0; k < WM; ++k) {
                                    const float *rec = pp + ((size_t)(k * NB + src)) * NF7_P_STRIDE;
                                    float4 v = *reinterpret_cast<const float4 *>(rec + (di * 3 + dj) * 4);
                                    if (di * 3 + dj == 8) {   // tap 8: the two lane halves' partial sums
                                        const float4 u = *reinterpret_cast<const float4 *>(rec + 36);
                                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                                    }
                                    o[m][0] += v.x; o[m][1] += v.y; o[m][2] += v.z; o[m][3] += v.w;
                                }
                            }
                        }
                    }
                    __syncthreads();   // the next band's l_1 overwrites the region
                }

                // ---- finish the coupling on the owned pixels ----
                gemm_finish_coupling<OWN, false>(type, a.params + prog.ops[op].off + NF7_CPL_E, P[NF7_CPL_S + 1], P[NF7_CPL_S + 2], T, pr, pc, act, o, z, ld2);
            } else if (type == NF_OP_SDN_DIV || type == NF_OP_SDN_MUL) {
                gemm_sdn<OWN>(type, prog.ops[op].off, a, T, pr, pc, act, z, ld);
            } else if (type == NF_OP_SCALE || type == NF_OP_SCALE_COND) {
                gemm_scale<OWN>(type == NF_OP_SCALE ? P[0] : a.cond_a[prog.ops[op].off & 3], z);
            }
        }

        gemm_epilogue<OWN, GT>(a, T, b, HW, pr, pc, act, z, ld, ld2, red, acc_nll, acc_sd);
    }
    gemm_flush_sums(a, acc_nll, acc_sd);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Variant B (widths <= 128): pixel tiles, not channel tiles, per wavefront.  Every wavefront owns ONE tile of 32 pixels per round
// (8 wavefronts = 256 pixels) and computes ALL channels for it: its h1 stays in its registers (WP / 2 floats per lane), the l_2
// accumulators of an output tile go straight into the transposed l_last (no cross-wavefront partial P sums), and the weights come
// through LDS — one slab per output tile (nf_gemm_layout.h, NF10_*), all slabs of a coupling staged once per patch and resident
// for its rounds (85 KiB at width 128).  Nothing is re-streamed per band and the band bookkeeping of variant A (4 barriers, a
// partial-sum gather over WM wavefronts) shrinks to 2 barriers per round.  Beyond 128 the slabs do not fit and variant A runs.
template <int WP, bool PHILOX, int OWN>
__global__ __launch_bounds__(GT) void nf_gemmb_kernel(const NfProgram prog, const NfLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MT = WP / 32;                // channel tiles
    constexpr int KC = WP / 8;                 // chunks of 4 K steps of l_2
    constexpr int RND = 32 * GW;               // pixels per round
    constexpr int SLAB = nf10_slab_floats(WP);
    static_assert(MT * SLAB * 4 <= 96 * 1024, "variant B keeps every slab of the coupling resident in LDS (widths <= 128)");
    const int H = a.H, W = a.W, HW = H * W;
    const int Wp = W + 2;
    const int PL = ((H + 2) * Wp + 3) & ~3;            // one channel plane of the z0 tile
    float *const wb = smem;                             // [MT][SLAB] weight slabs
    float *const prec = wb + MT * SLAB;                 // [RND][NF7_P_STRIDE] P records of the round
    float *const z0s = prec + RND * NF7_P_STRIDE;       // [2][PL]
    float *const red = z0s + 2 * PL;                    // [3][GW]

    const int t = threadIdx.x;
    const int wv = t >> 6, lane = t & 63, n = lane & 31, g = lane >> 5;

    for (int i = t; i < 2 * PL; i += GT) z0s[i] = 0.0f;
    __syncthreads();
    // the pixels this thread owns: p = t + GT m
    int pr[OWN], pc[OWN];
    bool act[OWN];
#pragma unroll
    for (int m = 0; m < OWN; ++m) {
        const int p = t + GT * m;
        act[m] = p < HW;
        pr[m] = act[m] ? p / W : 0;
        pc[m] = act[m] ? p - pr[m] * W : 0;
    }

    const int n_ops = prog.n_ops;
    const int n_rounds = (HW + RND - 1) / RND;
    double acc_nll = 0.0, acc_sd = 0.0;   // thread 0 only

    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        const GemmTile T = gemm_tile(a, b, H, W);
        float z[OWN][4];
        gemm_input<OWN, PHILOX>(a, T, pr, pc, act, z);

        float ld = 0.0f, ld2 = 0.0f;   // natural-log / log2 parts of this thread's log-det share

        for (int op = 0; op < n_ops; ++op) {
            const int type = prog.ops[op].type;
            const cfloat_p P = (cfloat_p)(a.params + prog.ops[op].off);   // wave-uniform, scalar loads

            if (type == NF_OP_MIX) {
                gemm_mix<OWN>(P, z);
            } else if (type == NF_OP_COUPLING_FWD || type == NF_OP_COUPLING_REV) {
                const float *const img = a.params + prog.ops[op].off + NF7_CPL_IMG;
                // ---- publish the pass-through half, stage this coupling's slabs ----
#pragma unroll
                for (int m = 0; m < OWN; ++m)
                    if (act[m]) {
                        z0s[(pr[m] + 1) * Wp + pc[m] + 1] = z[m][0];
                        z0s[PL + (pr[m] + 1) * Wp + pc[m] + 1] = z[m][1];
                    }
                float o[OWN][4];
#pragma unroll
                for (int m = 0; m < OWN; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[m][j] = 0.0f;
                for (int i = t; i < MT * (SLAB / 4); i += GT)
                    reinterpret_cast<float4 *>(wb)[i] = reinterpret_cast<const float4 *>(img + nf10_img_SLAB(WP))[i];
                __syncthreads();

                for (int rnd = 0; rnd < n_rounds; ++rnd) {
                    const int p0 = rnd * RND;
                    // ---- l_1: relu(W1 z0 + b1) for this wavefront's 32 pixels, all channels, into registers ----
                    float hr[MT][16];
                    {
                        int p = p0 + 32 * wv + n;
                        p = p < HW ? p : HW - 1;   // columns past the patch: never gathered
                        const int r = p / W, c = p - r * W;
                        const float *zb = z0s + g * PL + r * Wp + c;   // tap (di,dj) at + di*Wp + dj
                        float bt[9];
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) bt[tap] = zb[(tap / 3) * Wp + tap % 3];
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            v16f d;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 bb = ldg4(img + nf7_img_B1(WP) + m * 32 + g * 16 + 4 * q);
                                d[4 * q + 0] = bb.x; d[4 * q + 1] = bb.y; d[4 * q + 2] = bb.z; d[4 * q + 3] = bb.w;
                            }
#pragma unroll
                            for (int grp = 0; grp < 3; ++grp) {
                                const float4 aw = ldg4(img + nf7_img_A1(WP) + ((m * 3 + grp) * 64 + lane) * 4);
                                const float as[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
                                for (int s = 0; s < 4; ++s)
                                    if (grp * 4 + s < 9) d = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], bt[grp * 4 + s], d, 0, 0, 0);
                            }
#pragma unroll
                            for (int v = 0; v < 16; ++v) hr[m][v] = nf_relu(d[v]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // ---- per output tile: l_2 over the whole K from the slab in LDS, then its share of P = W3^T relu(h2) ----
                    v16f pa;
                    v4f p8 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < 16; ++v) pa[v] = 0.0f;
#pragma unroll 1
                    for (int m = 0; m < MT; ++m) {
                        const float *sl = wb + m * SLAB;
                        v16f acc;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = *reinterpret_cast<const float4 *>(sl + nf10_slab_B2(WP) + g * 16 + 4 * q);
                            acc[4 * q + 0] = bb.x; acc[4 * q + 1] = bb.y; acc[4 * q + 2] = bb.z; acc[4 * q + 3] = bb.w;
                        }
                        const float4 *ap = reinterpret_cast<const float4 *>(sl) + lane;   // chunk kc (K steps 4 kc .. 4 kc + 3) at + 64 kc
                        float4 ca = ap[0], na = ca;
#pragma unroll
                        for (int kc = 0; kc < KC; ++kc) {
                            if (kc + 1 < KC) na = ap[64 * (kc + 1)];
                            __builtin_amdgcn_sched_barrier(0);
                            const float as[4] = {ca.x, ca.y, ca.z, ca.w};
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const int kk = 4 * kc + s;   // consumes register kk % 16 of input tile kk / 16
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], hr[kk >> 4][kk & 15], acc, 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            ca = na;
                        }
                        // h2 tile m complete: ReLU and straight into P (taps 0 .. 7: one 32-row tile; tap 8: 4x4x1)
#pragma unroll
                        for (int grp = 0; grp < 4; ++grp) {
                            const float4 w0 = *reinterpret_cast<const float4 *>(sl + nf10_slab_A3(WP) + (grp * 64 + lane) * 4);
                            const float4 wc = *reinterpret_cast<const float4 *>(sl + nf10_slab_A3C(WP) + (grp * 8 + g * 4 + (lane & 3)) * 4);
                            const float ws0[4] = {w0.x, w0.y, w0.z, w0.w}, wcs[4] = {wc.x, wc.y, wc.z, wc.w};
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const float h = nf_relu(acc[grp * 4 + s]);
                                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(ws0[s], h, pa, 0, 0, 0);
                                p8 = __builtin_amdgcn_mfma_f32_4x4x1f32(wcs[s], h, p8, 0, 0, 0);
                            }
                        }
                    }
                    // ---- P records of the round ----
                    {
                        float *dst = prec + (size_t)(32 * wv + n) * NF7_P_STRIDE;
#pragma unroll
                        for (int aa = 0; aa < 4; ++aa)
                            *reinterpret_cast<float4 *>(dst + (2 * aa + g) * 4) = make_float4(pa[4 * aa + 0], pa[4 * aa + 1], pa[4 * aa + 2], pa[4 * aa + 3]);
                        *reinterpret_cast<float4 *>(dst + 32 + 4 * g) = make_float4(p8[0], p8[1], p8[2], p8[3]);
                    }
                    __syncthreads();
                    // gather: the taps of this round's pixels that fall on the output pixels this thread owns
#pragma unroll
                    for (int m = 0; m < OWN; ++m) {
                        const int q = t + GT * m;
                        if (!act[m] || q + W + 1 < p0 || q >= p0 + RND + W + 1) continue;
#pragma unroll
                        for (int di = 0; di < 3; ++di) {
                            const int rr = pr[m] + di - 1;
                            if (rr < 0 || rr >= H) continue;
#pragma unroll
                            for (int dj = 0; dj < 3; ++dj) {
                                const int cc = pc[m] + dj - 1;
                                const int src = rr * W + cc - p0;
                                if (cc < 0 || cc >= W || src < 0 || src >= RND) continue;
                                const float *rp = prec + (size_t)src * NF7_P_STRIDE;
                                float4 v = *reinterpret_cast<const float4 *>(rp + (di * 3 + dj) * 4);
                                if (di * 3 + dj == 8) {   // tap 8: the two lane halves' partial sums
                                    const float4 u = *reinterpret_cast<const float4 *>(rp + 36);
                                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                                }
                                o[m][0] += v.x; o[m][1] += v.y; o[m][2] += v.z; o[m][3] += v.w;
                            }
                        }
                    }
                    __syncthreads();   // the next round overwrites the records
                }

                // ---- finish the coupling on the owned pixels ----
                gemm_finish_coupling<OWN, false>(type, a.params + prog.ops[op].off + NF7_CPL_E, P[NF7_CPL_S + 1], P[NF7_CPL_S + 2], T, pr, pc, act, o, z, ld2);
            } else if (type == NF_OP_SDN_DIV || type == NF_OP_SDN_MUL) {
                gemm_sdn<OWN>(type, prog.ops[op].off, a, T, pr, pc, act, z, ld);
            } else if (type == NF_OP_SCALE || type == NF_OP_SCALE_COND) {
                gemm_scale<OWN>(type == NF_OP_SCALE ? P[0] : a.cond_a[prog.ops[op].off & 3], z);
            }
        }

        gemm_epilogue<OWN, GT>(a, T, b, HW, pr, pc, act, z, ld, ld2, red, acc_nll, acc_sd);
    }
    gemm_flush_sums(a, acc_nll, acc_sd);
}

size_t gemmb_lds_bytes(int wp, int H, int W)
{
    const int Wp = W + 2, PL = ((H + 2) * Wp + 3) & ~3;
    return ((size_t)(wp / 32) * nf10_slab_floats(wp) + (size_t)32 * GW * NF7_P_STRIDE + 2 * (size_t)PL + 3 * GW + 8) * sizeof(float);
}

template <int WP, bool PHILOX, int OWN>
hipError_t launch_gemmb(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    static std::atomic<size_t> lds_set[16];
    return gemm_launch_per_cu<GT>(&nf_gemmb_kernel<WP, PHILOX, OWN>, gemmb_lds_bytes(WP, a.H, a.W), lds_set, prog, a, n_cu, device, stream);
}

template <int WP, bool PHILOX>
hipError_t dispatch_ownb(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    return gemm_by_own<GT>(a.H * a.W, [&](auto own) { return launch_gemmb<WP, PHILOX, decltype(own)::value>(prog, a, n_cu, device, stream); });
}

size_t gemm_lds_bytes(int H, int W, bool bordered = true)
{
    const int Wp = W + 2, PL = ((bordered ? (H + 2) * Wp : H * W) + 3) & ~3;
    return ((size_t)NF7_BAND_FLOATS + 2 * (size_t)PL + (bordered ? 3 * GW + 8 : 0)) * sizeof(float);
}

template <int WP, bool PHILOX, int OWN, bool BRD = true>
hipError_t launch_gemm(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    static std::atomic<size_t> lds_set[16];
    return gemm_launch_per_cu<GT>(&nf_gemm_kernel<WP, PHILOX, OWN, BRD>, gemm_lds_bytes(a.H, a.W, BRD), lds_set, prog, a, n_cu, device, stream);
}

template <int WP, bool PHILOX>
hipError_t dispatch_own(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (a.H * a.W <= 2 * GT) return launch_gemm<WP, PHILOX, 2>(prog, a, n_cu, device, stream);
    if (a.H * a.W <= 4 * GT) return launch_gemm<WP, PHILOX, 4>(prog, a, n_cu, device, stream);
    // up to 64x64: 8 pixels per thread; the bordered tile of the largest shapes does not fit beside the band
    if (gemm_lds_bytes(a.H, a.W, true) <= 160 * 1024) return launch_gemm<WP, PHILOX, 8>(prog, a, n_cu, device, stream);
    return launch_gemm<WP, PHILOX, 8, false>(prog, a, n_cu, device, stream);
}

template <bool PHILOX>
hipError_t dispatch_gemm(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    switch (prog.width) {
    case 64: return dispatch_own<64, PHILOX>(prog, a, n_cu, device, stream);
    case 128: return dispatch_own<128, PHILOX>(prog, a, n_cu, device, stream);
    case 256: return dispatch_own<256, PHILOX>(prog, a, n_cu, device, stream);
    case 512: return dispatch_own<512, PHILOX>(prog, a, n_cu, device, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// variant B: programs in the NF10 layout (widths <= 128)
bool nf_gemmb_shape_ok(int wp, int H, int W)
{
    return wp <= 128 && H >= 1 && W >= 1 && H * W <= NF7_MAX_PIXELS && gemmb_lds_bytes(wp, H, W) <= 160 * 1024;
}
hipError_t nf_launch_gemmb(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (!nf_gemmb_shape_ok(prog.width, a.H, a.W)) return hipErrorInvalidValue;
    const bool ph = (a.flags & NF_K_PHILOX_IN) != 0;
    if (prog.width == 64) return ph ? dispatch_ownb<64, true>(prog, a, n_cu, device, stream) : dispatch_ownb<64, false>(prog, a, n_cu, device, stream);
    return ph ? dispatch_ownb<128, true>(prog, a, n_cu, device, stream) : dispatch_ownb<128, false>(prog, a, n_cu, device, stream);
}

// whether a patch shape fits the GEMM kernel (nf_create asks before accepting a width > 32)
bool nf_gemm_shape_ok(int H, int W)
{
    return H >= 1 && W >= 1 && H * W <= NF7_MAX_PIXELS && gemm_lds_bytes(H, W, false) <= 160 * 1024;
}

// entry point used by nf_host.hip: programs in the NF7 layout (coupling width padded to 64 / 128 / 256 / 512)
hipError_t nf_launch_gemm(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (!nf_gemm_shape_ok(a.H, a.W)) return hipErrorInvalidValue;
    if (a.flags & NF_K_PHILOX_IN) return dispatch_gemm<true>(prog, a, n_cu, device, stream);
    return dispatch_gemm<false>(prog, a, n_cu, device, stream);
}
