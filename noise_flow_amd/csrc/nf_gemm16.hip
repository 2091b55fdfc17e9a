// fp16-CNN variant of the GEMM kernel (nf_gemm.hip): NF_CFG_FP16_CNN at coupling widths 33 .. 512 on v_mfma_f32_32x32x16_f16.
//
// Same band structure, tile convention, transposed l_last and 9-tap gather as nf_gemm.hip; what changes with K = 16 per
// instruction and 16 x the MAC rate:
//  * the three CNN inputs are half precision (the rounding points of the oracle's cnn_dtype='fp16': folded weights, z0,
//    relu(h1), relu(h2), each rounded once; biases, border table, tanh / exp and the log-det stay fp32), so a band holds
//    NB = 65536 / WP pixels (128 at width 512) in the same 128 KiB of LDS, one ds_read_b128 per B operand;
//  * an MFMA lasts 32 cycles and needs 1 KiB of weights: streamed per wavefront as in the fp32 kernel the weight traffic would
//    be 64 B/clk/CU — the whole L2 link.  Every wavefront therefore owns 2 channel tiles x FOUR pixel tiles (8 accumulator
//    tiles, 128 VGPRs): one weight operand feeds 4 MFMAs (32 B/clk/CU), and the B operands cost 64 B/clk/CU of LDS;
//  * the partial P tiles of a band's 1024 / WM pixels x WM wavefronts no longer fit the dead h1 region at once: the transposed
//    l_last + gather run in two halves of the pixel tiles.
//
// Replaces (reference, /root/reference): layers.py:251-375, :452-498, :555-613, :651-674 at hps.width > 32, with the coupling-CNN
// convolutions in half precision (BASELINE configs[4] names that mode for width 4; no reference counterpart).
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
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4hh __attribute__((ext_vector_type(4)));
typedef _Float16 v2hh __attribute__((ext_vector_type(2)));

constexpr int GT = 512;          // threads per workgroup
constexpr int GW = GT / 64;      // wavefronts
constexpr int PSTR = 44;         // floats per pixel of a partial P tile (as NF7_P_STRIDE)

// relu(a), relu(b) rounded to half, packed in one dword (v_cvt_pk_f16_f32 + v_pk_max_f16)
__device__ __forceinline__ uint32_t relu_pack_h2(float a, float b)
{
    const v2hh h = __builtin_elementwise_max(v2hh{(_Float16)a, (_Float16)b}, v2hh{0, 0});
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ v8h as_v8h(uint4 q) { return __builtin_bit_cast(v8h, q); }
__device__ __forceinline__ uint4 ldg4u(const float *p) { return *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

//   WP      padded coupling width: 64, 128, 256, 512
//   PHILOX  input = in-kernel Philox/Box-Muller draw
//   OWN     pixels per thread: 2 (patches <= 1024 pixels), 4 (<= 2048) or 8 (<= 4096: 64x64)
template <int WP, bool PHILOX, int OWN>
__global__ __launch_bounds__(GT) void nf_gemm16_kernel(const NfProgram prog, const NfLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MT = WP / 32;                // channel tiles
    constexpr int NB = NF8_BAND_HALVES / WP;   // pixels per band
    constexpr int NT = NB / 32;                // pixel tiles per band
    constexpr int WM = MT / 2;                 // wavefronts along the channel axis (2 tiles each)
    constexpr int WN = GW / WM;                // wavefronts along the pixel axis (4 tiles each)
    constexpr int KS = WP / 16;                // K steps of l_2
    constexpr int NBH = NB / 2;                // pixels of one half of the P stage
    static_assert(MT * NT == 64 && WM * WN == GW && NT == 4 * WN && KS % 2 == 0, "tile split");
    static_assert(WM * NBH * PSTR <= NF8_BAND_HALVES / 2, "the partial P tiles reuse the h1 region");
    const int H = a.H, W = a.W, HW = H * W;
    const int Wp = W + 2;
    const int PL = ((H + 2) * Wp + 3) & ~3;            // the z0 tile: one half2 per pixel
    uint32_t *const h1 = reinterpret_cast<uint32_t *>(smem);           // [KS][2][NB][4] dwords; later the partial P tiles
    uint32_t *const z0h = h1 + NF8_BAND_HALVES / 2;                    // [PL] half2
    float *const red = smem + NF8_BAND_HALVES / 2 + PL;                // [3][GW]

    const int t = threadIdx.x;
    const int wv = t >> 6, lane = t & 63, n = lane & 31, g = lane >> 5;
    const int wm = wv / WN, wn = wv % WN;
    int toff[4];   // l_1: z0-tile offsets of the taps 4g .. 4g+3 this lane half contributes
#pragma unroll
    for (int q = 0; q < 4; ++q) toff[q] = ((4 * g + q) / 3) * Wp + (4 * g + q) % 3;

    for (int i = t; i < PL; i += GT) z0h[i] = 0u;
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
                const float *const img = a.params + prog.ops[op].off + NF8_CPL_IMG;
                // ---- publish the pass-through half (rounded to half: a CNN input) ----
#pragma unroll
                for (int m = 0; m < OWN; ++m)
                    if (act[m]) {
                        const v2hh zh = {(_Float16)z[m][0], (_Float16)z[m][1]};
                        z0h[(pr[m] + 1) * Wp + pc[m] + 1] = __builtin_bit_cast(uint32_t, zh);
                    }
                float o[OWN][4];
#pragma unroll
                for (int m = 0; m < OWN; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[m][j] = 0.0f;
                __syncthreads();

                for (int band = 0; band < n_bands; ++band) {
                    const int p0 = band * NB;
                    // ---- l_1: 64 tiles of relu(W1 z0 + b1) -> half, 8 per wavefront, into LDS in B-operand order ----
#pragma unroll 1
                    for (int i = 0; i < 8; ++i) {
                        const int tt = wv * 8 + i, m = tt / NT, nt = tt % NT;
                        int p = p0 + 32 * nt + n;
                        p = p < HW ? p : HW - 1;   // columns past the patch: never gathered
                        const int r = p / W, c = p - r * W;
                        const uint32_t *zb = z0h + r * Wp + c;   // tap (di,dj) at + di*Wp + dj
                        v16f d;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = ldg4(img + nf8_img_B1(WP) + m * 32 + g * 16 + 4 * q);
                            d[4 * q + 0] = bb.x; d[4 * q + 1] = bb.y; d[4 * q + 2] = bb.z; d[4 * q + 3] = bb.w;
                        }
                        const uint4 a0 = ldg4u(img + nf8_img_A1H(WP) + ((m * 2 + 0) * 64 + lane) * 4);
                        const uint4 a1 = ldg4u(img + nf8_img_A1H(WP) + ((m * 2 + 1) * 64 + lane) * 4);
                        const uint4 b0 = make_uint4(zb[toff[0]], zb[toff[1]], zb[toff[2]], zb[toff[3]]);
                        const uint4 b1 = make_uint4(g == 0 ? zb[2 * Wp + 2] : 0u, 0u, 0u, 0u);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(a0), as_v8h(b0), d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(a1), as_v8h(b1), d, 0, 0, 0);
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2)
                            *reinterpret_cast<uint4 *>(h1 + ((size_t)(((m * 2 + m2) * 2 + g) * NB) + 32 * nt + n) * 4) =
                                make_uint4(relu_pack_h2(d[8 * m2 + 0], d[8 * m2 + 1]), relu_pack_h2(d[8 * m2 + 2], d[8 * m2 + 3]),
                                           relu_pack_h2(d[8 * m2 + 4], d[8 * m2 + 5]), relu_pack_h2(d[8 * m2 + 6], d[8 * m2 + 7]));
                    }
                    __syncthreads();

                    // ---- l_2: 2 x 4 accumulator tiles per wavefront over the whole K; weights streamed from L2 ----
                    v16f acc[2][4];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = ldg4(img + nf8_img_B2(WP) + (2 * wm + mi) * 32 + g * 16 + 4 * q);
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni) {
                                acc[mi][ni][4 * q + 0] = bb.x; acc[mi][ni][4 * q + 1] = bb.y;
                                acc[mi][ni][4 * q + 2] = bb.z; acc[mi][ni][4 * q + 3] = bb.w;
                            }
                        }
                    }
                    {
                        const float *ap0 = img + nf8_img_A2H(WP) + ((size_t)(2 * wm + 0) * KS * 64 + lane) * 4;
                        const float *ap1 = img + nf8_img_A2H(WP) + ((size_t)(2 * wm + 1) * KS * 64 + lane) * 4;
                        const uint32_t *bp = h1 + ((size_t)(g * NB) + 32 * (4 * wn) + n) * 4;
                        // software pipeline over K steps, two operand sets in ping-pong; the scheduling barriers keep the loads of
                        // step ks + 1 in front of the 8 MFMAs of step ks (nf_gemm.hip)
                        uint4 xa0 = ldg4u(ap0), xa1 = ldg4u(ap1), xb[4], ya0, ya1, yb[4];
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) xb[ni] = *reinterpret_cast<const uint4 *>(bp + 32 * ni * 4);
#define NF_GEMM16_STEP(A0, A1, B)                                                                                              \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                           \
    {                                                                                                                          \
        acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(A0), as_v8h(B[ni]), acc[0][ni], 0, 0, 0);                   \
        acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(A1), as_v8h(B[ni]), acc[1][ni], 0, 0, 0);                   \
    }
#pragma unroll 1
                        for (int ks = 0; ks < KS; ks += 2) {
                            const int k1 = ks + 1, k2 = ks + 2 < KS ? ks + 2 : KS - 1;
                            ya0 = ldg4u(ap0 + (size_t)k1 * 256);
                            ya1 = ldg4u(ap1 + (size_t)k1 * 256);
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni) yb[ni] = *reinterpret_cast<const uint4 *>(bp + ((size_t)k1 * 2 * NB + 32 * ni) * 4);
                            __builtin_amdgcn_sched_barrier(0);
                            NF_GEMM16_STEP(xa0, xa1, xb)
                            __builtin_amdgcn_sched_barrier(0);
                            xa0 = ldg4u(ap0 + (size_t)k2 * 256);
                            xa1 = ldg4u(ap1 + (size_t)k2 * 256);
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni) xb[ni] = *reinterpret_cast<const uint4 *>(bp + ((size_t)k2 * 2 * NB + 32 * ni) * 4);
                            __builtin_amdgcn_sched_barrier(0);
                            NF_GEMM16_STEP(ya0, ya1, yb)
                            __builtin_amdgcn_sched_barrier(0);
                        }
#undef NF_GEMM16_STEP
                    }
                    float *const pp = reinterpret_cast<float *>(h1);   // [WM][NBH][PSTR] partial P tiles of one half
                    // ---- P = W3^T relu(h2) + gather, pixel tiles {0, 1} then {2, 3} of this wavefront ----
#pragma unroll
                    for (int hN = 0; hN < 2; ++hN) {
                        v16f pa[2];
                        v4f p8[2];
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj) {
#pragma unroll
                            for (int v = 0; v < 16; ++v) pa[nj][v] = 0.0f;
                            p8[nj] = v4f{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                            for (int m2 = 0; m2 < 2; ++m2) {
                                const uint4 w0 = ldg4u(img + nf8_img_A3H(WP) + (((2 * wm + mi) * 2 + m2) * 64 + lane) * 4);
                                const uint2 c0 = *reinterpret_cast<const uint2 *>(img + nf8_img_A3CH(WP) + (((2 * wm + mi) * 4 + 2 * m2 + 0) * 8 + g * 4 + (lane & 3)) * 2);
                                const uint2 c1 = *reinterpret_cast<const uint2 *>(img + nf8_img_A3CH(WP) + (((2 * wm + mi) * 4 + 2 * m2 + 1) * 8 + g * 4 + (lane & 3)) * 2);
#pragma unroll
                                for (int nj = 0; nj < 2; ++nj) {
                                    const v16f &e = acc[mi][2 * hN + nj];
                                    const uint32_t q0 = relu_pack_h2(e[8 * m2 + 0], e[8 * m2 + 1]), q1 = relu_pack_h2(e[8 * m2 + 2], e[8 * m2 + 3]);
                                    const uint32_t q2 = relu_pack_h2(e[8 * m2 + 4], e[8 * m2 + 5]), q3 = relu_pack_h2(e[8 * m2 + 6], e[8 * m2 + 7]);
                                    pa[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(w0), as_v8h(make_uint4(q0, q1, q2, q3)), pa[nj], 0, 0, 0);
                                    p8[nj] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4hh, c0), __builtin_bit_cast(v4hh, make_uint2(q0, q1)), p8[nj], 0, 0, 0);
                                    p8[nj] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4hh, c1), __builtin_bit_cast(v4hh, make_uint2(q2, q3)), p8[nj], 0, 0, 0);
                                }
                            }
                        }
                        __syncthreads();   // first half: every wavefront is done with h1; second: with the first half's records

                        // per pixel [tap 0..7][j] (register group a of lane half g holds tap 2 a + g), then tap 8 of lane half 0 / 1
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj) {
                            float *dst = pp + ((size_t)(wm * NBH + 32 * (2 * wn + nj) + n)) * PSTR;
#pragma unroll
                            for (int aa = 0; aa < 4; ++aa)
                                *reinterpret_cast<float4 *>(dst + (2 * aa + g) * 4) =
                                    make_float4(pa[nj][4 * aa + 0], pa[nj][4 * aa + 1], pa[nj][4 * aa + 2], pa[nj][4 * aa + 3]);
                            *reinterpret_cast<float4 *>(dst + 32 + 4 * g) = make_float4(p8[nj][0], p8[nj][1], p8[nj][2], p8[nj][3]);
                        }
                        __syncthreads();

                        // gather: the taps of this half's pixels that fall on the output pixels this thread owns
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
                                    const int nt = src >> 5;                      // pixel tile nt belongs to wavefront column nt / 4,
                                    if (((nt >> 1) & 1) != hN) continue;          // ... and to its half (nt / 2) & 1
                                    const int rec = 32 * (2 * (nt >> 2) + (nt & 1)) + (src & 31);
#pragma unroll
                                    for (int k = 0; k < WM; ++k) {
                                        const float *rp = pp + ((size_t)(k * NBH + rec)) * PSTR;
                                        float4 v = *reinterpret_cast<const float4 *>(rp + (di * 3 + dj) * 4);
                                        if (di * 3 + dj == 8) {   // tap 8: the two lane halves' partial sums
                                            const float4 u = *reinterpret_cast<const float4 *>(rp + 36);
                                            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                                        }
                                        o[m][0] += v.x; o[m][1] += v.y; o[m][2] += v.z; o[m][3] += v.w;
                                    }
                                }
                            }
                        }
                    }
                    __syncthreads();   // the next band's l_1 overwrites the region
                }

                // ---- finish the coupling on the owned pixels ----
                gemm_finish_coupling<OWN, true>(type, a.params + prog.ops[op].off + NF8_CPL_E, P[NF8_CPL_S + 1], P[NF8_CPL_S + 2], T, pr, pc, act, o, z, ld2);
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
// Variant B: pixel tiles, not channel tiles, per wavefront.  Every wavefront owns ONE tile of 32 pixels per round (8 wavefronts =
// 256 pixels) and computes ALL channels for it: its h1 never leaves its registers (WP / 4 dwords of packed halves per lane — 128
// VGPRs at width 512), the l_2 accumulators of an output tile are consumed by the transposed l_last the moment they are complete
// (no cross-wavefront partial P sums), and the WEIGHTS come through LDS: one slab per output tile (nf_gemm_layout.h, NF9_*: the
// l_2 rows of the tile, its l_last columns, its bias), ALL slabs of a coupling staged once per patch and resident for its 4
// rounds — which is what limits this variant to widths <= 128 (42 KiB of slabs).  The weights cross L2 -> CU once per patch
// and coupling instead of once per band, and the operand every wavefront needs sits in LDS (1 KiB per MFMA and wavefront =
// 128 of the 256 B/clk ds_read_b128 delivers).  A streamed version for 256 / 512 (slabs double-buffered behind the previous
// tile's MFMAs, one barrier per tile) was measured below variant A (599 against 795 TFLOP/s at 512) and is not kept.
template <int WP, bool PHILOX, int OWN>
__global__ __launch_bounds__(GT) void nf_gemm16b_kernel(const NfProgram prog, const NfLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MT = WP / 32;                // channel tiles
    constexpr int KS = WP / 16;                // K steps of l_2
    constexpr int RND = 32 * GW;               // pixels per round
    constexpr int SLAB = nf9_slab_dwords(WP);
    static_assert(MT * SLAB * 4 <= 64 * 1024, "variant B keeps every slab of the coupling resident in LDS (widths <= 128)");
    constexpr int NBUF = MT;
    const int H = a.H, W = a.W, HW = H * W;
    const int Wp = W + 2;
    const int PL = ((H + 2) * Wp + 3) & ~3;            // the z0 tile: one half2 per pixel
    uint32_t *const wb = reinterpret_cast<uint32_t *>(smem);            // [NBUF][SLAB] weight slabs
    float *const prec = smem + NBUF * SLAB;                             // [RND][PSTR] P records of the round
    uint32_t *const z0h = reinterpret_cast<uint32_t *>(prec + RND * PSTR);   // [PL] half2
    float *const red = reinterpret_cast<float *>(z0h + PL);            // [3][GW]

    const int t = threadIdx.x;
    const int wv = t >> 6, lane = t & 63, n = lane & 31, g = lane >> 5;
    int toff[4];   // l_1: z0-tile offsets of the taps 4g .. 4g+3 this lane half contributes
#pragma unroll
    for (int q = 0; q < 4; ++q) toff[q] = ((4 * g + q) / 3) * Wp + (4 * g + q) % 3;

    for (int i = t; i < PL; i += GT) z0h[i] = 0u;
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
                const float *const img = a.params + prog.ops[op].off + NF8_CPL_IMG;
                const float *const slabs = img + nf9_img_SLAB(WP);
                // ---- publish the pass-through half (rounded to half: a CNN input) ----
#pragma unroll
                for (int m = 0; m < OWN; ++m)
                    if (act[m]) {
                        const v2hh zh = {(_Float16)z[m][0], (_Float16)z[m][1]};
                        z0h[(pr[m] + 1) * Wp + pc[m] + 1] = __builtin_bit_cast(uint32_t, zh);
                    }
                float o[OWN][4];
#pragma unroll
                for (int m = 0; m < OWN; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[m][j] = 0.0f;
                // this coupling's slabs
                for (int i = t; i < MT * (SLAB / 4); i += GT)
                    reinterpret_cast<uint4 *>(wb)[i] = reinterpret_cast<const uint4 *>(slabs)[i];
                __syncthreads();

                for (int rnd = 0; rnd < n_rounds; ++rnd) {
                    const int p0 = rnd * RND;
                    // ---- l_1: relu(W1 z0 + b1) -> half for this wavefront's 32 pixels, all channels, into registers ----
                    uint32_t hr[MT][8];
                    {
                        int p = p0 + 32 * wv + n;
                        p = p < HW ? p : HW - 1;   // columns past the patch: never gathered
                        const int r = p / W, c = p - r * W;
                        const uint32_t *zb = z0h + r * Wp + c;   // tap (di,dj) at + di*Wp + dj
                        const uint4 b0 = make_uint4(zb[toff[0]], zb[toff[1]], zb[toff[2]], zb[toff[3]]);
                        const uint4 b1 = make_uint4(g == 0 ? zb[2 * Wp + 2] : 0u, 0u, 0u, 0u);
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            v16f d;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 bb = ldg4(img + nf8_img_B1(WP) + m * 32 + g * 16 + 4 * q);
                                d[4 * q + 0] = bb.x; d[4 * q + 1] = bb.y; d[4 * q + 2] = bb.z; d[4 * q + 3] = bb.w;
                            }
                            const uint4 a0 = ldg4u(img + nf8_img_A1H(WP) + ((m * 2 + 0) * 64 + lane) * 4);
                            const uint4 a1 = ldg4u(img + nf8_img_A1H(WP) + ((m * 2 + 1) * 64 + lane) * 4);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(a0), as_v8h(b0), d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(a1), as_v8h(b1), d, 0, 0, 0);
#pragma unroll
                            for (int i = 0; i < 8; ++i) hr[m][i] = relu_pack_h2(d[2 * i], d[2 * i + 1]);
                            __builtin_amdgcn_sched_barrier(0);   // keep the tiles' operand loads from being hoisted on top of each other
                        }
                    }
                    // ---- per output tile: l_2 over the whole K from the slab in LDS, then its share of P = W3^T relu(h2) ----
                    v16f pa;
                    v4f p8 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < 16; ++v) pa[v] = 0.0f;
#pragma unroll 1
                    for (int m = 0; m < MT; ++m) {
                        const uint32_t *sl = wb + m * SLAB;
                        v16f acc0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = *reinterpret_cast<const float4 *>(sl + nf9_slab_B2(WP) + g * 16 + 4 * q);
                            acc0[4 * q + 0] = bb.x; acc0[4 * q + 1] = bb.y; acc0[4 * q + 2] = bb.z; acc0[4 * q + 3] = bb.w;
                        }
                        const uint4 *ap = reinterpret_cast<const uint4 *>(sl) + lane;   // K step ks at + 64 ks
                        // pairs of K steps; the LDS reads of the next pair are issued before the MFMAs of this one (one accumulator
                        // chain: the SIMD's other wavefront fills the dependent-issue gaps; a second chain costs 16 VGPRs this kernel
                        // does not have at width 512)
                        uint4 ca[2], na[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) ca[u] = ap[64 * u];
#pragma unroll
                        for (int kg = 0; kg < KS; kg += 2) {
                            if (kg + 2 < KS) {
#pragma unroll
                                for (int u = 0; u < 2; ++u) na[u] = ap[64 * (kg + 2 + u)];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int ks = kg + u;
                                const uint4 bq = make_uint4(hr[ks >> 1][4 * (ks & 1) + 0], hr[ks >> 1][4 * (ks & 1) + 1],
                                                            hr[ks >> 1][4 * (ks & 1) + 2], hr[ks >> 1][4 * (ks & 1) + 3]);
                                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(ca[u]), as_v8h(bq), acc0, 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < 2; ++u) ca[u] = na[u];
                        }
                        // h2 tile m complete: relu, round to half, and straight into P (taps 0 .. 7: one 32-row tile; tap 8: 4x4x4)
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) {
                            const uint4 w0 = *reinterpret_cast<const uint4 *>(sl + nf9_slab_A3H(WP) + (m2 * 64 + lane) * 4);
                            const uint2 c0 = *reinterpret_cast<const uint2 *>(sl + nf9_slab_A3CH(WP) + ((2 * m2 + 0) * 8 + g * 4 + (lane & 3)) * 2);
                            const uint2 c1 = *reinterpret_cast<const uint2 *>(sl + nf9_slab_A3CH(WP) + ((2 * m2 + 1) * 8 + g * 4 + (lane & 3)) * 2);
                            uint32_t q[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                q[i] = relu_pack_h2(acc0[8 * m2 + 2 * i], acc0[8 * m2 + 2 * i + 1]);
                            pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_v8h(w0), as_v8h(make_uint4(q[0], q[1], q[2], q[3])), pa, 0, 0, 0);
                            p8 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4hh, c0), __builtin_bit_cast(v4hh, make_uint2(q[0], q[1])), p8, 0, 0, 0);
                            p8 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4hh, c1), __builtin_bit_cast(v4hh, make_uint2(q[2], q[3])), p8, 0, 0, 0);
                        }
                    }
                    // ---- P records of the round: per pixel [tap 0..7][j] (register group a of lane half g = tap 2 a + g), tap 8 of half 0 / 1 ----
                    {
                        float *dst = prec + (size_t)(32 * wv + n) * PSTR;
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
                                const float *rp = prec + (size_t)src * PSTR;
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
                gemm_finish_coupling<OWN, true>(type, a.params + prog.ops[op].off + NF8_CPL_E, P[NF8_CPL_S + 1], P[NF8_CPL_S + 2], T, pr, pc, act, o, z, ld2);
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

size_t gemm16b_lds_bytes(int wp, int H, int W)
{
    const int Wp = W + 2, PL = ((H + 2) * Wp + 3) & ~3, MT = wp / 32, SLAB = nf9_slab_dwords(wp);
    return ((size_t)MT * SLAB + (size_t)32 * GW * PSTR + (size_t)PL + 3 * GW + 8) * sizeof(float);
}

template <int WP, bool PHILOX, int OWN>
hipError_t launch_gemm16b(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    static std::atomic<size_t> lds_set[16];
    return gemm_launch_per_cu<GT>(&nf_gemm16b_kernel<WP, PHILOX, OWN>, gemm16b_lds_bytes(WP, a.H, a.W), lds_set, prog, a, n_cu, device, stream);
}

template <int WP, bool PHILOX>
hipError_t dispatch_own16b(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    return gemm_by_own<GT>(a.H * a.W, [&](auto own) { return launch_gemm16b<WP, PHILOX, decltype(own)::value>(prog, a, n_cu, device, stream); });
}

template <bool PHILOX>
hipError_t dispatch_gemm16b(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    switch (prog.width) {
    case 64: return dispatch_own16b<64, PHILOX>(prog, a, n_cu, device, stream);
    case 128: return dispatch_own16b<128, PHILOX>(prog, a, n_cu, device, stream);
    }
    return hipErrorInvalidValue;
}

size_t gemm16_lds_bytes(int H, int W)
{
    const int Wp = W + 2, PL = ((H + 2) * Wp + 3) & ~3;
    return ((size_t)NF8_BAND_HALVES / 2 + (size_t)PL + 3 * GW + 8) * sizeof(float);
}

template <int WP, bool PHILOX, int OWN>
hipError_t launch_gemm16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    static std::atomic<size_t> lds_set[16];
    return gemm_launch_per_cu<GT>(&nf_gemm16_kernel<WP, PHILOX, OWN>, gemm16_lds_bytes(a.H, a.W), lds_set, prog, a, n_cu, device, stream);
}

template <int WP, bool PHILOX>
hipError_t dispatch_own16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    return gemm_by_own<GT>(a.H * a.W, [&](auto own) { return launch_gemm16<WP, PHILOX, decltype(own)::value>(prog, a, n_cu, device, stream); });
}

template <bool PHILOX>
hipError_t dispatch_gemm16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    switch (prog.width) {
    case 64: return dispatch_own16<64, PHILOX>(prog, a, n_cu, device, stream);
    case 128: return dispatch_own16<128, PHILOX>(prog, a, n_cu, device, stream);
    case 256: return dispatch_own16<256, PHILOX>(prog, a, n_cu, device, stream);
    case 512: return dispatch_own16<512, PHILOX>(prog, a, n_cu, device, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// variant B: programs in the NF9 layout
hipError_t nf_launch_gemm16b(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (a.H < 1 || a.W < 1 || a.H * a.W > NF7_MAX_PIXELS || gemm16b_lds_bytes(prog.width, a.H, a.W) > 160 * 1024) return hipErrorInvalidValue;
    if (a.flags & NF_K_PHILOX_IN) return dispatch_gemm16b<true>(prog, a, n_cu, device, stream);
    return dispatch_gemm16b<false>(prog, a, n_cu, device, stream);
}

// entry point used by nf_host.hip: programs in the NF8 layout (NF_CFG_FP16_CNN, coupling width padded to 64 / 128 / 256 / 512)
hipError_t nf_launch_gemm16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (a.H < 1 || a.W < 1 || a.H * a.W > NF7_MAX_PIXELS || gemm16_lds_bytes(a.H, a.W) > 160 * 1024) return hipErrorInvalidValue;
    if (a.flags & NF_K_PHILOX_IN) return dispatch_gemm16<true>(prog, a, n_cu, device, stream);
    return dispatch_gemm16<false>(prog, a, n_cu, device, stream);
}
