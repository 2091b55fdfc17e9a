// Fused Noise Flow stack for coupling width 16 on the f32 matrix cores of gfx950: the nf_wide.hip design
// (channels on the M axis, pixels on N, layers chained through the accumulator registers, l_last evaluated
// transposed with a strip-local shift-add) re-cut for v_mfma_f32_16x16x4_f32 — exact fp32, 64 FLOP/clk/SIMD,
// 4 D registers, K = 4 per instruction:
//
//  * a TILE is 16 consecutive pixels of one image row (lane n = lane & 15); the four lane groups g = lane >> 4 are the
//    instruction's four K slices.  D register v of group g holds channel c(v, g) = 4 g + v, and K step s of the next
//    layer consumes D[s] with the four slices standing for channels c(s, 0..3): l_1 (5 steps: 18 tap-channels padded
//    to 20) -> ReLU -> l_2 (4) -> ReLU -> P (2 chains x 4) never leave the registers.
//  * P rows: chain A = taps (0,0) (1,0) (2,0) (0,1) in groups 0..3, chain B = (0,2) (1,2) (2,2) (2,1); the centre tap on
//    v_mfma_f32_4x4x1.  A tile is exactly one DPP row: row_shr:1 / row_shl:1 with row_mask 0x7 shift the three
//    column-neighbour taps, zero-fill the tile ends and leave group 3 (a no-shift tap) untouched — one instruction.
//  * a wavefront owns a strip of TPW (8 or 16) consecutive rows of one 16-pixel column block; vertical sums stay in
//    registers, strip boundaries and the column seams between blocks go through LDS.  Lane group g owns the rows
//    row0 + 4 q + g: a v_permlane32_swap + v_permlane16_swap reduce-scatter finishes four tiles at a time.
//
// Replaces (reference): layers.py:251-375, 452-498, 555-613, 651-674 at hps.width = 16.
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>
#include "../../include/noiseflow_hip.h"   // NF_SUMS_SLOTS / NF_SUMS_STRIDE
#include "nf_device.h"
#include "nf_dev_util.h"
#include "nf_gemm_common.h"   // GemmTile: where a patch or, NF_K_TILED, a tile of an image sits

namespace {

#define DPP_ROW_SHR1 0x111
#define DPP_ROW_SHL1 0x101

// groups 0..2: the value of the lane one pixel to the left / right inside the 16-lane tile, 0 at the tile end;
// group 3: unchanged
__device__ __forceinline__ float shr_g012(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), DPP_ROW_SHR1, 0x7, 0xf, true));
}
__device__ __forceinline__ float shl_g012(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), DPP_ROW_SHL1, 0x7, 0xf, true));
}

// Reduce-scatter of four tiles' per-group partial sums: group g ends with the total of tile g.
__device__ __forceinline__ float quad_sums(float x0, float x1, float x2, float x3)
{
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x2), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x1), __float_as_uint(x3), false, false);
    const float r1 = __uint_as_float(a[0]) + __uint_as_float(a[1]);   // low half: tile 0 over groups {g, g+2}; high half: tile 2
    const float r2 = __uint_as_float(b[0]) + __uint_as_float(b[1]);   // low half: tile 1;                     high half: tile 3
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r2), false, false);
    return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}

//   THREADS  64 x number of strips;  TPW  rows per strip (8 or 16);  PHILOX  in-kernel eps
// tuning knobs (A/B builds): occupancy target of the 512-thread geometry; re-read the A operands per tile
#ifndef NF_W16_WPE
#define NF_W16_WPE 1
#endif
#ifndef NF_W16_RELOAD
#define NF_W16_RELOAD 0
#endif
#ifndef NF_W16_PLANE_PAD
#define NF_W16_PLANE_PAD 1
#endif
__host__ __device__ inline int nf_w16_plane(int px)
{
    int pl = (px + 3) & ~3;
    if (NF_W16_PLANE_PAD) pl += (16 - (pl & 31)) & 31;     // pl = 16 (mod 32)
    return pl;
}

template <int THREADS, int TPW, bool PHILOX>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 512 ? NF_W16_WPE : 1))) void nf_wide16_kernel(const NfProgram prog, const NfLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = THREADS / 64;
    constexpr int OWN = TPW / 4;
    const int H = a.H, W = a.W, HW = H * W;
    const int Wp = W + 2;
    // one channel plane of the z0 tile.  Lane groups 0 / 1 (and 2 / 3) of an l_1 operand read the SAME tap in the two planes, 16
    // consecutive words each, in one 32-lane pass of ds_read_b32: the planes sit 16 banks (mod 32) apart so that the two groups
    // do not meet (NF_W16_PLANE_PAD=0: the planes back to back — 2-way conflicts on 12 of 16 banks at 32x32)
    const int PL = nf_w16_plane((H + 2) * Wp);
    const int TC = (W + 15) >> 4;                        // 16-pixel column blocks per image row
    float *const z0s = smem;                             // [2][PL]
    float *const wbuf = z0s + 2 * PL;                    // [NF6_IMG_SIZE] weights of the current coupling
    float *const exch = wbuf + NF6_IMG_SIZE;             // [NW][2 dir][2 groups][16][4] strip-boundary partials
    float *const side = exch + NW * 256;                 // [2][H+2][TC][3][4] column-seam taps
    float *const red = side + 2 * (H + 2) * TC * 12;     // [3][NW] reduction scratch

    const int t = threadIdx.x;
    const int w = t >> 6, lane = t & 63, n = lane & 15, g = lane >> 4;
    const int rg = w / TC, tc = w - rg * TC;
    const int row0 = rg * TPW, c = tc * 16 + n;
    const bool col_on = c < W;
    const bool strip_on = row0 < H;
    // 0/1 multipliers (see the combine below): groups 0..2 hold column-neighbour taps, group 3 in-place taps
    const float m0 = g == 0 ? 1.f : 0.f, m1 = g == 1 ? 1.f : 0.f, m2 = g == 2 ? 1.f : 0.f, m3 = g == 3 ? 1.f : 0.f;
    const float mrg = g == 3 ? 1.f : (c + 1 < W ? 1.f : 0.f);   // a right neighbour inside the image (the tile end is zero-filled by the DPP shift)
    const float wa_dn = m0 + m3, wb_dn = mrg * m0, wa_c = m1, wb_c = mrg * m1, wa_up = m2, wb_up = mrg * m2 + m3;
    int toff[5];   // l_1: z0-tile offset of the (tap, channel) this lane group contributes to K step s
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int kk = 4 * s + g, tap = kk >> 1;
        toff[s] = kk < 18 ? (kk & 1) * PL + (tap / 3) * Wp + tap % 3 : 0;
    }

    for (int i = t; i < 2 * PL + NF6_IMG_SIZE + NW * 256 + 2 * (H + 2) * TC * 12; i += THREADS) smem[i] = 0.0f;
    __syncthreads();

    const int n_ops = prog.n_ops;
    double acc_nll = 0.0, acc_sd = 0.0;   // thread 0 only

    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        const GemmTile T = gemm_tile(a, b, H, W);      // the patch on its own, or (NF_K_TILED) a tile of an image: nf_gemm_common.h
        const size_t patch_off = T.patch_off;

        float z[OWN][4];
#pragma unroll
        for (int m = 0; m < OWN; ++m) {
            const int r = row0 + 4 * m + g;
            const bool act = r < H && col_on && strip_on;
            const int gi = T.gi(act, r, c);
            if (PHILOX) {
                philox_normal4(a.seed, a.patch_base + T.patch_id, (uint32_t)gi, NF_STREAM_SAMP, z[m]);
#pragma unroll
                for (int q = 0; q < 4; ++q) z[m][q] *= a.in_scale;
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (act) v = reinterpret_cast<const float4 *>(a.in + patch_off)[gi];
                z[m][0] = v.x * a.in_scale;
                z[m][1] = v.y * a.in_scale;
                z[m][2] = v.z * a.in_scale;
                z[m][3] = v.w * a.in_scale;
            }
        }

        float ld = 0.0f, ld2 = 0.0f;   // natural-log / log2 parts of this lane's log-det share

        for (int op = 0; op < n_ops; ++op) {
            const int type = prog.ops[op].type;
            const cfloat_p P = (cfloat_p)(a.params + prog.ops[op].off);   // wave-uniform, scalar loads

            if (type == NF_OP_MIX) {
                float mm[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) mm[i] = P[i];
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float s = z[m][0] * mm[j];
                        s = fmaf(z[m][1], mm[4 + j], s);
                        s = fmaf(z[m][2], mm[8 + j], s);
                        s = fmaf(z[m][3], mm[12 + j], s);
                        o[j] = s;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[m][j] = o[j];
                }
            } else if (type == NF_OP_COUPLING_FWD || type == NF_OP_COUPLING_REV) {
                // ---- phase A: publish the pass-through half, stage this coupling's weights ----
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int r = row0 + 4 * m + g;
                    if (r < H && col_on && strip_on) {
                        z0s[(r + 1) * Wp + c + 1] = z[m][0];
                        z0s[PL + (r + 1) * Wp + c + 1] = z[m][1];
                    }
                }
                {
                    const float4 *src = reinterpret_cast<const float4 *>(a.params + prog.ops[op].off + NF4_CPL_IMG);
                    float4 *dst = reinterpret_cast<float4 *>(wbuf);
                    for (int i = t; i < NF6_IMG_SIZE / 4; i += THREADS) dst[i] = src[i];
                }
                __syncthreads();

                // ---- phase B: the CNN on the matrix cores, strip-local shift-add ----
                const float4 *const wb4 = reinterpret_cast<const float4 *>(wbuf);
                float cp[TPW][4];
#pragma unroll
                for (int k = 0; k < TPW; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) cp[k][j] = 0.0f;
                if (strip_on) {
#pragma unroll
                    for (int k = 0; k < TPW; ++k) {
                        const int r = row0 + k;
                        if (r >= H) continue;   // wave-uniform
                        // 16-wave workgroups leave 128 VGPRs: re-read the 33 A operands per tile instead of holding them
                        if constexpr (TPW == 16 || NF_W16_RELOAD) asm volatile("" ::: "memory");
                    const float4 a1 = wb4[NF6_IMG_A1 / 4 + lane];        // l_1 steps 0..3
                    const float a1e = wbuf[NF6_IMG_A1 + 256 + lane];     // l_1 step 4
                    const float4 bb1 = wb4[NF6_IMG_B1 / 4 + g], bb2 = wb4[NF6_IMG_B2 / 4 + g];
                    const float4 a2 = wb4[NF6_IMG_A2 / 4 + lane];
                    const float4 a3a = wb4[NF6_IMG_A3A / 4 + lane], a3b = wb4[NF6_IMG_A3B / 4 + lane];
                    const float4 a3c = wb4[NF6_IMG_A3C / 4 + g * 4 + (lane & 3)];
                    const float a1s[5] = {a1.x, a1.y, a1.z, a1.w, a1e};
                    const float a2s[4] = {a2.x, a2.y, a2.z, a2.w};
                    const float a3as[4] = {a3a.x, a3a.y, a3a.z, a3a.w}, a3bs[4] = {a3b.x, a3b.y, a3b.z, a3b.w};
                    const float a3cs[4] = {a3c.x, a3c.y, a3c.z, a3c.w};
                        const float *zb = z0s + r * Wp + c;
                        v4f d = {bb1.x, bb1.y, bb1.z, bb1.w};
#pragma unroll
                        for (int s = 0; s < 5; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1s[s], zb[toff[s]], d, 0, 0, 0);
                        v4f e = {bb2.x, bb2.y, bb2.z, bb2.w};
#pragma unroll
                        for (int s = 0; s < 4; ++s) e = __builtin_amdgcn_mfma_f32_16x16x4f32(a2s[s], nf_relu(d[s]), e, 0, 0, 0);
                        v4f pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f}, pc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const float h = nf_relu(e[s]);
                            pa = __builtin_amdgcn_mfma_f32_16x16x4f32(a3as[s], h, pa, 0, 0, 0);
                            pb = __builtin_amdgcn_mfma_f32_16x16x4f32(a3bs[s], h, pb, 0, 0, 0);
                            pc = __builtin_amdgcn_mfma_f32_4x4x1f32(a3cs[s], h, pc, 0, 0, 0);
                        }
                        // column seams: the raw taps of the tile's edge pixels that belong to the neighbour block
                        if (TC > 1 && g < 3) {
                            if (n == 15 && tc + 1 < TC)
                                *reinterpret_cast<float4 *>(side + (((0 * (H + 2) + r + 1) * TC + tc) * 3 + g) * 4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
                            if (n == 0 && tc > 0)
                                *reinterpret_cast<float4 *>(side + (((1 * (H + 2) + r + 1) * TC + tc) * 3 + g) * 4) = make_float4(pb[0], pb[1], pb[2], pb[3]);
                        }
                        // Horizontal part of the shift-add.  Group g' of chain A holds tap (g',0) [g' < 3] | (0,1) [g' = 3], of
                        // chain B tap (g',2) | (2,1):  groups 0 / 1 / 2 carry R[.][di = 0 / 1 / 2] (rows below / same / above),
                        // group 3 its two in-place taps, which go below (chain A) and above (chain B).
                        float dn[4], up[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float sa = shr_g012(pa[j]), sb = shl_g012(pb[j]);
                            dn[j] = fmaf(sa, wa_dn, sb * wb_dn);
                            up[j] = fmaf(sa, wa_up, sb * wb_up);
                            cp[k][j] += fmaf(sa, wa_c, fmaf(sb, wb_c, pc[j]));
                        }
                        if (k + 1 < TPW) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) cp[k + 1][j] += dn[j];
                        }
                        if (k > 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) cp[k - 1][j] += up[j];
                        }
                        // strip boundaries: the contributing groups park their partials for the neighbour strip
                        if (k == 0 && row0 > 0 && g >= 2)
                            *reinterpret_cast<float4 *>(exch + ((w * 2 + 1) * 2 + (g - 2)) * 64 + n * 4) = make_float4(up[0], up[1], up[2], up[3]);
                        if (k == TPW - 1 && row0 + TPW < H && (g == 0 || g == 3))
                            *reinterpret_cast<float4 *>(exch + ((w * 2 + 0) * 2 + (g == 3)) * 64 + n * 4) = make_float4(dn[0], dn[1], dn[2], dn[3]);
                    }
                }
                __syncthreads();

                // ---- phase C: strip-boundary rows, column seams, then each lane group finishes the rows it owns ----
                if (strip_on) {
                    if (row0 > 0 && (g == 0 || g == 3)) {      // from the last row of the strip above
                        const float4 v = *reinterpret_cast<const float4 *>(exch + (((w - TC) * 2 + 0) * 2 + (g == 3)) * 64 + n * 4);
                        cp[0][0] += v.x; cp[0][1] += v.y; cp[0][2] += v.z; cp[0][3] += v.w;
                    }
                    if (row0 + TPW < H && g >= 2) {            // from the first row of the strip below
                        const float4 v = *reinterpret_cast<const float4 *>(exch + (((w + TC) * 2 + 1) * 2 + (g - 2)) * 64 + n * 4);
                        cp[TPW - 1][0] += v.x; cp[TPW - 1][1] += v.y; cp[TPW - 1][2] += v.z; cp[TPW - 1][3] += v.w;
                    }
                    if (TC > 1 && g == 0 && ((n == 0 && tc > 0) || (n == 15 && tc + 1 < TC && c + 1 < W))) {
                        const int sd = n == 0 ? 0 : 1, tn = n == 0 ? tc - 1 : tc + 1;   // left neighbour's (di,0) taps | right neighbour's (di,2)
#pragma unroll
                        for (int k = 0; k < TPW; ++k) {
                            const int r = row0 + k;
                            if (r >= H) continue;
#pragma unroll
                            for (int di = 0; di < 3; ++di) {   // P rows r-1, r, r+1 (tile index r + di), tap row di
                                const float4 v = *reinterpret_cast<const float4 *>(side + (((sd * (H + 2) + r + di) * TC + tn) * 3 + di) * 4);
                                cp[k][0] += v.x; cp[k][1] += v.y; cp[k][2] += v.z; cp[k][3] += v.w;
                            }
                        }
                    }
                }
                const float scl = P[NF4_CPL_S + 1], m2scl = P[NF4_CPL_S + 2];
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int r = row0 + 4 * m + g;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = quad_sums(cp[4 * m][j], cp[4 * m + 1][j], cp[4 * m + 2][j], cp[4 * m + 3][j]);
                    const bool act = r < H && col_on && strip_on;
                    const int bm = T.border(r, c);
                    const float4 eb = *reinterpret_cast<const float4 *>(a.params + prog.ops[op].off + NF4_CPL_E + 4 * (act ? bm : 0));
                    o[0] += eb.x; o[1] += eb.y; o[2] += eb.z; o[3] += eb.w;
                    // raw columns are pre-scaled by 2 log2(e): t = exp2(raw') = exp(2 raw); ls*log2(e) = scl - 2 scl/(t + 1)
                    const float l0 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[2]) + 1.0f), m2scl, scl);
                    const float l1 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[3]) + 1.0f), m2scl, scl);
                    if (type == NF_OP_COUPLING_FWD) {
                        z[m][2] = fmaf(z[m][2], __builtin_amdgcn_exp2f(l0), o[0]);
                        z[m][3] = fmaf(z[m][3], __builtin_amdgcn_exp2f(l1), o[1]);
                        if (T.own(act, r, c)) ld2 += l0 + l1;
                    } else {
                        z[m][2] = (z[m][2] - o[0]) * __builtin_amdgcn_exp2f(-l0);
                        z[m][3] = (z[m][3] - o[1]) * __builtin_amdgcn_exp2f(-l1);
                    }
                }
            } else if (type == NF_OP_SDN_DIV || type == NF_OP_SDN_MUL) {
                // AffineCouplingSdnEx5: scale = sqrt(beta1*y/gain + beta2)  (cond_utils.py:238)
                const float4 *y4 = reinterpret_cast<const float4 *>(a.y + patch_off);
                const float ck1 = a.cond_a[prog.ops[op].off & 3], cb2 = a.cond_b[prog.ops[op].off & 3];
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int r = row0 + 4 * m + g;
                    const bool act = r < H && col_on && strip_on;
                    float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (act) yv = y4[T.gi(act, r, c)];
                    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = fmaf(yy[q], ck1, cb2);
                        if (type == NF_OP_SDN_DIV) {
                            z[m][q] = z[m][q] * __builtin_amdgcn_rsqf(v);
                            if (T.own(act, r, c)) ld = fmaf(-0.34657359027997264f, __builtin_amdgcn_logf(v), ld);
                        } else {
                            z[m][q] = z[m][q] * __builtin_amdgcn_sqrtf(v);
                        }
                    }
                }
            } else if (type == NF_OP_SCALE || type == NF_OP_SCALE_COND) {
                const float s = type == NF_OP_SCALE ? P[0] : a.cond_a[prog.ops[op].off & 3];
#pragma unroll
                for (int m = 0; m < OWN; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[m][q] *= s;
            }
        }

        // ---- epilogue (as nf_flow_kernel) ----
        if (a.out) {
            float4 *out4 = reinterpret_cast<float4 *>(a.out + patch_off);
#pragma unroll
            for (int m = 0; m < OWN; ++m) {
                const int r = row0 + 4 * m + g;
                const bool act = r < H && col_on && strip_on;
                if (T.own(act, r, c)) out4[T.gi(act, r, c)] = make_float4(z[m][0], z[m][1], z[m][2], z[m][3]);
            }
        }
        if (a.nll_out || a.sd_out || a.ld_out || a.sums || (T.tiled && a.tile_part)) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < OWN; ++m)
                if (T.own(row0 + 4 * m + g < H && col_on && strip_on, row0 + 4 * m + g, c)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s1 += z[m][q];
                        s2 = fmaf(z[m][q], z[m][q], s2);
                    }
                }
            float r0 = wave_sum(fmaf(ld2, 0.6931471805599453f, ld)), r1 = wave_sum(s1), r2 = wave_sum(s2);
            if (lane == 0) {
                red[w] = r0;
                red[NW + w] = r1;
                red[2 * NW + w] = r2;
            }
            __syncthreads();
            if (t == 0) {
                r0 = 0.f; r1 = 0.f; r2 = 0.f;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    r0 += red[i];
                    r1 += red[NW + i];
                    r2 += red[2 * NW + i];
                }
            }
            if (t == 0 && T.tiled) {
                // the tile's share of its image's sums (nf_tile_combine_kernel forms nll / sd / log-det per image)
                *reinterpret_cast<float4 *>(a.tile_part + (size_t)b * 4u) = make_float4(r0, r1, r2, 0.f);
            } else if (t == 0) {
                const double npx = (double)HW * 4.0;
                const double logdet = (double)r0 + a.ld_const;
                double nll = -logdet;   // prior: sum -0.5*(log 2pi + z^2)   (noise_flow_model.py:537-539)
                if (a.flags & NF_K_PRIOR) nll += 0.5 * npx * 1.8378770664093453 + 0.5 * (double)r2;
                const double mean = (double)r1 / npx;
                double var = (double)r2 / npx - mean * mean;   // noise_flow_model.py:477-478
                var = var > 0.0 ? var : 0.0;
                const double sd = (double)__builtin_amdgcn_sqrtf((float)var);
                if (a.nll_out) a.nll_out[b] = (float)nll;
                if (a.sd_out) a.sd_out[b] = (float)sd;
                if (a.ld_out) a.ld_out[b] = (float)logdet;
                acc_nll += (double)(float)nll;
                acc_sd += (double)(float)sd;
            }
            __syncthreads();   // scratch is reused by the next patch
        }
    }

    if (a.sums && t == 0 && !(a.flags & NF_K_TILED)) {
        double *sp = a.sums;
        if (a.flags & NF_K_SUMS_WIDE) sp += (size_t)(blockIdx.x & (NF_SUMS_SLOTS - 1)) * NF_SUMS_STRIDE;
        atomicAdd(&sp[0], acc_nll);
        atomicAdd(&sp[1], acc_sd);
        if (blockIdx.x == 0) atomicAdd(&sp[2], (double)a.B);
    }
}

size_t wide16_lds_bytes(int H, int W, int threads)
{
    const int Wp = W + 2, PL = nf_w16_plane((H + 2) * Wp), NW = threads / 64, TC = (W + 15) >> 4;
    size_t f = 2 * (size_t)PL + NF6_IMG_SIZE + (size_t)NW * 256 + 2 * (size_t)(H + 2) * TC * 12 + ((3 * NW + 3) & ~3);
    return f * sizeof(float);
}

template <int THREADS, int TPW, bool PHILOX>
hipError_t launch_wide16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    const size_t lds = wide16_lds_bytes(a.H, a.W, THREADS);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const void *fn = reinterpret_cast<const void *>(&nf_wide16_kernel<THREADS, TPW, PHILOX>);
    static std::atomic<uint64_t> cache{0};   // (device << 40 | lds bytes << 8 | resident workgroups per CU) of the last query
    const uint64_t key = ((uint64_t)(device & 0xff) << 40) | ((uint64_t)lds << 8);
    uint64_t cv = cache.load(std::memory_order_relaxed);
    int occ;
    if ((cv & ~(uint64_t)0xff) == key && (cv & 0xff) != 0) {
        occ = (int)(cv & 0xff);
    } else {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        occ = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, THREADS, lds);
        if (e != hipSuccess) return e;
        if (occ < 1) occ = 1;
        if (occ > 32) occ = 32;
        cache.store(key | (uint64_t)occ, std::memory_order_relaxed);
    }
    int64_t groups = (int64_t)n_cu * occ;
    if (a.B < groups) groups = a.B;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL((nf_wide16_kernel<THREADS, TPW, PHILOX>), dim3((unsigned)groups), dim3(THREADS), lds, stream, prog, a);
    return hipGetLastError();
}

template <bool PHILOX>
hipError_t dispatch_wide16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    const int TC = (a.W + 15) >> 4;
    const int s8 = ((a.H + 7) >> 3) * TC;      // strips of 8 rows
    if (s8 <= 4) return launch_wide16<256, 8, PHILOX>(prog, a, n_cu, device, stream);
    if (s8 <= 8) return launch_wide16<512, 8, PHILOX>(prog, a, n_cu, device, stream);
    if (s8 <= 16) return launch_wide16<1024, 8, PHILOX>(prog, a, n_cu, device, stream);
    return launch_wide16<1024, 16, PHILOX>(prog, a, n_cu, device, stream);   // up to 64x64: 4 x 4 strips of 16 rows
}

}  // namespace

// entry point used by nf_host.hip: programs in the NF6 layout (coupling width 16; 8 zero-padded), patches up to 64x64
hipError_t nf_launch_wide16(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (prog.width != 16 || a.H < 1 || a.W < 1 || a.H > 64 || a.W > 64) return hipErrorInvalidValue;
    if (a.flags & NF_K_PHILOX_IN) return dispatch_wide16<true>(prog, a, n_cu, device, stream);
    return dispatch_wide16<false>(prog, a, n_cu, device, stream);
}
