// Fused Noise Flow stack for the paper-scale coupling CNN (width 32) on the f32 matrix cores of gfx950.
//
// Same program, same I/O and same per-patch workgroup as nf_kernels.hip, but the three convolutions of
// every coupling CNN (layers.py:463-497: 3x3x2->w, 1x1 w->w, 3x3x(w+1)->4; 2.8 kMAC per pixel at w = 32)
// are GEMMs on v_mfma_f32_32x32x2_f32 — exact fp32, 64 FLOP/clk/SIMD:
//
//  * a TILE is 32 consecutive pixels of one image row on the instruction's N axis (lane n = lane & 31);
//    the two lane halves g = lane >> 5 are its two K slices.  Channels sit on the M axis, so the D
//    register v of lane half g holds channel c(v, g) = 8 (v >> 2) + 4 g + (v & 3) of the lane's pixel
//    and IS the B operand of K step v of the next layer: l_1 -> ReLU -> l_2 -> ReLU -> l_last chain through
//    registers with no data movement at all.  The A operands (weights) are pre-permuted on the host into
//    fetch order (nf_device.h, NF4_*) and staged in LDS once per coupling: by LDS-DMA into the other half of a double
//    buffer while the current coupling's CNN runs.
//  * l_1 reads its B operand (tap (di,dj), channel g of the pass-through half) from a zero-bordered LDS
//    tile of z0 — 9 MFMAs per tile.
//  * l_last is evaluated transposed: P[pixel][tap][j] = sum_c h2[pixel][c] W3[tap][c][j] is one 32-row
//    GEMM for the 8 off-centre taps (+ the centre tap on v_mfma_f32_4x4x1) straight from the h2 registers;
//    the output is the shift-add  o[r][c] = sum_tap P[r+di-1][c+dj-1][tap].  The horizontal shifts are
//    lane shifts inside the tile, the vertical ones stay in registers because a wavefront owns a STRIP of
//    8 consecutive rows; only strip boundaries (and the column 31|32 seam of 64-wide patches) go through
//    LDS.  h2 never exists in memory, 'SAME' zero padding falls out of the zero-filled shifts.
//
// Replaces (reference, /root/reference): layers.py:251-375 (AffineCoupling), :452-498 (real_nvp_conv_template),
// :555-613, :651-674 (conv2d / add_edge_padding / conv2d_zeros) at hps.width = 32 (job_noise_flow.sh:19).
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>
#include "../../include/noiseflow_hip.h"   // NF_SUMS_SLOTS / NF_SUMS_STRIDE
#include "nf_device.h"
#include "nf_dev_util.h"

// Instrumented build (-DNF_TIMELINE, tools/timeline_wide.py only): thread 0 of every workgroup stamps the 100 MHz s_memrealtime
// counter at the phase boundaries of its MIDDLE patch into NfLaunch::sd_out, reinterpreted as int64[grid][64] (sd_z is not written
// in this build): 0 patch start, 1 inputs in registers, per coupling c: 2+4c weights staged (after the barrier of phase A), +1 phase B
// done (this wavefront), +2 every wavefront's phase B done (barrier), +3 phase C done; 40 outputs written, 41 patch done.
#ifdef NF_TIMELINE
#define NF_WSTAMP(i) do { if (t == 0 && stamp_on) reinterpret_cast<long long *>(a.sd_out)[(size_t)blockIdx.x * 64 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NF_WSTAMP(i) do { } while (0)
#endif

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4hh __attribute__((ext_vector_type(4)));
typedef _Float16 v2hh __attribute__((ext_vector_type(2)));

// relu(a), relu(b) rounded to half, packed in one dword (v_cvt_pk_f16_f32 + v_pk_max_f16)
__device__ __forceinline__ uint32_t relu_pack_h2(float a, float b)
{
    const v2hh h = __builtin_elementwise_max(v2hh{(_Float16)a, (_Float16)b}, v2hh{0, 0});
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ v8h as_v8h(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return __builtin_bit_cast(v8h, make_uint4(a, b, c, d));
}

constexpr int TPW = 8;   // tiles (= rows of a strip) per wavefront

// tuning knobs (A/B builds): re-read the A operands per tile instead of holding all 57 live; occupancy target
#ifndef NF_WIDE_CLOBBER
#define NF_WIDE_CLOBBER 0
#endif
#ifndef NF_WIDE_WPE
#define NF_WIDE_WPE 3
#endif
#ifndef NF_WIDE_WPE16
#define NF_WIDE_WPE16 3   // the fp16 instantiation
#endif
#ifndef NF_WIDE_HOIST16
#define NF_WIDE_HOIST16 0
#endif

// value of the lane one pixel to the left / right (callers multiply the tile ends away: lane 32 receives lane 31's
// value and vice versa).  NF_WIDE_DPP=1: DPP wave_shr:1 / wave_shl:1 — a VALU operand modifier, no LDS traffic and no
// address register; 0: ds_bpermute.
#ifndef NF_WIDE_DPP
#define NF_WIDE_DPP 1
#endif
__device__ __forceinline__ float from_prev(float x, int lane)
{
#if NF_WIDE_DPP
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));
#else
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane - 1) << 2, __float_as_int(x)));
#endif
}
__device__ __forceinline__ float from_next(float x, int lane)
{
#if NF_WIDE_DPP
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true));
#else
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane + 1) << 2, __float_as_int(x)));
#endif
}
// a(lanes 0-31) + a(lanes 32-63) in the low half, b(lanes 0-31) + b(lanes 32-63) in the high half: the two lane
// halves hold partial sums of the same pixels; the low half finishes tile `a`, the high half tile `b` — one
// v_permlane32_swap + one add for two tiles
__device__ __forceinline__ float half_sums(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

//   THREADS  64 x number of strips: (rows / 8) x TPR
//   PHILOX   input = in-kernel Philox/Box-Muller draw
//   TPR      tiles per image row: 1 (W <= 32) or 2 (W <= 64)
//   PREC     0 = fp32 (v_mfma_f32_32x32x2_f32), 1 = fp16 CNN convs (v_mfma_f32_32x32x16_f16, fp32 accumulate; NF5_* layout)
// Pixel ownership: the CNN of a tile needs all 64 lanes (lane half g = K slice), everything else is per pixel —
// so lane half g OWNS the rows row0 + 2m + g (m = 0..3) of its strip: their 4 channel values live in its registers,
// it does their global I/O, 1x1 mixes, signal-dependent scaling, affine update and log-det.
//   TILED    NF_K_TILED launches (nf_device.h): its own kernels (nf_wide32_tiled_kernel), so that the per-tile geometry costs
//            the whole-patch kernels neither scalar registers nor instructions
template <int THREADS, bool PHILOX, int TPR, int PREC, bool TILED>
__device__ __forceinline__ void nf_wide32_body(const NfProgram &prog, const NfLaunch &a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = THREADS / 64;
    constexpr int OWN = TPW / 2;
    constexpr bool H16 = PREC == 1;
    constexpr int IMG = H16 ? NF5_IMG_SIZE : NF4_IMG_SIZE;
    const int H = a.H, W = a.W, HW = H * W;
    const int Wp = W + 2;
    const int PL = ((H + 2) * Wp + 3) & ~3;             // one channel plane of the z0 tile
    float *const z0s = smem;                             // fp32: [2][PL] channel planes; fp16: [PL] half2 per pixel (+ unused plane)
    // [2][CPLB]: the parameter block (border table, scale, weight image) of the current coupling and, arriving by LDS-DMA while
    // this one's CNN runs, of the next one (the first coupling of the next patch behind the last one)
    constexpr int CPLF = NF4_CPL_IMG + IMG;              // floats of a block
    constexpr int CHUNKS = (CPLF + 255) / 256;           // global_load_lds_dwordx4 moves 1 KiB per wavefront instruction
    constexpr int CPLB = CHUNKS * 256;
    float *const wbuf0 = z0s + 2 * PL;
    float *const exch = wbuf0 + 2 * CPLB;                // [NW][2][32][4] strip-boundary rows
    float *const side = exch + NW * 256;                 // TPR == 2: [2][H+2][3][4] column-seam taps
    float *const red = side + (TPR == 2 ? 2 * (H + 2) * 12 : 0);   // [3][NW] reduction scratch

    const int t = threadIdx.x;
    const int w = t >> 6, lane = t & 63, n = lane & 31, g = lane >> 5;
    const int rg = w / TPR, hf = w % TPR;
    const int row0 = rg * TPW, c = hf * 32 + n;
    const bool col_on = c < W;
    // 0/1 multipliers: the seam neighbour of a 64-wide row arrives through `side`, not through a lane shift
    const float ml = n > 0 ? 1.0f : 0.0f, mr = (n < 31 && c + 1 < W) ? 1.0f : 0.0f;
    const float mg0 = g == 0 ? 1.0f : 0.0f, mg1 = g == 1 ? 1.0f : 0.0f;
    const float ml0 = ml * mg0, mr1 = mr * mg1;
    [[maybe_unused]] int toff[4];   // fp16 l_1: z0-tile offsets of the taps 4g .. 4g+3 this lane half contributes
#pragma unroll
    for (int q = 0; q < 4; ++q) toff[q] = ((4 * g + q) / 3) * Wp + (4 * g + q) % 3;

    for (int i = t; i < 2 * PL + 2 * CPLB + NW * 256 + (TPR == 2 ? 2 * (H + 2) * 12 : 0); i += THREADS) smem[i] = 0.0f;
    __syncthreads();

    const int n_ops = prog.n_ops;
    // LDS-DMA of one coupling's block: LDS address = wavefront-uniform base + 16 B x lane, global address per lane (the lanes
    // behind the block's end re-read its start into the padding of the last KiB)
    auto stage = [&](int off, int buf) {
        for (int ch = w; ch < CHUNKS; ch += NW) {
            const int fo = ch * 256 + lane * 4;
            const float *src = a.params + off + (fo < CPLF ? fo : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(wbuf0 + buf * CPLB + ch * 256), 16, 0, 0);
        }
    };
    int first_cpl = -1;
    for (int op = n_ops - 1; op >= 0; --op)
        if (prog.ops[op].type == NF_OP_COUPLING_FWD || prog.ops[op].type == NF_OP_COUPLING_REV) first_cpl = op;
    int cur = 0;   // which half of wbuf0 the next coupling reads
    if (first_cpl >= 0 && (int64_t)blockIdx.x < a.B) stage(prog.ops[first_cpl].off, 0);
    double acc_nll = 0.0, acc_sd = 0.0;   // thread 0 only

    [[maybe_unused]] bool stamp_on = false;
    [[maybe_unused]] int64_t stamp_it = 0;
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
#ifdef NF_TIMELINE
        stamp_on = stamp_it++ == (a.B / gridDim.x) / 2;
        int n_cpl = 0;
        NF_WSTAMP(0);
#endif
        // Where this "patch" sits: on its own ([B,H,W,4] tensors), or — NF_K_TILED (nf_device.h, "overlapping tiles") — as
        // tile b % tiles of image b / tiles: pixel (r, c) of the tile is pixel (oy + r, ox + c) of an IH x IW image, border
        // masks follow the image border, and results are reported for the core window [cy0, cy1) x [cx0, cx1) only.
        size_t patch_off = (size_t)b * (size_t)HW * 4u;
        int64_t patch_id = b;
        int oy = 0, ox = 0, IH = H, IW = W, cy0 = 0, cy1 = H, cx0 = 0, cx1 = W;
        constexpr bool tiled = TILED;
        if constexpr (TILED) {
            const int nt = a.tile_ny * a.tile_nx;
            const int64_t img = b / nt;
            const int ti = (int)(b - img * nt);
            const int ty = ti / a.tile_nx, tx = ti - ty * a.tile_nx;
            IH = a.img_H;
            IW = a.img_W;
            oy = nf_tile_origin(ty, IH, H, a.tile_halo);
            ox = nf_tile_origin(tx, IW, W, a.tile_halo);
            cy0 = nf_tile_core0(ty, IH, H, a.tile_halo);
            cy1 = nf_tile_core1(ty, a.tile_ny, IH, H, a.tile_halo);
            cx0 = nf_tile_core0(tx, IW, W, a.tile_halo);
            cx1 = nf_tile_core1(tx, a.tile_nx, IW, W, a.tile_halo);
            patch_off = (size_t)img * (size_t)IH * (size_t)IW * 4u;
            patch_id = img;
        }
        const int C = ox + c;                                               // image column of this lane's pixels
        const bool col_own = col_on && C >= cx0 && C < cx1;
        const int cmask = (C == 0 ? 4 : 0) | (C == IW - 1 ? 8 : 0);

        float z[OWN][4];
#pragma unroll
        for (int m = 0; m < OWN; ++m) {
            const int r = row0 + 2 * m + g;
            const bool act = r < H && col_on;
            const int gi = act ? (oy + r) * IW + C : 0;
            if (PHILOX) {
                philox_normal4(a.seed, a.patch_base + patch_id, (uint32_t)gi, NF_STREAM_SAMP, z[m]);
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
#ifdef NF_TIMELINE
        asm volatile("" ::"v"(z[0][0]));
        NF_WSTAMP(1);
#endif

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
                // ---- phase A: publish the pass-through half; this coupling's weights arrive (requested during the previous CNN phase) ----
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int r = row0 + 2 * m + g;
                    if (r < H && col_on) {
                        if constexpr (H16) {
                            const v2hh zh = {(_Float16)z[m][0], (_Float16)z[m][1]};
                            reinterpret_cast<uint32_t *>(z0s)[(r + 1) * Wp + c + 1] = __builtin_bit_cast(uint32_t, zh);
                        } else {
                            z0s[(r + 1) * Wp + c + 1] = z[m][0];
                            z0s[PL + (r + 1) * Wp + c + 1] = z[m][1];
                        }
                    }
                }
                // this coupling's block was requested a whole CNN phase ago: every wavefront retires its own pieces, the barrier
                // publishes them (and the z0 tile)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const float *const wblk = wbuf0 + cur * CPLB;      // [E][S][image] as in global memory (nf_device.h, NF4_CPL_*)
                const float *const wbuf = wblk + NF4_CPL_IMG;
                {   // request the next coupling's block into the other half (last read two barriers ago)
                    int nx = op + 1;
                    while (nx < n_ops && prog.ops[nx].type != NF_OP_COUPLING_FWD && prog.ops[nx].type != NF_OP_COUPLING_REV) ++nx;
                    if (nx >= n_ops) nx = b + (int64_t)gridDim.x < a.B ? first_cpl : -1;
                    if (nx >= 0) stage(prog.ops[nx].off, cur ^ 1);
                }
                cur ^= 1;
#ifdef NF_TIMELINE
                if (n_cpl < 8) NF_WSTAMP(2 + 4 * n_cpl);
#endif

                // ---- phase B: the CNN on the matrix cores, strip-local shift-add ----
                const float4 *const wb4 = reinterpret_cast<const float4 *>(wbuf);
#if NF_WIDE_HOIST16
                // fp16: the six A operands and the centre-tap rows of the coupling are read ONCE, ahead of the strip (the tile loop
                // stores to LDS, so the compiler must otherwise re-read them for every tile: 13 of the 21 LDS reads of a tile, each
                // a wait in front of the matrix instruction that consumes it)
                [[maybe_unused]] uint4 hA1[2], hA2[2], hA3[2];
                [[maybe_unused]] uint2 hC[4];
                if constexpr (H16) {
                    const uint4 *const wq0 = reinterpret_cast<const uint4 *>(wbuf);
                    const uint2 *const wc0 = reinterpret_cast<const uint2 *>(wbuf + NF5_IMG_A3CH);
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        hA1[mm] = wq0[NF5_IMG_A1H / 4 + mm * 64 + lane];
                        hA2[mm] = wq0[NF5_IMG_A2H / 4 + mm * 64 + lane];
                        hA3[mm] = wq0[NF5_IMG_A3H / 4 + mm * 64 + lane];
                        hC[2 * mm] = wc0[(2 * mm + 0) * 8 + g * 4 + (lane & 3)];
                        hC[2 * mm + 1] = wc0[(2 * mm + 1) * 8 + g * 4 + (lane & 3)];
                    }
                }
#endif
                float cp[TPW][4];
#pragma unroll
                for (int k = 0; k < TPW; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) cp[k][j] = 0.0f;
#pragma unroll
                for (int k = 0; k < TPW; ++k) {
                    const int r = row0 + k;
                    if (r >= H) continue;   // wave-uniform
                    if constexpr (NF_WIDE_CLOBBER) asm volatile("" ::: "memory");
                    v16f p;
                    v4f pc = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (H16) {
                        const uint4 *const wq = reinterpret_cast<const uint4 *>(wbuf);
                        const uint32_t *const zh = reinterpret_cast<const uint32_t *>(z0s) + r * Wp + c;   // tap (di,dj) at + di*Wp + dj
                        v16f d;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = wb4[NF5_IMG_B1 / 4 + g * 4 + q];
                            d[4 * q + 0] = bb.x; d[4 * q + 1] = bb.y; d[4 * q + 2] = bb.z; d[4 * q + 3] = bb.w;
                        }
                        {   // l_1: K = 18 in two instructions; lane half g brings taps 4g .. 4g+3, then (half 0) tap 8
#if NF_WIDE_HOIST16
                            const uint4 a0 = hA1[0], a1 = hA1[1];
#else
                            const uint4 a0 = wq[NF5_IMG_A1H / 4 + lane], a1 = wq[NF5_IMG_A1H / 4 + 64 + lane];
#endif
                            const v8h b0 = as_v8h(zh[toff[0]], zh[toff[1]], zh[toff[2]], zh[toff[3]]);
                            const v8h b1 = as_v8h(zh[2 * Wp + 2], 0u, 0u, 0u);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a0), b0, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a1), b1, d, 0, 0, 0);
                        }
                        v16f e;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bb = wb4[NF5_IMG_B2 / 4 + g * 4 + q];
                            e[4 * q + 0] = bb.x; e[4 * q + 1] = bb.y; e[4 * q + 2] = bb.z; e[4 * q + 3] = bb.w;
                        }
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm) {
#if NF_WIDE_HOIST16
                            const uint4 aw = hA2[mm];
#else
                            const uint4 aw = wq[NF5_IMG_A2H / 4 + mm * 64 + lane];
#endif
                            const v8h hb = as_v8h(relu_pack_h2(d[8 * mm + 0], d[8 * mm + 1]), relu_pack_h2(d[8 * mm + 2], d[8 * mm + 3]),
                                                  relu_pack_h2(d[8 * mm + 4], d[8 * mm + 5]), relu_pack_h2(d[8 * mm + 6], d[8 * mm + 7]));
                            e = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, aw), hb, e, 0, 0, 0);
                        }
#pragma unroll
                        for (int v = 0; v < 16; ++v) p[v] = 0.0f;
                        const uint2 *const wc = reinterpret_cast<const uint2 *>(wbuf + NF5_IMG_A3CH);
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm) {
#if NF_WIDE_HOIST16
                            const uint4 aw = hA3[mm];
#else
                            const uint4 aw = wq[NF5_IMG_A3H / 4 + mm * 64 + lane];
#endif
                            const uint32_t h0 = relu_pack_h2(e[8 * mm + 0], e[8 * mm + 1]), h1 = relu_pack_h2(e[8 * mm + 2], e[8 * mm + 3]);
                            const uint32_t h2 = relu_pack_h2(e[8 * mm + 4], e[8 * mm + 5]), h3 = relu_pack_h2(e[8 * mm + 6], e[8 * mm + 7]);
                            p = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, aw), as_v8h(h0, h1, h2, h3), p, 0, 0, 0);
#if NF_WIDE_HOIST16
                            const uint2 c0 = hC[2 * mm], c1 = hC[2 * mm + 1];
#else
                            const uint2 c0 = wc[(2 * mm + 0) * 8 + g * 4 + (lane & 3)], c1 = wc[(2 * mm + 1) * 8 + g * 4 + (lane & 3)];
#endif
                            pc = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4hh, c0), __builtin_bit_cast(v4hh, make_uint2(h0, h1)), pc, 0, 0, 0);
                            pc = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(v4hh, c1), __builtin_bit_cast(v4hh, make_uint2(h2, h3)), pc, 0, 0, 0);
                        }
                    } else {
                    v16f d;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bb = wb4[NF4_IMG_B1 / 4 + g * 4 + q];
                        d[4 * q + 0] = bb.x; d[4 * q + 1] = bb.y; d[4 * q + 2] = bb.z; d[4 * q + 3] = bb.w;
                    }
                    const float *zb = z0s + g * PL + r * Wp + c;   // tap (di,dj) at + di*Wp + dj
#pragma unroll
                    for (int grp = 0; grp < 3; ++grp) {
                        const float4 aw = wb4[NF4_IMG_A1 / 4 + grp * 64 + lane];
                        const float as[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int tap = grp * 4 + s;
                            if (tap < 9) d = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], zb[(tap / 3) * Wp + tap % 3], d, 0, 0, 0);
                        }
                    }
                    v16f e;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bb = wb4[NF4_IMG_B2 / 4 + g * 4 + q];
                        e[4 * q + 0] = bb.x; e[4 * q + 1] = bb.y; e[4 * q + 2] = bb.z; e[4 * q + 3] = bb.w;
                    }
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        const float4 aw = wb4[NF4_IMG_A2 / 4 + grp * 64 + lane];
                        const float as[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            e = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], nf_relu(d[grp * 4 + s]), e, 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < 16; ++v) p[v] = 0.0f;
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        const float4 aw = wb4[NF4_IMG_A3 / 4 + grp * 64 + lane];
                        const float4 ac = wb4[NF4_IMG_A3C / 4 + grp * 8 + g * 4 + (lane & 3)];
                        const float as[4] = {aw.x, aw.y, aw.z, aw.w};
                        const float cs[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const float h = nf_relu(e[grp * 4 + s]);
                            p = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], h, p, 0, 0, 0);
                            pc = __builtin_amdgcn_mfma_f32_4x4x1f32(cs[s], h, pc, 0, 0, 0);
                        }
                    }
                    }
                    // horizontal part of the shift-add.  Register group a of lane half g' holds the taps
                    //   a=0: (0,0)|(2,0) -> from the left   a=1: (0,2)|(2,2) -> from the right
                    //   a=2: (0,1)|(2,1) -> in place         a=3: (1,0)|(1,2) -> left | right
                    float rm[4];   // g'=0: R[.][di=0] (goes one row down), g'=1: R[.][di=2] (goes one row up)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        rm[j] = fmaf(from_next(p[4 + j], lane), mr, fmaf(from_prev(p[j], lane), ml, p[8 + j]));
                        cp[k][j] += fmaf(from_next(p[12 + j], lane), mr1, fmaf(from_prev(p[12 + j], lane), ml0, pc[j]));
                    }
                    if (k + 1 < TPW) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) cp[k + 1][j] = fmaf(rm[j], mg0, cp[k + 1][j]);
                    }
                    if (k > 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) cp[k - 1][j] = fmaf(rm[j], mg1, cp[k - 1][j]);
                    }
                    if (k == 0 && g == 1 && row0 > 0)
                        *reinterpret_cast<float4 *>(exch + (w * 2 + 1) * 128 + n * 4) = make_float4(rm[0], rm[1], rm[2], rm[3]);
                    if (k == TPW - 1 && g == 0 && row0 + TPW < H)
                        *reinterpret_cast<float4 *>(exch + (w * 2 + 0) * 128 + n * 4) = make_float4(rm[0], rm[1], rm[2], rm[3]);
                    if constexpr (TPR == 2) {
                        // the column seam 31|32: the taps that cross it, raw, for the neighbour half's phase C
                        float *sl = side + (0 * (H + 2) + r + 1) * 12, *sr = side + (1 * (H + 2) + r + 1) * 12;
                        if (hf == 0 && n == 31) {   // (di,0) taps of pixel 31 belong to pixel 32
                            if (g == 0) {
                                *reinterpret_cast<float4 *>(sl + 0) = make_float4(p[0], p[1], p[2], p[3]);
                                *reinterpret_cast<float4 *>(sl + 4) = make_float4(p[12], p[13], p[14], p[15]);
                            } else {
                                *reinterpret_cast<float4 *>(sl + 8) = make_float4(p[0], p[1], p[2], p[3]);
                            }
                        }
                        if (hf == 1 && n == 0) {    // (di,2) taps of pixel 32 belong to pixel 31
                            if (g == 0) {
                                *reinterpret_cast<float4 *>(sr + 0) = make_float4(p[4], p[5], p[6], p[7]);
                            } else {
                                *reinterpret_cast<float4 *>(sr + 8) = make_float4(p[4], p[5], p[6], p[7]);
                                *reinterpret_cast<float4 *>(sr + 4) = make_float4(p[12], p[13], p[14], p[15]);
                            }
                        }
                    }
                }
#ifdef NF_TIMELINE
                asm volatile("" ::"v"(cp[0][0]), "v"(cp[TPW - 1][3]));
                if (n_cpl < 8) NF_WSTAMP(3 + 4 * n_cpl);
#endif
                __syncthreads();
#ifdef NF_TIMELINE
                if (n_cpl < 8) NF_WSTAMP(4 + 4 * n_cpl);
#endif

                // ---- phase C: strip-boundary rows, column seam, then each lane half finishes the rows it owns ----
                if (row0 > 0 && g == 0) {
                    const float4 v = *reinterpret_cast<const float4 *>(exch + ((w - TPR) * 2 + 0) * 128 + n * 4);
                    cp[0][0] += v.x; cp[0][1] += v.y; cp[0][2] += v.z; cp[0][3] += v.w;
                }
                if (row0 + TPW < H && g == 1) {
                    const float4 v = *reinterpret_cast<const float4 *>(exch + ((w + TPR) * 2 + 1) * 128 + n * 4);
                    cp[TPW - 1][0] += v.x; cp[TPW - 1][1] += v.y; cp[TPW - 1][2] += v.z; cp[TPW - 1][3] += v.w;
                }
                if constexpr (TPR == 2) {
                    if (g == 0 && ((hf == 1 && n == 0) || (hf == 0 && n == 31))) {
#pragma unroll
                        for (int k = 0; k < TPW; ++k) {
                            const int r = row0 + k;
                            if (r >= H) continue;
                            const float *sb = side + ((hf == 1 ? 0 : 1) * (H + 2) + r) * 12;   // rows r-1, r, r+1 at +0, +12, +24
#pragma unroll
                            for (int di = 0; di < 3; ++di) {
                                const float4 v = *reinterpret_cast<const float4 *>(sb + di * 12 + di * 4);
                                cp[k][0] += v.x; cp[k][1] += v.y; cp[k][2] += v.z; cp[k][3] += v.w;
                            }
                        }
                    }
                }
                const float scl = P[NF4_CPL_S + 1], m2scl = P[NF4_CPL_S + 2];
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int r = row0 + 2 * m + g;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = half_sums(cp[2 * m][j], cp[2 * m + 1][j]);
                    const bool act = r < H && col_on;
                    const int R = oy + r;
                    const bool own = r < H && col_own && R >= cy0 && R < cy1;
                    const int bm = (R == 0 ? 1 : 0) | (R == IH - 1 ? 2 : 0) | cmask;
                    const float4 eb = *reinterpret_cast<const float4 *>(wblk + NF4_CPL_E + 4 * (act ? bm : 0));
                    o[0] += eb.x; o[1] += eb.y; o[2] += eb.z; o[3] += eb.w;   // (fp16 layouts too: 2 log2(e) sits inside the rounded weights)
                    // raw columns are pre-scaled by 2 log2(e):  t = exp2(raw') = exp(2 raw);
                    // ls*log2(e) = scl*tanh(raw) = scl - 2 scl/(t + 1); log-det accumulated in log2 units
                    const float l0 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[2]) + 1.0f), m2scl, scl);
                    const float l1 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[3]) + 1.0f), m2scl, scl);
                    if (type == NF_OP_COUPLING_FWD) {
                        z[m][2] = fmaf(z[m][2], __builtin_amdgcn_exp2f(l0), o[0]);
                        z[m][3] = fmaf(z[m][3], __builtin_amdgcn_exp2f(l1), o[1]);
                        if (own) ld2 += l0 + l1;
                    } else {
                        z[m][2] = (z[m][2] - o[0]) * __builtin_amdgcn_exp2f(-l0);
                        z[m][3] = (z[m][3] - o[1]) * __builtin_amdgcn_exp2f(-l1);
                    }
                }
#ifdef NF_TIMELINE
                asm volatile("" ::"v"(z[0][2]));
                if (n_cpl < 8) NF_WSTAMP(5 + 4 * n_cpl);
                ++n_cpl;
#endif
            } else if (type == NF_OP_SDN_DIV || type == NF_OP_SDN_MUL) {
                // AffineCouplingSdnEx5: scale = sqrt(beta1*y/gain + beta2)  (cond_utils.py:238)
                const float4 *y4 = reinterpret_cast<const float4 *>(a.y + patch_off);
                const float ck1 = a.cond_a[prog.ops[op].off & 3], cb2 = a.cond_b[prog.ops[op].off & 3];
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int r = row0 + 2 * m + g;
                    const bool act = r < H && col_on;
                    const int R = oy + r;
                    const bool own = r < H && col_own && R >= cy0 && R < cy1;
                    float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (act) yv = y4[R * IW + C];
                    const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = fmaf(yy[q], ck1, cb2);
                        if (type == NF_OP_SDN_DIV) {
                            z[m][q] = z[m][q] * __builtin_amdgcn_rsqf(v);
                            if (own) ld = fmaf(-0.34657359027997264f, __builtin_amdgcn_logf(v), ld);
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
                const int r = row0 + 2 * m + g, R = oy + r;
                if (r < H && col_own && R >= cy0 && R < cy1) out4[R * IW + C] = make_float4(z[m][0], z[m][1], z[m][2], z[m][3]);
            }
        }
        NF_WSTAMP(40);
        if (a.nll_out || a.sd_out || a.ld_out || a.sums || (tiled && a.tile_part)) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < OWN; ++m)
                if (row0 + 2 * m + g < H && col_own && oy + row0 + 2 * m + g >= cy0 && oy + row0 + 2 * m + g < cy1) {
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
            if (t == 0 && tiled) {
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
                const double sd = sqrt(var);
                if (a.nll_out) a.nll_out[b] = (float)nll;
#ifndef NF_TIMELINE
                if (a.sd_out) a.sd_out[b] = (float)sd;
#endif
                if (a.ld_out) a.ld_out[b] = (float)logdet;
                acc_nll += (double)(float)nll;
                acc_sd += (double)(float)sd;
            }
            __syncthreads();   // scratch is reused by the next patch
        }
        NF_WSTAMP(41);
    }

    if (a.sums && t == 0 && !TILED) {
        double *sp = a.sums;
        if (a.flags & NF_K_SUMS_WIDE) sp += (size_t)(blockIdx.x & (NF_SUMS_SLOTS - 1)) * NF_SUMS_STRIDE;
        atomicAdd(&sp[0], acc_nll);
        atomicAdd(&sp[1], acc_sd);
        if (blockIdx.x == 0) atomicAdd(&sp[2], (double)a.B);
    }
}

template <int THREADS, bool PHILOX, int TPR, int PREC>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 256 ? (PREC == 1 ? NF_WIDE_WPE16 : NF_WIDE_WPE) : 1))) void nf_wide32_kernel(const NfProgram prog, const NfLaunch a)
{
    nf_wide32_body<THREADS, PHILOX, TPR, PREC, false>(prog, a);
}

template <int THREADS, bool PHILOX, int TPR, int PREC>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 256 ? NF_WIDE_WPE : 1))) void nf_wide32_tiled_kernel(const NfProgram prog, const NfLaunch a)
{
    nf_wide32_body<THREADS, PHILOX, TPR, PREC, true>(prog, a);
}

size_t wide_lds_bytes(int H, int W, int threads, int tpr, int prec)
{
    const int Wp = W + 2, PL = ((H + 2) * Wp + 3) & ~3, NW = threads / 64;
    const size_t cplb = ((size_t)(NF4_CPL_IMG + (prec ? NF5_IMG_SIZE : NF4_IMG_SIZE)) + 255) / 256 * 256;   // as CPLB in the kernel
    size_t f = 2 * (size_t)PL + 2 * cplb + (size_t)NW * 256 + (tpr == 2 ? 2 * (size_t)(H + 2) * 12 : 0) + ((3 * NW + 3) & ~3);
    return f * sizeof(float);
}

template <int THREADS, bool PHILOX, int TPR, int PREC>
hipError_t launch_wide_p(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    const size_t lds = wide_lds_bytes(a.H, a.W, THREADS, TPR, PREC);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const bool tiled = (a.flags & NF_K_TILED) != 0;
    void (*const kern)(const NfProgram, const NfLaunch) =
        tiled ? &nf_wide32_tiled_kernel<THREADS, PHILOX, TPR, PREC> : &nf_wide32_kernel<THREADS, PHILOX, TPR, PREC>;
    const void *fn = reinterpret_cast<const void *>(kern);
    // (tiled << 48 | device << 40 | lds bytes << 8 | resident workgroups per CU) of the last query; racy but idempotent
    static std::atomic<uint64_t> cache{0};
    const uint64_t key = ((uint64_t)(tiled ? 1 : 0) << 48) | ((uint64_t)(device & 0xff) << 40) | ((uint64_t)lds << 8);
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
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(THREADS), lds, stream, prog, a);
    return hipGetLastError();
}

template <int THREADS, bool PHILOX, int TPR>
hipError_t launch_wide(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (a.flags & NF_K_FP16_CNN) return launch_wide_p<THREADS, PHILOX, TPR, 1>(prog, a, n_cu, device, stream);
    return launch_wide_p<THREADS, PHILOX, TPR, 0>(prog, a, n_cu, device, stream);
}

template <bool PHILOX>
hipError_t dispatch_wide(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (a.W <= 32) {
        if (a.H <= 32) return launch_wide<256, PHILOX, 1>(prog, a, n_cu, device, stream);
        return launch_wide<512, PHILOX, 1>(prog, a, n_cu, device, stream);
    }
    if (a.H <= 32) return launch_wide<512, PHILOX, 2>(prog, a, n_cu, device, stream);
    return launch_wide<1024, PHILOX, 2>(prog, a, n_cu, device, stream);
}

}  // namespace

// entry point used by nf_host.hip: programs in the NF4 layout (coupling width 32), patches up to 64x64
hipError_t nf_launch_wide(const NfProgram &prog, const NfLaunch &a, int n_cu, int device, hipStream_t stream)
{
    if (prog.width != 32 || a.H < 1 || a.W < 1 || a.H > 64 || a.W > 64) return hipErrorInvalidValue;
    if (a.flags & NF_K_PHILOX_IN) return dispatch_wide<true>(prog, a, n_cu, device, stream);
    return dispatch_wide<false>(prog, a, n_cu, device, stream);
}
