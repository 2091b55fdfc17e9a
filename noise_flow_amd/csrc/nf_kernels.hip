// Fused Noise Flow bijector-stack kernels for gfx950 (MI355X, CDNA4).
//
// One workgroup evaluates whole patches: the 4 channel values of every pixel
// stay in registers across all ~18 bijectors; the two tensors a coupling CNN
// needs with a spatial halo (the pass-through half z0 and the hidden map h2) are
// staged in zero-bordered LDS tiles, so 'SAME' padding needs no bounds checks and
// HBM traffic is the algorithmic minimum (read x and y once, write 1-3 scalars).
// Weights reach the FMAs in one of two ways (same arithmetic):
//   scalar      wave-uniform s_loads into SGPRs, v_pk_fma_f32 with SGPR operands (widths 4..32);
//   matrix core width 4: every conv has exactly 4 output channels and maps onto
//               v_mfma_f32_4x4x1_16b_f32 (fp32) or v_mfma_f32_4x4x4_16b_f16 (fp16-CNN mode) with the
//               weights as the A operand, read j-major from an LDS image of the whole model.
//
// Replaces (reference, /root/reference): the TF graph built by
// borealisflows/noise_flow_model.py:394-456 out of layers.py:74-145 (Conv2d1x1),
// layers.py:251-375 + 452-498 (AffineCoupling + coupling CNN),
// noise_flow_layers/AffineCouplingSdnEx5.py, AffineCouplingGainEx4.py and the
// Gaussian prior noise_flow_model.py:486-541.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include "../../include/noiseflow_hip.h"   // NF_SUMS_SLOTS / NF_SUMS_STRIDE
#include "nf_device.h"
#include "nf_dev_util.h"

// s_setprio level of a wave while it streams MFMAs (0 = off): keeps bursts of matrix
// instructions contiguous on a SIMD that other waves share (+5 % at 512-thread geometry)
#ifndef NF_PRIO
#define NF_PRIO 2
#endif
// NF_FAIR: progress-based wave priority.  A SIMD arbitrates its resident waves oldest-first, so of the 4 workgroups
// that share a CU the oldest runs ahead and the youngest starves: at B = 1024 (one patch per workgroup) workgroup
// lifetimes spread 37..57 us around a 48 us mean and the launch lasts as long as the slowest one (tools/timeline.py).
// Lowering a wave's priority as it advances through the couplings (3,3,2,2,1,1,0,0) is a negative feedback that keeps
// co-resident workgroups level.  0 = off (MFMA bursts at NF_PRIO, the round-1 behaviour).
#ifndef NF_BR_SWIZZLE
#define NF_BR_SWIZZLE 1
#endif
#ifndef NF_WAVE_PRIO
#define NF_WAVE_PRIO 1
#endif
#ifndef NF_FAIR
#define NF_FAIR 1
#endif
#if NF_FAIR
#define NF_PRIO_UP()   do { } while (0)
#define NF_PRIO_DOWN() do { } while (0)
#else
#define NF_PRIO_UP()   do { if (NF_PRIO) __builtin_amdgcn_s_setprio(NF_PRIO); } while (0)
#define NF_PRIO_DOWN() do { if (NF_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
#endif
// occupancy target (waves per SIMD) the register allocator must honour, per geometry
#ifndef NF_WPE_512
#define NF_WPE_512 5
#endif
#ifndef NF_WPE_256
#define NF_WPE_256 3
#endif
// warm-up touches before the LDS set-up: every NF_WARM_STEP-th owned pixel (2x2-blocked lanes: 2 = one per image row)
#ifndef NF_WARM_STEP
#define NF_WARM_STEP 2
#endif
#define NF_MIN_WAVES(T, P, M) (((T) == 512 && (P) == 2 && (M)) ? NF_WPE_512 : ((T) == 256 && (P) == 4 && (M)) ? NF_WPE_256 : 1)

// Instrumented build (-DNF_TIMELINE, tools/timeline.py only): thread 0 of every workgroup stamps the 100 MHz
// s_memrealtime counter at the phase boundaries of its MIDDLE patch (the first one when it has one) into NfLaunch::sd_out, reinterpreted as
// int64[grid][16] (sd_z is not written in this build).
#ifdef NF_TIMELINE
#define NF_STAMP(i) do { if (t == 0 && stamp_on) reinterpret_cast<long long *>(a.sd_out)[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NF_STAMP(i) do { } while (0)
#endif

namespace {

// Batch-statistics pass (nf_nll_batchstats / nf_sample_batchstats): per-channel sum and sum of
// squares of one pixel's pre-normalisation activations, reduced over the wavefront and added to
// the call-wide fp64 accumulators stats[0..WIDTH) / stats[WIDTH..2*WIDTH) of this workgroup's slot
// (NF_STATS_SLOTS slots spread the atomics; the host adds them up)  (layers.py:388-391).
template <int WIDTH>
__device__ __forceinline__ void stats_accumulate(const float (&h)[WIDTH], bool active, float (&s1)[WIDTH], float (&s2)[WIDTH])
{
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) {
        const float v = active ? h[j] : 0.0f;
        s1[j] += v;
        s2[j] = fmaf(v, v, s2[j]);
    }
}
template <int WIDTH>
__device__ __forceinline__ void stats_direct(const float (&h)[WIDTH], bool active, double *stats, int t)
{
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) {
        const float v = active ? h[j] : 0.0f;
        const float s = wave_sum(v), q = wave_sum(v * v);
        if ((t & 63) == 0) {
            atomicAdd(&stats[j], (double)s);
            atomicAdd(&stats[WIDTH + j], (double)q);
        }
    }
}
// one wavefront reduction and one pair of fp64 atomics per channel per wavefront per patch
template <int WIDTH>
__device__ __forceinline__ void stats_flush(const float (&s1)[WIDTH], const float (&s2)[WIDTH], double *stats, int t)
{
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) {
        const float s = wave_sum(s1[j]), q = wave_sum(s2[j]);
        if ((t & 63) == 0) {
            atomicAdd(&stats[j], (double)s);
            atomicAdd(&stats[WIDTH + j], (double)q);
        }
    }
}

// --------------------------------------------------------------------------
// The fused flow kernel.
//   WIDTH   coupling CNN width
//   THREADS workgroup size (multiple of 64)
//   PX      pixels per thread; H*W <= THREADS*PX
//   PHILOX  input = in-kernel Philox/Box-Muller draw (sampling without a supplied eps)
//   MFMA    matrix-core weight delivery (width 4)
//   FULL    square 32x32 / 64x64 patch filling the workgroup exactly: compile-time geometry, no masks
//   PREC    0 = all fp32, 1 = fp16 coupling-CNN convs (fp32 accumulate; FULL matrix-core only)
//   BS      matrix-core kernel of the batch-statistics mode (nf_*_batchstats at width 4): applies the pending
//           re-fold of NfLaunch::fix_* to its LDS weight image in the prologue and gathers NfLaunch::stats
// Global I/O is 16 bytes per lane per pixel ([H,W,4] fp32, NHWC).
// --------------------------------------------------------------------------
//   TF      NF_K_TILED launches whose tiles are full 64x64 blocks (FULL geometry, per-tile addresses and border masks);
//           masked (!FULL) instantiations take tiled launches of any tile shape at run time
template <int WIDTH, int THREADS, int PX, bool PHILOX, bool MFMA, bool FULL, int PREC, bool BS, bool TF>
__device__ __forceinline__ void nf_flow_body(const NfProgram &prog, const NfLaunch &a)
{
    static_assert(!BS || (MFMA && PREC == 0), "the batch-statistics variant is the fp32 matrix-core kernel");
    static_assert(!TF || (FULL && MFMA && PX == 4 && PREC != 1 && !BS), "tiled full-geometry launches: the fp32 matrix-core kernel and the fp16-CNN kernel on v_mfma_f32_16x16x32_f16");
    static_assert(WIDTH % 4 == 0, "WIDTH must be a multiple of 4");
    static_assert(!MFMA || WIDTH == 4, "the matrix-core path is the width-4 specialisation");
    static_assert(PREC == 0 || (MFMA && FULL && PX == 4), "the fp16-CNN mode exists for full 2x2-blocked patches only");
    constexpr bool H16 = PREC == 1;   // coupling-CNN convs on fp16 matrix cores (fp32 accumulate): v_mfma_f32_4x4x4_16b_f16
    constexpr bool HB = PREC == 2;    // the same on v_mfma_f32_16x16x32_f16 (nf_device.h, NF11_*)
    constexpr bool HALF = PREC != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // FULL = square patch that fills the workgroup exactly (32x32 or 64x64): the geometry is a
    // compile-time constant, so every tile offset folds into a DS-instruction immediate
    constexpr int SIDE = THREADS * PX == 1024 ? 32 : THREADS * PX == 4096 ? 64 : 0;
    static_assert(!FULL || SIDE != 0, "FULL needs a 32x32 or 64x64 workgroup footprint");
    const int H = FULL ? SIDE : a.H, W = FULL ? SIDE : a.W, HW = H * W;
    // row pitch of the plain row-major tiles; fp16 mode at 32x32: 48 entries so that the next
    // block row (2 tile rows down) starts a multiple of 128 B (half2) / 256 B (4 x half) away
    const int Wp = HB ? nf11_pitch(SIDE) : (H16 && SIDE == 32) ? 48 : W + 2;
    const int tile_px = ((H + 2) * Wp + 1) & ~1;         // even -> 16-byte aligned sections
    // MFMA: a 4x4 identity behind the model (the `mix` of a coupling that has none in front: the pair loop below has ONE shape)
    [[maybe_unused]] const int ident_off = (a.n_params + 3) & ~3;
    // LDS order.  Default: tiles | reduction scratch | model image (+ identity) | BS partials.  HB: the model image FIRST — every
    // weight read of a coupling is then `coupling base + lane term` plus an immediate below 64 KiB (behind 63 KiB of tiles the
    // A3 / A2 operands were not, and cost a v_add_u32 each), while the tile accesses keep their compile-time immediates relative to
    // lane addresses that carry the (run-time) tile base
    constexpr bool WFIRST = HB;
    constexpr int TILE_WORDS = HALF ? 3 : 2 + WIDTH;                  // 32-bit words per tile pixel
    constexpr int RED_WORDS = (6 * (THREADS / 64) + 3) & ~3;
    float *const tiles = WFIRST ? smem + ident_off + 16 : smem;
    float2 *const t0 = reinterpret_cast<float2 *>(tiles);  // z0 tile  [tile_px] float2
    float *const th = tiles + 2 * tile_px;                 // h2 tile  [tile_px][WIDTH]
    // fp16-CNN mode: the same two tiles hold half2 / 4 x half per pixel, plain row-major
    uint32_t *const t0h = reinterpret_cast<uint32_t *>(tiles);         // [tile_px] half2
    uint2 *const thh = reinterpret_cast<uint2 *>(tiles + tile_px);     // [tile_px] 4 x half
    float *const red = tiles + TILE_WORDS * tile_px;       // reduction scratch [2][3][THREADS/64] (+pad), alternating by patch
    float *const wl = WFIRST ? smem : red + RED_WORDS;     // MFMA: the whole folded model, j-major
    // BS: [THREADS/64][8] per-wavefront statistics partials, then 8 doubles (scale[4], mean[4]) of the pending re-fold
    [[maybe_unused]] float *const bs_part = WFIRST ? red + RED_WORDS : wl + ident_off + 16;

    const int t = threadIdx.x;
    const int j4 = t & 3;   // MFMA: which output channel's weights this lane feeds as the A operand
    [[maybe_unused]] bool stamp_on = true;
    [[maybe_unused]] int64_t stamp_it = 0;
    NF_STAMP(0);

    // Pixel ownership.
    //  * default: pixel p of the patch belongs to thread p % THREADS (slot p / THREADS);
    //  * BLK (matrix-core path, full 32x32 / 64x64 patches): every lane owns a 2x2 pixel block, so
    //    the 4x4 input window of the block is read from LDS ONCE (16 loads instead of 36 per conv)
    //    and the LDS tiles are split in two column-parity planes to keep those reads conflict-free:
    //    entry(r', c') = ((r'*2 + (c'&1)) * PW + (c'>>1)),  r' = row+1, c' = col+1, PW = W/2 + 1.
    constexpr bool BLK = MFMA && FULL && PX == 4;
    const int PW = (W >> 1) + 1;
    int lidx[PX];      // index of the pixel inside the zero-bordered tiles
    int gidx[PX];      // index of the pixel inside the patch (float4 units)
    int bmask[PX];     // border mask: top | bottom<<1 | left<<2 | right<<3
    bool act[PX];
    bool own[PX];      // pixels whose results this workgroup reports: act[k], narrowed to the tile's core window by NF_K_TILED launches
    [[maybe_unused]] int prow[PX], pcol[PX];   // masked instantiations: row / column of the pixel inside the patch (tile)
    int wbase = 0;     // BLK: tile entry of the window origin (r' = 2*br, c' = 2*bc)
    [[maybe_unused]] int wbase3 = 0;   // HB: first entry of this lane's l_last B operand (unit 0, instruction 0)
    if constexpr (HB) {
        // nf_device.h, NF11_*: a wavefront owns 8 rows x 32 columns = 4 units of 2 rows; lane 16 g + n is pixel
        // (a, p) = (g >> 1, g & 1) of the 2x2 block of lane column n in each of them
        const int wv = t >> 6, g = (t >> 4) & 3, n = t & 15;
        const int q = SIDE == 64 ? (wv & 1) : 0, band = SIDE == 64 ? (wv >> 1) : wv;
        const int c = 32 * q + 2 * n + (g & 1);
        wbase = (band * 8 + nf11_l1_row(g)) * Wp + 32 * q + 2 * n;
        wbase3 = (band * 8 + nf11_l3_row(g, 0)) * Wp + 32 * q + 2 * n + 2 * (g >> 1);
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const int r = band * 8 + 2 * k + (g >> 1);
            act[k] = true;
            own[k] = true;
            prow[k] = r;
            pcol[k] = c;
            gidx[k] = r * W + c;
            lidx[k] = (r + 1) * Wp + (c + 1);
            bmask[k] = (r == 0 ? 1 : 0) | (r == H - 1 ? 2 : 0) | (c == 0 ? 4 : 0) | (c == W - 1 ? 8 : 0);
        }
    } else if constexpr (BLK) {
        const int bw = W >> 1;
        int br = t / bw, bc = t - br * bw;
#if NF_BR_SWIZZLE
        // 32x32, fp32 tiles: a block row is 16 lanes and the tile entries of two neighbouring block rows sit 4*PW = 68 entries
        // apart, i.e. 4 bank quads (mod 16) — the 16-byte LDS reads of a half-wavefront are served in lane groups that mix
        // lanes of two block rows ({0-3,12-15,20-27}, ...), and a quad offset of 4 makes a quarter of them collide
        // (SQ_LDS_BANK_CONFLICT = 19 % of the LDS-active cycles).  Dealing the block rows to the wavefronts with a stride of 4
        // (rows w, w+4, w+8, w+12) puts the two halves 272 entries = 0 quads apart: conflict-free.
        if (!H16 && bw == 16) {
            br = (t >> 6) + 4 * ((t >> 4) & 3);
            bc = t & 15;
        }
#endif
        wbase = H16 ? (2 * br) * Wp + 2 * bc : (2 * br * 2) * PW + bc;
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const int dy = k >> 1, dx = k & 1;
            const int r = 2 * br + dy, c = 2 * bc + dx;
            act[k] = true;
            own[k] = true;
            prow[k] = r;
            pcol[k] = c;
            gidx[k] = r * W + c;
            lidx[k] = H16 ? (r + 1) * Wp + (c + 1) : wbase + ((dy + 1) * 2 + ((dx + 1) & 1)) * PW + ((dx + 1) >> 1);
            bmask[k] = (r == 0 ? 1 : 0) | (r == H - 1 ? 2 : 0) | (c == 0 ? 4 : 0) | (c == W - 1 ? 8 : 0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const int p = t + THREADS * k;
            act[k] = FULL || p < HW;   // FULL: the patch fills the workgroup exactly -> no masking code at all
            own[k] = act[k];
            const int pp = act[k] ? p : 0;
            const int r = pp / W, c = pp - r * W;
            prow[k] = r;
            pcol[k] = c;
            gidx[k] = pp;
            lidx[k] = (r + 1) * Wp + (c + 1);
            bmask[k] = (r == 0 ? 1 : 0) | (r == H - 1 ? 2 : 0) | (c == 0 ? 4 : 0) | (c == W - 1 ? 8 : 0);
        }
    }

    // The model image on its way to LDS: every lane's share is requested HERE, in one batch ahead of everything else (the copy loop
    // at the end of the set-up was np4 / THREADS dependent round trips to L2 — 3 at 32x32 —, and requests issued behind the x / y
    // touches below would make the stores wait for those HBM loads too: the counter retires loads in order); the stores follow the
    // tile zeroing.  NF_WSTAGE 0: the old loop.
#ifndef NF_WSTAGE
#define NF_WSTAGE 1
#endif
    constexpr int WU = MFMA && NF_WSTAGE ? (THREADS >= 1024 ? 3 : 6) : 1;   // float4 per lane in the batch: NF11_MAX_FLOATS / NF2_MAX_FLOATS fit
    [[maybe_unused]] float4 wreg[WU];
    if constexpr (MFMA && NF_WSTAGE) {
        const int np4 = a.n_params >> 2;
        const float4 *const p4 = reinterpret_cast<const float4 *>(a.params);
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int i = t + u * THREADS;
            wreg[u] = p4[i < np4 ? i : 0];
        }
    }

    // Touch the first patch's inputs before the LDS set-up below: with one patch per workgroup (B = the
    // resident capacity) every workgroup would otherwise sit through the set-up and THEN through the HBM
    // latency of its first loads, all at the same time.  The values are discarded; the real loads hit L2.
    // The first patch's x goes straight into registers and y gets one touch per image row of the lane's 2x2 block
    // (the values are discarded; the real loads of the sdn layer hit L2) — both in flight during the set-up.
    float warm = 0.0f;
    float4 zin[PX];   // raw x of the NEXT patch this workgroup evaluates (software prefetch across the patch loop)
#pragma unroll
    for (int k = 0; k < PX; ++k) zin[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // YPF — one 1024-thread workgroup per CU (64x64 patches): nothing else on the CU covers the HBM latency of the clean image
    // the leading sdn layer reads (tools/timeline.py: 2.3 us between two patches, 12 % of a patch), so y of the next patch
    // travels with its x, and the leading sdn layer is peeled off the op loop so that these registers are dead inside it
#ifndef NF_YPF
#define NF_YPF 1
#endif
    constexpr bool YPF = NF_YPF && MFMA && FULL && !PHILOX && THREADS == 1024 && !TF && !BS && (PREC != 0 || NF_YPF > 1);   // fp32 at 64x64 has no 16 registers to spare
    [[maybe_unused]] float4 yin[YPF ? PX : 1];
    [[maybe_unused]] const bool sdn_first = YPF && a.y && prog.n_ops > 0 && prog.ops[0].type == NF_OP_SDN_DIV;
    if constexpr (YPF) {
#pragma unroll
        for (int k = 0; k < PX; ++k) yin[k] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    // NF_STAGGER: a one-round launch (B = the resident capacity) would request all its 32 MiB of inputs in the same
    // microsecond and compute nothing until the last byte arrives (tools/timeline.py).  Only the first quarter of the
    // grid — with round-robin dispatch the first workgroup of every CU — asks before its LDS set-up; the second quarter
    // asks after it, the third and fourth NF_STAGGER_SLEEP x 64 cycles later each: the early ones compute while the late
    // ones load (53.3 -> 49.8 us per 1 024 patches together with the progress-based priority; neutral in steady state)
#ifndef NF_STAGGER
#define NF_STAGGER 2
#endif
#ifndef NF_STAGGER_SLEEP
#define NF_STAGGER_SLEEP 16
#endif
    const bool early = !NF_STAGGER || PHILOX || blockIdx.x < (NF_STAGGER >= 2 ? (gridDim.x >> 2) : (gridDim.x >> 1));
    // NF_K_TILED (nf_device.h): the pixel -> address map changes from tile to tile, so the loads sit at the top of the
    // patch loop instead of one patch ahead
    bool tiled = TF;
    if constexpr (!FULL) tiled = (a.flags & NF_K_TILED) != 0;
    if ((int64_t)blockIdx.x < a.B && early && !tiled) {
        const size_t off0 = (size_t)blockIdx.x * (size_t)HW;
        if constexpr (!PHILOX) {
#pragma unroll
            for (int k = 0; k < PX; ++k)
                if (act[k]) zin[k] = reinterpret_cast<const float4 *>(a.in)[off0 + gidx[k]];
        }
        if constexpr (YPF) {
            if (sdn_first) {
#pragma unroll
                for (int k = 0; k < PX; ++k) yin[k] = reinterpret_cast<const float4 *>(a.y)[off0 + gidx[k]];
            }
        } else {
#ifndef NF_NO_WARM
#pragma unroll
        for (int k = 0; k < PX; k += NF_WARM_STEP)
            if (a.y && act[k]) warm += reinterpret_cast<const float *>(a.y)[4 * (off0 + gidx[k])];
#endif
        }
    }

    // zero both tiles once (the 1-pixel border is never written again) and stage the weight image, 16 bytes per lane
    {
#ifndef NF_ZERO_BORDER
#define NF_ZERO_BORDER 0
#endif
#ifndef NF_HB_ZB
#define NF_HB_ZB 1
#endif
        if constexpr (NF_ZERO_BORDER && BLK && !HALF) {
            // blocked fp32 tiles: every interior entry has an owner that writes it (z0 / h2 of each coupling) before anyone
            // reads it, so only the 2 (W + 2) + 2 H border entries need the zero — 3 KiB instead of 27 KiB of LDS stores per
            // 32x32 workgroup.  Measured and left OFF: + 0.7 % with a freshly initialised model (49.46 ->
            // 49.10 us per launch), - 2 .. 5 % with the shipped checkpoint, i.e. through bench.py (50.0 -> 52.5 us;
            // tools/ab_headline.py, tools/ab_bench.sh): the shorter set-up lets three quarters of the grid ask for their
            // first inputs almost at once, which is what NF_STAGGER exists to prevent
            const int nb = 2 * (W + 2) + 2 * H;
            for (int i = t; i < nb; i += THREADS) {
                int rp, cp;
                if (i < W + 2) {
                    rp = 0;
                    cp = i;
                } else if (i < 2 * (W + 2)) {
                    rp = H + 1;
                    cp = i - (W + 2);
                } else {
                    const int j = i - 2 * (W + 2);
                    rp = 1 + (j >> 1);
                    cp = (j & 1) ? W + 1 : 0;
                }
                const int e = (rp * 2 + (cp & 1)) * PW + (cp >> 1);
                t0[e] = make_float2(0.f, 0.f);
                *reinterpret_cast<float4 *>(th + (size_t)e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if constexpr (NF_HB_ZB && HB) {
            // every interior entry of both tiles has an owner that writes it before anyone reads it, and no operand read reaches
            // beyond column W + 1 of the padded rows: only the border ring needs the zero (3.5 KiB instead of 63 KiB at 64x64)
            const int nb = 2 * (W + 2) + 2 * H;
            for (int i = t; i < nb; i += THREADS) {
                int rp, cp;
                if (i < W + 2) {
                    rp = 0;
                    cp = i;
                } else if (i < 2 * (W + 2)) {
                    rp = H + 1;
                    cp = i - (W + 2);
                } else {
                    const int j = i - 2 * (W + 2);
                    rp = 1 + (j >> 1);
                    cp = (j & 1) ? W + 1 : 0;
                }
                t0h[rp * Wp + cp] = 0u;
                thh[rp * Wp + cp] = make_uint2(0u, 0u);
            }
        } else {
        const int nz4 = (tile_px * TILE_WORDS) >> 2;
        float4 *const s4 = reinterpret_cast<float4 *>(tiles);
        for (int i = t; i < nz4; i += THREADS) s4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = (nz4 << 2) + t; i < tile_px * TILE_WORDS; i += THREADS) tiles[i] = 0.0f;
        }
        if (MFMA) {
            const int np4 = a.n_params >> 2;   // every section of the matrix-core layouts is a multiple of 4 floats
            const float4 *const p4 = reinterpret_cast<const float4 *>(a.params);
            float4 *const w4 = reinterpret_cast<float4 *>(wl);
            if constexpr (NF_WSTAGE) {
#pragma unroll
                for (int u = 0; u < WU; ++u) {
                    const int i = t + u * THREADS;
                    if (i < np4) w4[i] = wreg[u];
                }
                for (int i = t + WU * THREADS; i < np4; i += THREADS) w4[i] = p4[i];   // (beyond the batch: models that barely fit)
            } else {
                for (int i = t; i < np4; i += THREADS) w4[i] = p4[i];
            }
            for (int i = (np4 << 2) + t; i < a.n_params; i += THREADS) wl[i] = a.params[i];
            if (t < 16) wl[ident_off + t] = (t >> 2) == (t & 3) ? 1.0f : 0.0f;
        }
    }
    __syncthreads();
    asm volatile("" ::"v"(warm));   // keep the warm-up loads
    if (NF_STAGGER && !early && (int64_t)blockIdx.x < a.B && !tiled) {
#if NF_STAGGER >= 2
        {   // quarters of the grid = the 4 workgroups of a CU: the 3rd and 4th wait a little longer still
            const int q = (int)((blockIdx.x * 4u) / gridDim.x);
            if (q == 2) __builtin_amdgcn_s_sleep(NF_STAGGER_SLEEP);
            if (q == 3) { __builtin_amdgcn_s_sleep(NF_STAGGER_SLEEP); __builtin_amdgcn_s_sleep(NF_STAGGER_SLEEP); }
        }
#endif
        const size_t off0 = (size_t)blockIdx.x * (size_t)HW;
        if constexpr (!PHILOX) {
#pragma unroll
            for (int k = 0; k < PX; ++k)
                if (act[k]) zin[k] = reinterpret_cast<const float4 *>(a.in)[off0 + gidx[k]];
        }
        if constexpr (YPF) {
            if (sdn_first) {
#pragma unroll
                for (int k = 0; k < PX; ++k) yin[k] = reinterpret_cast<const float4 *>(a.y)[off0 + gidx[k]];
            }
        }
    }
    if constexpr (BS) {
        // Pending re-fold (layers.py:388-391 + the BN-eval folding of fold_coupling): the previous launch of this call
        // gathered sum / sum of squares of one normalisation's input; every workgroup turns them into moments and
        // rescales ITS LDS copy of that layer (W[.][j] *= 1/sqrt(var_j + eps), B[j] = (B[j] - mean_j)/sqrt(var_j + eps));
        // workgroup 0 also writes the patched image to the parameter block the NEXT launch reads, and the moments.
        if (a.fix_stats) {
            double *const fsc = reinterpret_cast<double *>(bs_part + (THREADS / 64) * 8);   // [4] scale, then [4] mean
            if (t < 64) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = a.fix_stats[(size_t)t * 8 + q];   // slot t (NF_STATS_SLOTS == 64)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = wave_sum(v[q]);
                if (t < 4) {
                    const double m = v[t] / a.fix_n;
                    double var = v[4 + t] / a.fix_n - m * m;   // tf.nn.moments: population variance
                    if (var < 0.0) var = 0.0;
                    const float mf = (float)m, vf = (float)var;
                    fsc[t] = 1.0 / sqrt((double)vf + 1e-4);
                    fsc[4 + t] = (double)mf;
                    if (blockIdx.x == 0 && a.fix_mom_out) {
                        a.fix_mom_out[t] = mf;
                        a.fix_mom_out[4 + t] = vf;
                    }
                }
            }
            __syncthreads();
            float *const blk = wl + a.fix_off;
            if (a.fix_stage == 1) {
                for (int i = t; i < 96; i += THREADS) blk[NF2_CPL_W1T + i] = (float)((double)blk[NF2_CPL_W1T + i] * fsc[i / 24]);
                if (t < 4) blk[NF2_CPL_B1 + t] = (float)(((double)blk[NF2_CPL_B1 + t] - fsc[4 + t]) * fsc[t]);
            } else {
                for (int i = t; i < 16; i += THREADS) blk[NF2_CPL_W2T + i] = (float)((double)blk[NF2_CPL_W2T + i] * fsc[i / 4]);
                if (t < 4) blk[NF2_CPL_B2 + t] = (float)(((double)blk[NF2_CPL_B2 + t] - fsc[4 + t]) * fsc[t]);
            }
            __syncthreads();
            if (blockIdx.x == 0 && a.fix_params_out)
                for (int i = t; i < a.n_params; i += THREADS) a.fix_params_out[i] = wl[i];
        }
    }
    NF_STAMP(1);

#if NF_WAVE_PRIO
    if constexpr (MFMA && THREADS == 1024) {
        // one workgroup per CU, 4 waves per SIMD (waves w, w+4, w+8, w+12), all in the same phase between two barriers:
        // a fixed, distinct priority per wave of a SIMD lets the leaders' VALU tail run under the followers' MFMA burst
        const int wq = NF_WAVE_PRIO == 1 ? ((t >> 8) & 3) : NF_WAVE_PRIO == 2 ? ((t >> 8) & 1) : (((t >> 8) & 3) == 0 ? 1 : 0);
        if (wq == 0) __builtin_amdgcn_s_setprio(0);
        else if (wq == 1) __builtin_amdgcn_s_setprio(1);
        else if (wq == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
#endif
    const int n_ops = prog.n_ops;
    double acc_nll = 0.0, acc_sd = 0.0;   // thread 0 only
    const double inv_n = 1.0 / ((double)HW * 4.0);   // once per launch: two fp64 divisions per patch were a third of the epilogue
    [[maybe_unused]] int red_sel = 0;
    // Workgroups of 1 024 threads (one per CU, several patches each: 64x64 patches and tiles) finish the cross-wavefront sum of a
    // patch's results one patch LATE (see the epilogue): + 1.4 % at 64x64 fp16, same box; at 256 threads a workgroup sees 1 - 2
    // patches per launch and the extra barrier behind the loop costs what the deferral saves (- 0.3 %).  pend_b = the patch whose
    // wavefront partials sit in the other scratch set; synced = this patch's body has passed a barrier since they were written
    constexpr bool DEFER = THREADS == 1024;
    [[maybe_unused]] int64_t pend_b = -1;
    [[maybe_unused]] bool synced = false;
    [[maybe_unused]] const int fair_t1 = a.fair_t1, fair_t2 = a.fair_t2, fair_t3 = a.fair_t3;

    // The matrix-core kernels walk a run of `mix, coupling, mix, coupling, ...` (an `unc` layer = Conv2d1x1 + AffineCoupling,
    // noise_flow_model.py:79-104) as a COUNTED loop over parameter blocks a constant stride apart: the run is found once per launch
    // (here), so the loop body holds no scalar load — the op interpreter's loads of the next ops' type / offset were four
    // serialised scalar-cache round trips per coupling and wavefront, each one a parked wavefront (SQ_WAIT_ANY).  Only the first such
    // run is kept in registers (found on the host: nf_launch_flow); pairs outside it go one by one through the same loop with their
    // offsets loaded at its door.
    [[maybe_unused]] const int run_first = a.run_first, run_n = a.run_n, run_moff = a.run_moff, run_coff = a.run_coff, run_stride = a.run_stride,
                               run_type = a.run_type;

    // nll / sd / log-det of patch (tile) pb from its three sums: the wavefronts' partials in `rd`, or (one wavefront) r0 r1 r2
    auto finish_patch = [&](const float *rd, int64_t pb, float r0, float r1, float r2) {
        constexpr int NW = THREADS / 64;
        if constexpr (NW > 1) {
            if (t < 16) {   // one DPP row adds the wavefronts' partials: 3 reads + 12 adds instead of thread 0's chain of 3 NW
                static_assert(NW <= 16, "one lane per wavefront");
                r0 = row16_sum(t < NW ? rd[t] : 0.f);
                r1 = row16_sum(t < NW ? rd[NW + t] : 0.f);
                r2 = row16_sum(t < NW ? rd[2 * NW + t] : 0.f);
            }
        }
        if (tiled) {
            // the tile's share of its image's sums; nf_tile_combine_kernel forms nll / sd / log-det per image
            if (t == 0) *reinterpret_cast<float4 *>(a.tile_part + (size_t)pb * 4u) = make_float4(r0, r1, r2, 0.f);
        } else if (t == 0) {
            const double n = (double)HW * 4.0;
            const double logdet = (double)r0 + a.ld_const;
            // prior: sum -0.5*(log 2pi + z^2)   (noise_flow_model.py:537-539)
            double nll = -logdet;
            if (a.flags & NF_K_PRIOR) nll += 0.5 * n * 1.8378770664093453 + 0.5 * (double)r2;
            // sd of the base measure: population variance over the patch (noise_flow_model.py:477-478)
            const double mean = (double)r1 * inv_n;
            double var = (double)r2 * inv_n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            // the result is rounded to fp32 anyway: v_sqrt_f32 (1 ulp) instead of the ~60-instruction fp64 routine
            const double sd = (double)__builtin_amdgcn_sqrtf((float)var);
            if (a.nll_out) a.nll_out[pb] = (float)nll;
#ifndef NF_TIMELINE
            if (a.sd_out) a.sd_out[pb] = (float)sd;
#endif
            if (a.ld_out) a.ld_out[pb] = (float)logdet;
            acc_nll += (double)(float)nll;
            acc_sd += (double)(float)sd;
        }
    };

    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        synced = false;
        size_t patch_off = (size_t)b * (size_t)HW * 4u;
        [[maybe_unused]] int64_t patch_id = b;   // Philox key of the patch (NF_K_TILED: of the image the tile belongs to)
#ifdef NF_TIMELINE
        stamp_on = stamp_it++ == (a.B / gridDim.x) / 2;
#endif
        if constexpr (!FULL || TF) {
            if (tiled) {
                const int nt = a.tile_ny * a.tile_nx;
                const int64_t img = b / nt;
                const int ti = (int)(b - img * nt);
                const int ty = ti / a.tile_nx, tx = ti - ty * a.tile_nx;
                const int oy = nf_tile_origin(ty, a.img_H, H, a.tile_halo), ox = nf_tile_origin(tx, a.img_W, W, a.tile_halo);
                const int cy0 = nf_tile_core0(ty, a.img_H, H, a.tile_halo), cy1 = nf_tile_core1(ty, a.tile_ny, a.img_H, H, a.tile_halo);
                const int cx0 = nf_tile_core0(tx, a.img_W, W, a.tile_halo), cx1 = nf_tile_core1(tx, a.tile_nx, a.img_W, W, a.tile_halo);
                patch_off = (size_t)img * (size_t)a.img_H * (size_t)a.img_W * 4u;
                patch_id = img;
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    const int r = oy + prow[k], c = ox + pcol[k];
                    gidx[k] = act[k] ? r * a.img_W + c : 0;
                    bmask[k] = (r == 0 ? 1 : 0) | (r == a.img_H - 1 ? 2 : 0) | (c == 0 ? 4 : 0) | (c == a.img_W - 1 ? 8 : 0);
                    own[k] = act[k] && r >= cy0 && r < cy1 && c >= cx0 && c < cx1;
                }
                if constexpr (!PHILOX) {
                    const float4 *in4 = reinterpret_cast<const float4 *>(a.in + patch_off);
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (act[k]) v = in4[gidx[k]];
                        zin[k] = v;
                    }
                }
            }
        }

        // ---- prologue: the 4 channels of each owned pixel -> registers ----
        float z[PX][4];
        if (PHILOX) {
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                philox_normal4(a.seed, a.patch_base + patch_id, (uint32_t)gidx[k], NF_STREAM_SAMP, z[k]);
#pragma unroll
                for (int c = 0; c < 4; ++c) z[k][c] *= a.in_scale;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                z[k][0] = zin[k].x * a.in_scale;
                z[k][1] = zin[k].y * a.in_scale;
                z[k][2] = zin[k].z * a.in_scale;
                z[k][3] = zin[k].w * a.in_scale;
            }
        }

        float ld = 0.0f;    // this thread's share of the data-dependent log-det (natural log)
        if constexpr (!MFMA || BS) {
            if (a.flags & NF_K_CARRY_IN) ld = a.ld_carry[(size_t)b * THREADS + t];
        }
        float ld2 = 0.0f;   // ... and the part accumulated in log2 units (matrix-core couplings)
        [[maybe_unused]] int n_cpl = 0;
        [[maybe_unused]] int cpl_seen = 0;

        // Conv2d1x1 on the matrix cores: per-pixel z <- z @ M   (layers.py:108-124)
        [[maybe_unused]] auto mix_load = [&](int moff) { return *reinterpret_cast<const float4 *>(wl + moff + 4 * j4); };   // M[0..3][j4]
        [[maybe_unused]] auto mix_apply = [&](const float4 m) {
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                v4f acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m.x, z[k][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m.y, z[k][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m.z, z[k][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m.w, z[k][3], acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[k][j] = acc[j];
            }
        };
        [[maybe_unused]] auto mix_mfma = [&](int mop) { mix_apply(mix_load(prog.ops[mop].off)); };

        // AffineCouplingSdnEx5 and its relatives: scale = sqrt(beta1*y/gain + beta2)  (cond_utils.py:238)
        auto sdn_apply = [&](int stype, int slot, const float4 (&yv)[PX]) {
            const float ck1 = a.cond_a[slot & 3], cb2 = a.cond_b[slot & 3];
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const float yy[4] = {yv[k].x, yv[k].y, yv[k].z, yv[k].w};
                // scale = sqrt(v), v = beta1*y/gain + beta2 > 0: z/scale = z*rsq(v) (v_rsq_f32: 1 ulp), and the pixel's share of the
                // log-det, -sum_c log scale_c = ln2 * log2(prod_c rsq(v_c)): ONE v_log_f32 per pixel on the product of the four
                // reciprocal roots (3 multiplies instead of 3 more quarter-rate logarithms; the product stays finite while every
                // v_c > 1e-18, i.e. a noise sd above 1e-9 — variances of raw images in [0, 1] are 1e-7 .. 1e-2)
#ifndef NF_SDN_1LOG
#define NF_SDN_1LOG 1
#endif
                float rp = 1.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = fmaf(yy[c], ck1, cb2);
                    if (stype == NF_OP_SDN_DIV) {
                        const float r = __builtin_amdgcn_rsqf(v);
                        z[k][c] = z[k][c] * r;
                        if (NF_SDN_1LOG) rp = c == 0 ? r : rp * r;
                        else if (own[k]) ld = fmaf(-0.34657359027997264f, __builtin_amdgcn_logf(v), ld);
                    } else {
                        z[k][c] = z[k][c] * __builtin_amdgcn_sqrtf(v);
                    }
                }
                if (NF_SDN_1LOG && stype == NF_OP_SDN_DIV && own[k]) ld = fmaf(0.6931471805599453f, __builtin_amdgcn_logf(rp), ld);
            }
        };
        int op_begin = 0;
        if constexpr (YPF) {
            if (sdn_first) {   // the leading sdn layer on the prefetched clean image
                float4 yv[PX];
#pragma unroll
                for (int k = 0; k < PX; ++k) yv[k] = yin[k];
                sdn_apply(NF_OP_SDN_DIV, prog.ops[0].off, yv);
                op_begin = 1;
            }
        }

        for (int op = op_begin; op < n_ops; ++op) {
            int type = prog.ops[op].type;
            // matrix-core kernels: the counted loop over (mix, coupling) pairs that starts at this op — n pairs, their blocks at
            // moff / coff + i * stride floats, one coupling direction; a coupling without a mix in front takes the identity block
            [[maybe_unused]] int pr_n = 1, pr_moff = 0, pr_coff = 0, pr_stride = 0;
            if constexpr (MFMA) {
                if (type == NF_OP_MIX) {
#ifdef NF_TIMELINE
                    if (n_cpl == 0) { asm volatile("" ::"v"(z[0][0])); NF_STAMP(2); }   // inputs have arrived (first use after the sdn layer)
#endif
                    if (op == run_first) {
                        pr_n = run_n; pr_moff = run_moff; pr_coff = run_coff; pr_stride = run_stride;
                        type = run_type;
                    } else {
                        const int nt = op + 1 < n_ops ? prog.ops[op + 1].type : 0;
                        if (nt != NF_OP_COUPLING_FWD && nt != NF_OP_COUPLING_REV) {   // a mix on its own
                            mix_mfma(op);
                            continue;
                        }
                        pr_moff = prog.ops[op].off;
                        pr_coff = prog.ops[op + 1].off;
                        type = nt;
                    }
                    ++op;   // the coupling of the first pair
                } else if (type == NF_OP_COUPLING_FWD || type == NF_OP_COUPLING_REV) {
                    pr_moff = ident_off;
                    pr_coff = prog.ops[op].off;
                }
            }
            [[maybe_unused]] int coff = prog.ops[op].off;   // parameter block of the op (of the current coupling inside the pair loop)
            cfloat_p P = (cfloat_p)(a.params + coff);   // wave-uniform, scalar loads

            if (type == NF_OP_MIX) {
                // Conv2d1x1 on the scalar-weight kernel
                if constexpr (!MFMA) {
                    float m[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) m[i] = P[i];
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float s = z[k][0] * m[j];
                            s = fmaf(z[k][1], m[4 + j], s);
                            s = fmaf(z[k][2], m[8 + j], s);
                            s = fmaf(z[k][3], m[12 + j], s);
                            o[j] = s;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[k][j] = o[j];
                    }
                }
            } else if (type == NF_OP_COUPLING_FWD || type == NF_OP_COUPLING_REV) {
                // ---- AffineCoupling (layers.py:275-291 / 355-375) ----
                // A run of couplings with a mix in front of each (unc | unc | ...) stays in this inner loop: with the pointwise
                // layers' code paths out of the way z has ONE set of registers across the run — the mix accumulates into it, the
                // coupling updates its transformed half in place — instead of being copied between the loop-carried set and a
                // working set around every op (28 v_mov_b64 per mix + coupling, a fifth of the VALU instructions)
                bool patch_done = false;
                synced = true;   // every coupling passes barriers (the exchange of the pass-through half and of h2)
                if constexpr (MFMA) op -= 2;
                // NF_MIX_AHEAD 1: the mix matrix of a pair is read one coupling ahead (in front of the previous coupling's second
                // barrier) instead of at the head of the loop body, where the read sits between the affine stage and the mix's first
                // MFMA.  Measured neutral (fp32 32x32: 2.131e7 / 2.130e7; fp16 64x64 within the 1 % run-to-run band): left off
#ifndef NF_MIX_AHEAD
#define NF_MIX_AHEAD 0
#endif
                [[maybe_unused]] float4 mix_m = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (MFMA) mix_m = mix_load(pr_moff);
#pragma nounroll
                for (int pr_i = 0; pr_i < pr_n; ++pr_i) {
                if constexpr (MFMA) {
                    if (!NF_MIX_AHEAD) mix_m = mix_load(pr_moff);
                    mix_apply(mix_m);
                    coff = pr_coff;
                    pr_moff += pr_i + 1 < pr_n ? pr_stride : 0;   // (the last pair re-reads its own matrix: always inside the image)
                    pr_coff += pr_stride;
                    op += 2;
                }
#if NF_FAIR
                if constexpr (MFMA && !(NF_WAVE_PRIO && THREADS == 1024)) {
                    // lvl = (cpl_seen * 4) / cpl_total, 0 .. 3, wave-uniform — against thresholds formed once per launch (the division
                    // was ~30 scalar instructions per coupling)
                    const int lvl = (cpl_seen >= fair_t1 ? 1 : 0) + (cpl_seen >= fair_t2 ? 1 : 0) + (cpl_seen >= fair_t3 ? 1 : 0);
                    ++cpl_seen;
                    if (lvl == 0) __builtin_amdgcn_s_setprio(3);
                    else if (lvl == 1) __builtin_amdgcn_s_setprio(2);
                    else if (lvl == 2) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
                // HB: every weight operand of the coupling is requested before the barriers, so that after a barrier only the
                // tile reads stand between a wavefront and its matrix instructions (NF_HB_EARLY=0: at first use, A/B aid)
#ifndef NF_HB_EARLY
#define NF_HB_EARLY 1
#endif
                [[maybe_unused]] v8h hb_a1, hb_a3a, hb_a3b, hb_a2a, hb_a2b;
                [[maybe_unused]] v4h hb_w2h;
                // l_2 on v_mfma_f32_16x16x32_f16 as well (nf_device.h, NF11_CPL_A2; 64x64 layouts only)
#ifndef NF_HB_L2BIG
#define NF_HB_L2BIG 1
#endif
                constexpr bool L2B = NF_HB_L2BIG && HB && THREADS == 1024;
                [[maybe_unused]] float4 hb_b1, hb_b2, hb_e[PX];
                if constexpr (HB) {
                    const float *wb = wl + coff;
                    const uint32_t *wbw = reinterpret_cast<const uint32_t *>(wb);
                    if (NF_HB_EARLY) {
                        hb_b1 = *reinterpret_cast<const float4 *>(wb + NF11_CPL_B1);
                        hb_b2 = *reinterpret_cast<const float4 *>(wb + NF11_CPL_B2);
                        hb_a1 = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A1 + (t & 63) * 4));
                        if constexpr (L2B) {
                            hb_a2a = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A2 + (t & 63) * 4));
                            hb_a2b = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A2 + 256 + (t & 63) * 4));
                        } else {
                            hb_w2h = *reinterpret_cast<const v4h *>(wbw + NF11_CPL_W2H + j4 * 2);
                        }
                    }
                }
                // 1) publish the pass-through half
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    if constexpr (HALF) {
                        const v2h zh = {(_Float16)z[k][0], (_Float16)z[k][1]};
                        t0h[lidx[k]] = __builtin_bit_cast(uint32_t, zh);
                    } else {
                        if (act[k]) t0[lidx[k]] = make_float2(z[k][0], z[k][1]);
                    }
                }
                __syncthreads();

                // 2) l_1 (3x3 SAME, BN folded) -> ReLU -> l_2 (1x1, BN folded) -> ReLU
                if constexpr (HB) {
                    const float *wb = wl + coff;
                    const uint32_t *wbw = reinterpret_cast<const uint32_t *>(wb);
                    if (!NF_HB_EARLY) {
                        hb_b1 = *reinterpret_cast<const float4 *>(wb + NF11_CPL_B1);
                        hb_b2 = *reinterpret_cast<const float4 *>(wb + NF11_CPL_B2);
                        hb_a1 = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A1 + (t & 63) * 4));
                        if constexpr (L2B) {
                            hb_a2a = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A2 + (t & 63) * 4));
                            hb_a2b = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A2 + 256 + (t & 63) * 4));
                        } else {
                            hb_w2h = *reinterpret_cast<const v4h *>(wbw + NF11_CPL_W2H + j4 * 2);
                        }
                    }
                    const float4 b1 = hb_b1, b2 = hb_b2;
                    const v8h a1 = hb_a1;
                    [[maybe_unused]] v4h w2h = {0, 0, 0, 0};
                    if constexpr (!L2B) w2h = hb_w2h;
                    // the four units of the wavefront side by side: every stage's dependent latency (LDS, the 4-pass MFMA, the
                    // conversions) is covered by the same stage of the other three
                    v8h bop[PX];
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        // this lane's K slot: one window row, 4 columns x 2 channels (two 8-byte reads, 8-byte aligned)
                        const uint2 lo = *reinterpret_cast<const uint2 *>(t0h + wbase + 2 * k * Wp);
                        const uint2 hi = *reinterpret_cast<const uint2 *>(t0h + wbase + 2 * k * Wp + 2);
                        bop[k] = __builtin_bit_cast(v8h, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    }
                    v4f h1[PX];
#pragma unroll
                    for (int k = 0; k < PX; ++k)
                        h1[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bop[k], v4f{b1.x, b1.y, b1.z, b1.w}, 0, 0, 0);
                    v4h r1[PX];
#pragma unroll
                    for (int k = 0; k < PX; ++k)
                        r1[k] = __builtin_elementwise_max(
                            v4h{(_Float16)h1[k][0], (_Float16)h1[k][1], (_Float16)h1[k][2], (_Float16)h1[k][3]}, v4h{0, 0, 0, 0});
                    v4f h2[PX];
                    if constexpr (L2B) {
                        // units 2u and 2u + 1 share one B operand (the lane's two pixels, 8 halves); the A operand picks the unit
#pragma unroll
                        for (int u = 0; u < PX / 2; ++u) {
                            const v8h bb = __builtin_shufflevector(r1[2 * u], r1[2 * u + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                            h2[2 * u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb_a2a, bb, v4f{b2.x, b2.y, b2.z, b2.w}, 0, 0, 0);
                            h2[2 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb_a2b, bb, v4f{b2.x, b2.y, b2.z, b2.w}, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                    for (int k = 0; k < PX; ++k)
                        h2[k] = __builtin_amdgcn_mfma_f32_4x4x4f16(w2h, r1[k], v4f{b2.x, b2.y, b2.z, b2.w}, 0, 0, 0);
                    }
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        const v4h r2 = __builtin_elementwise_max(
                            v4h{(_Float16)h2[k][0], (_Float16)h2[k][1], (_Float16)h2[k][2], (_Float16)h2[k][3]}, v4h{0, 0, 0, 0});
                        thh[lidx[k]] = __builtin_bit_cast(uint2, r2);
                    }
                    if (NF_HB_EARLY) {   // l_last's weights and border-table rows: in flight across the barrier
                        hb_a3a = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A3 + (t & 63) * 4));
                        hb_a3b = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A3 + 256 + (t & 63) * 4));
#pragma unroll
                        for (int k = 0; k < PX; ++k) hb_e[k] = *reinterpret_cast<const float4 *>(wb + NF11_CPL_E + 4 * bmask[k]);
                    }
                } else if constexpr (H16) {
                    const float *wb = wl + coff;
                    const uint32_t *wbw = reinterpret_cast<const uint32_t *>(wb);
                    const float4 b1 = *reinterpret_cast<const float4 *>(wb + NF3_CPL_B1);
                    const float4 b2 = *reinterpret_cast<const float4 *>(wb + NF3_CPL_B2);
                    v4h w1h[3][4];   // [filter row][group]: this lane's A operands (4 halves each)
                    // 16-byte LDS reads (two A operands each): ds_read_b128 moves 256 B/clk, the ds_read2_b64 the compiler
                    // would form from 8-byte reads only 128 B/clk
#pragma unroll
                    for (int di = 0; di < 3; ++di)
#pragma unroll
                        for (int g = 0; g < 4; g += 2) {
                            const uint4 q = *reinterpret_cast<const uint4 *>(wbw + NF3_CPL_W1H + ((j4 * 3 + di) * 4 + g) * 2);
                            w1h[di][g] = __builtin_bit_cast(v4h, make_uint2(q.x, q.y));
                            w1h[di][g + 1] = __builtin_bit_cast(v4h, make_uint2(q.z, q.w));
                        }
                    const v4h w2h = *reinterpret_cast<const v4h *>(wbw + NF3_CPL_W2H + j4 * 2);
                    v4f h1[PX];
#pragma unroll
                    for (int k = 0; k < PX; ++k) h1[k] = v4f{b1.x, b1.y, b1.z, b1.w};
                    NF_PRIO_UP();
#pragma unroll
                    for (int wr = 0; wr < 4; ++wr) {
                        // one window row = two 8-byte pairs of horizontally adjacent pixels (2 ch each)
                        const v4h p01 = *reinterpret_cast<const v4h *>(t0h + wbase + wr * Wp);
                        const v4h p23 = *reinterpret_cast<const v4h *>(t0h + wbase + wr * Wp + 2);
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy) {
                            const int di = wr - dy;
                            if (di < 0 || di > 2) continue;
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                const int k = dy * 2 + dx;
                                h1[k] = __builtin_amdgcn_mfma_f32_4x4x4f16(w1h[di][2 * dx + 0], p01, h1[k], 0, 0, 0);
                                h1[k] = __builtin_amdgcn_mfma_f32_4x4x4f16(w1h[di][2 * dx + 1], p23, h1[k], 0, 0, 0);
                            }
                        }
                    }
                    NF_PRIO_DOWN();
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        // ReLU after the conversion, two halves per instruction (v_pk_max_f16)
                        const v4h a1 = __builtin_elementwise_max(
                            v4h{(_Float16)h1[k][0], (_Float16)h1[k][1], (_Float16)h1[k][2], (_Float16)h1[k][3]}, v4h{0, 0, 0, 0});
                        const v4f h2 = __builtin_amdgcn_mfma_f32_4x4x4f16(w2h, a1, v4f{b2.x, b2.y, b2.z, b2.w}, 0, 0, 0);
                        const v4h a2 = __builtin_elementwise_max(
                            v4h{(_Float16)h2[0], (_Float16)h2[1], (_Float16)h2[2], (_Float16)h2[3]}, v4h{0, 0, 0, 0});
                        thh[lidx[k]] = __builtin_bit_cast(uint2, a2);
                    }
                } else if constexpr (MFMA) {
                    const float *wb = wl + coff;
                    const float4 b1 = *reinterpret_cast<const float4 *>(wb + NF2_CPL_B1);
                    const float4 b2 = *reinterpret_cast<const float4 *>(wb + NF2_CPL_B2);
                    const float4 w2 = *reinterpret_cast<const float4 *>(wb + NF2_CPL_W2T + 4 * j4);
                    v4f h1[PX];
#pragma unroll
                    for (int k = 0; k < PX; ++k) h1[k] = v4f{b1.x, b1.y, b1.z, b1.w};
                    if constexpr (BLK) {
                        float w1[3][6];   // this lane's A operands: W1[(di,dj,c)][j4]
#pragma unroll
                        for (int di = 0; di < 3; ++di) {
                            const float4 wA = *reinterpret_cast<const float4 *>(wb + NF2_CPL_W1T + 24 * j4 + 8 * di);
                            const float2 wB = *reinterpret_cast<const float2 *>(wb + NF2_CPL_W1T + 24 * j4 + 8 * di + 4);
                            w1[di][0] = wA.x; w1[di][1] = wA.y; w1[di][2] = wA.z; w1[di][3] = wA.w; w1[di][4] = wB.x; w1[di][5] = wB.y;
                        }
                        NF_PRIO_UP();
#pragma unroll
                        for (int wr = 0; wr < 4; ++wr) {
                            float2 v[4];   // one row of the 4x4 window, shared by the 2x2 output pixels
#pragma unroll
                            for (int wc = 0; wc < 4; ++wc) v[wc] = t0[wbase + (wr * 2 + (wc & 1)) * PW + (wc >> 1)];
#pragma unroll
                            for (int dy = 0; dy < 2; ++dy) {
                                const int di = wr - dy;
                                if (di < 0 || di > 2) continue;
#pragma unroll
                                for (int dj = 0; dj < 3; ++dj)
#pragma unroll
                                    for (int dx = 0; dx < 2; ++dx) {
                                        const int k = dy * 2 + dx;
                                        h1[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1[di][2 * dj + 0], v[dx + dj].x, h1[k], 0, 0, 0);
                                        h1[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1[di][2 * dj + 1], v[dx + dj].y, h1[k], 0, 0, 0);
                                    }
                            }
                        }
                    } else {
                    NF_PRIO_UP();
#pragma unroll
                    for (int di = 0; di < 3; ++di) {
                        // this lane's A operands of one filter row: W1[(di,dj,c)][j4]
                        const float4 wA = *reinterpret_cast<const float4 *>(wb + NF2_CPL_W1T + 24 * j4 + 8 * di);
                        const float2 wB = *reinterpret_cast<const float2 *>(wb + NF2_CPL_W1T + 24 * j4 + 8 * di + 4);
                        const float wr[6] = {wA.x, wA.y, wA.z, wA.w, wB.x, wB.y};
#pragma unroll
                        for (int dj = 0; dj < 3; ++dj) {
                            const int doff = (di - 1) * Wp + (dj - 1);
#pragma unroll
                            for (int k = 0; k < PX; ++k) {
                                const float2 v = t0[lidx[k] + doff];
                                h1[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[2 * dj + 0], v.x, h1[k], 0, 0, 0);
                                h1[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[2 * dj + 1], v.y, h1[k], 0, 0, 0);
                            }
                        }
                    }
                    }
                    NF_PRIO_DOWN();
                    [[maybe_unused]] int bs_stage = 0;
                    [[maybe_unused]] float bs1[4] = {0.f, 0.f, 0.f, 0.f}, bs2[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (BS) bs_stage = (a.stats && op == a.stats_op) ? a.stats_stage : 0;
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        if constexpr (BS) {
                            if (bs_stage == 1) {
                                const float h1k[4] = {h1[k][0], h1[k][1], h1[k][2], h1[k][3]};
                                stats_accumulate<4>(h1k, own[k], bs1, bs2);
                            }
                        }
                        v4f h2 = {b2.x, b2.y, b2.z, b2.w};
                        h2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w2.x, nf_relu(h1[k][0]), h2, 0, 0, 0);
                        h2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w2.y, nf_relu(h1[k][1]), h2, 0, 0, 0);
                        h2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w2.z, nf_relu(h1[k][2]), h2, 0, 0, 0);
                        h2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w2.w, nf_relu(h1[k][3]), h2, 0, 0, 0);
                        if constexpr (BS) {
                            if (bs_stage == 2) {
                                const float h2k[4] = {h2[0], h2[1], h2[2], h2[3]};
                                stats_accumulate<4>(h2k, own[k], bs1, bs2);
                            }
                        }
                        if (act[k])
                            *reinterpret_cast<float4 *>(th + (size_t)lidx[k] * 4) =
                                make_float4(nf_relu(h2[0]), nf_relu(h2[1]), nf_relu(h2[2]), nf_relu(h2[3]));
                    }
                    if constexpr (BS) {
                        // wavefront sums -> LDS; after the barrier below 8 lanes add the workgroup's totals to its slot
                        // (one fp64 atomic per value per WORKGROUP per patch: same-line atomics serialise at ~10 ns)
                        if (bs_stage) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float sa = wave_sum(bs1[j]), sb = wave_sum(bs2[j]);
                                if ((t & 63) == 0) {
                                    bs_part[(t >> 6) * 8 + j] = sa;
                                    bs_part[(t >> 6) * 8 + 4 + j] = sb;
                                }
                            }
                        }
                    }
                } else
                {
                    const cfloat_p W1 = P + nf_cpl_off_W1(WIDTH);
                    const cfloat_p B1 = P + nf_cpl_off_B1(WIDTH);
                    const cfloat_p W2 = P + nf_cpl_off_W2(WIDTH);
                    const cfloat_p B2 = P + nf_cpl_off_B2(WIDTH);
                    float h1[PX][WIDTH];
#pragma unroll
                    for (int k = 0; k < PX; ++k)
#pragma unroll
                        for (int j = 0; j < WIDTH; ++j) h1[k][j] = B1[j];
                    // one filter row per (rolled) iteration: bounds the live scalar weights
#pragma unroll 1
                    for (int di = 0; di < 3; ++di) {
                        const cfloat_p W1r = W1 + di * (3 * 2 * WIDTH);
                        const int roff = (di - 1) * Wp - 1;
#pragma unroll
                        for (int dj = 0; dj < 3; ++dj) {
#pragma unroll
                            for (int k = 0; k < PX; ++k) {
                                const float2 v = t0[lidx[k] + roff + dj];
#pragma unroll
                                for (int j = 0; j < WIDTH; ++j) {
                                    h1[k][j] = fmaf(v.x, W1r[(dj * 2 + 0) * WIDTH + j], h1[k][j]);
                                    h1[k][j] = fmaf(v.y, W1r[(dj * 2 + 1) * WIDTH + j], h1[k][j]);
                                }
                            }
                        }
                    }
                    // batch-statistics pass: the host folded an identity normalisation into the layer
                    // under measurement, so h1 / h2 here ARE its pre-normalisation activations
                    const int stats_stage = (a.stats && op == a.stats_op) ? a.stats_stage : 0;
                    double *const stats = a.stats + (blockIdx.x & (NF_STATS_SLOTS - 1)) * (2 * WIDTH);
                    // several pixels per lane: sum them in registers first (one reduction + atomics per patch, not per pixel)
                    constexpr int SW = PX > 1 ? WIDTH : 1;
                    float st1[SW], st2[SW];
#pragma unroll
                    for (int j = 0; j < SW; ++j) st1[j] = st2[j] = 0.0f;
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        if constexpr (PX > 1) {
                            if (stats_stage == 1) stats_accumulate<WIDTH>(h1[k], own[k], st1, st2);
                        } else {
                            if (stats_stage == 1) stats_direct<WIDTH>(h1[k], own[k], stats, t);
                        }
                        float h2[WIDTH];
#pragma unroll
                        for (int j = 0; j < WIDTH; ++j) h2[j] = B2[j];
#pragma unroll
                        for (int i = 0; i < WIDTH; ++i) {
                            const float hi = nf_relu(h1[k][i]);
#pragma unroll
                            for (int j = 0; j < WIDTH; ++j) h2[j] = fmaf(hi, W2[i * WIDTH + j], h2[j]);
                        }
                        if constexpr (PX > 1) {
                            if (stats_stage == 2) stats_accumulate<WIDTH>(h2, own[k], st1, st2);
                        } else {
                            if (stats_stage == 2) stats_direct<WIDTH>(h2, own[k], stats, t);
                        }
                        if (act[k]) {
                            float4 *dst = reinterpret_cast<float4 *>(th + (size_t)lidx[k] * WIDTH);
#pragma unroll
                            for (int q = 0; q < WIDTH / 4; ++q)
                                dst[q] = make_float4(nf_relu(h2[4 * q + 0]), nf_relu(h2[4 * q + 1]),
                                                     nf_relu(h2[4 * q + 2]), nf_relu(h2[4 * q + 3]));
                        }
                    }
                    if constexpr (PX > 1) {
                        if (stats_stage) stats_flush<WIDTH>(st1, st2, stats, t);
                    }
                }
                if constexpr (MFMA && NF_MIX_AHEAD) mix_m = mix_load(pr_moff);
                __syncthreads();
                if constexpr (BS) {
                    if (a.stats && op == a.stats_op && t < 8) {
                        double tot = 0.0;
#pragma unroll
                        for (int wv = 0; wv < THREADS / 64; ++wv) tot += (double)bs_part[wv * 8 + t];
                        atomicAdd(&a.stats[(blockIdx.x & (NF_STATS_SLOTS - 1)) * 8 + t], tot);
                    }
                }
                if constexpr (!MFMA || BS) {
                    if (a.stats && op == a.stats_op) {   // statistics gathered: this patch is done
                        patch_done = true;
                        break;
                    }
                }

                // 3) l_last (zero pad + border-indicator channel, 3x3 VALID, *exp(3 logs) folded)
                {
                    float o[PX][4];
                    float sc;
                    if constexpr (HB) {
                        const float *wb = wl + coff;
                        const uint32_t *wbw = reinterpret_cast<const uint32_t *>(wb);
                        sc = 0.0f;
                        if (!NF_HB_EARLY) {
                            hb_a3a = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A3 + (t & 63) * 4));
                            hb_a3b = __builtin_bit_cast(v8h, *reinterpret_cast<const uint4 *>(wbw + NF11_CPL_A3 + 256 + (t & 63) * 4));
#pragma unroll
                            for (int k = 0; k < PX; ++k) hb_e[k] = *reinterpret_cast<const float4 *>(wb + NF11_CPL_E + 4 * bmask[k]);
                        }
                        const v8h a3a = hb_a3a, a3b = hb_a3b;
                        // K slot of this lane: two adjacent window pixels x 4 channels = one aligned 16-byte read per instruction.
                        // NF_HB_RDAHEAD units' reads are requested before the first matrix instruction.  Left to itself (0) the compiler
                        // reads two operands, waits, multiplies, and only then reads the next two — four LDS latencies per phase;
                        // requesting all eight first (4) needs the whole 128-register budget of a 16-wavefront workgroup and measured
                        // SLOWER (same box, 64x64 fp16, B = 1 024: 1.226e7 against 1.258e7 patches/s; 2: 1.241e7): left at 0
#ifndef NF_HB_RDAHEAD
#define NF_HB_RDAHEAD 0
#endif
                        uint4 q0[PX], q1[PX];
#pragma unroll
                        for (int k = 0; k < PX && k < NF_HB_RDAHEAD; ++k) {
                            q0[k] = *reinterpret_cast<const uint4 *>(thh + wbase3 + 2 * k * Wp);
                            q1[k] = *reinterpret_cast<const uint4 *>(thh + wbase3 + (2 * k + 1) * Wp);
                        }
                        if (NF_HB_RDAHEAD) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < PX; ++k) {
                            const float4 e = hb_e[k];
                            if (k >= NF_HB_RDAHEAD) {
                                q0[k] = *reinterpret_cast<const uint4 *>(thh + wbase3 + 2 * k * Wp);
                                q1[k] = *reinterpret_cast<const uint4 *>(thh + wbase3 + (2 * k + 1) * Wp);
                            }
                            v4f acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3a, __builtin_bit_cast(v8h, q0[k]), v4f{e.x, e.y, e.z, e.w}, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3b, __builtin_bit_cast(v8h, q1[k]), acc, 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[k][j] = acc[j];
                        }
                    } else if constexpr (H16) {
                        const float *wb = wl + coff;
                        const uint32_t *wbw = reinterpret_cast<const uint32_t *>(wb);
                        sc = 0.0f;
                        v4h w3h[9];
#pragma unroll
                        for (int tap = 0; tap < 8; tap += 2) {
                            const uint4 q = *reinterpret_cast<const uint4 *>(wbw + NF3_CPL_W3H + j4 * NF3_W3H_STRIDE + tap * 2);
                            w3h[tap] = __builtin_bit_cast(v4h, make_uint2(q.x, q.y));
                            w3h[tap + 1] = __builtin_bit_cast(v4h, make_uint2(q.z, q.w));
                        }
                        w3h[8] = *reinterpret_cast<const v4h *>(wbw + NF3_CPL_W3H + j4 * NF3_W3H_STRIDE + 16);
                        v4f acc[PX];
#pragma unroll
                        for (int k = 0; k < PX; ++k) {
                            const float4 e = *reinterpret_cast<const float4 *>(wb + NF3_CPL_E + 4 * bmask[k]);
                            acc[k] = v4f{e.x, e.y, e.z, e.w};
                        }
                        NF_PRIO_UP();
#pragma unroll
                        for (int wr = 0; wr < 4; ++wr) {
                            const uint4 q01 = *reinterpret_cast<const uint4 *>(thh + wbase + wr * Wp);
                            const uint4 q23 = *reinterpret_cast<const uint4 *>(thh + wbase + wr * Wp + 2);
                            const v4h hv[4] = {__builtin_bit_cast(v4h, make_uint2(q01.x, q01.y)), __builtin_bit_cast(v4h, make_uint2(q01.z, q01.w)),
                                               __builtin_bit_cast(v4h, make_uint2(q23.x, q23.y)), __builtin_bit_cast(v4h, make_uint2(q23.z, q23.w))};
#pragma unroll
                            for (int dy = 0; dy < 2; ++dy) {
                                const int di = wr - dy;
                                if (di < 0 || di > 2) continue;
#pragma unroll
                                for (int dj = 0; dj < 3; ++dj)
#pragma unroll
                                    for (int dx = 0; dx < 2; ++dx) {
                                        const int k = dy * 2 + dx;
                                        acc[k] = __builtin_amdgcn_mfma_f32_4x4x4f16(w3h[di * 3 + dj], hv[dx + dj], acc[k], 0, 0, 0);
                                    }
                            }
                        }
                        NF_PRIO_DOWN();
#pragma unroll
                        for (int k = 0; k < PX; ++k)
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[k][j] = acc[k][j];
                    } else if constexpr (MFMA) {
                        const float *wb = wl + coff;
                        sc = 0.0f;
                        v4f acc[PX];
#pragma unroll
                        for (int k = 0; k < PX; ++k) {
                            const float4 e = *reinterpret_cast<const float4 *>(wb + NF2_CPL_E + 4 * bmask[k]);
                            acc[k] = v4f{e.x, e.y, e.z, e.w};
                        }
                        if constexpr (BLK) {
                            float w3[3][12];   // this lane's A operands: W3[(di,dj,i)][j4]
#pragma unroll
                            for (int q = 0; q < 9; ++q) {
                                const float4 w = *reinterpret_cast<const float4 *>(wb + NF2_CPL_W3T + 36 * j4 + 4 * q);
                                w3[q / 3][4 * (q % 3) + 0] = w.x; w3[q / 3][4 * (q % 3) + 1] = w.y;
                                w3[q / 3][4 * (q % 3) + 2] = w.z; w3[q / 3][4 * (q % 3) + 3] = w.w;
                            }
                            NF_PRIO_UP();
#pragma unroll
                            for (int wr = 0; wr < 4; ++wr) {
                                float4 hv[4];   // one row of the 4x4 window of h2
#pragma unroll
                                for (int wc = 0; wc < 4; ++wc)
                                    hv[wc] = *reinterpret_cast<const float4 *>(th + (size_t)(wbase + (wr * 2 + (wc & 1)) * PW + (wc >> 1)) * 4);
#pragma unroll
                                for (int dy = 0; dy < 2; ++dy) {
                                    const int di = wr - dy;
                                    if (di < 0 || di > 2) continue;
#pragma unroll
                                    for (int dj = 0; dj < 3; ++dj)
#pragma unroll
                                        for (int dx = 0; dx < 2; ++dx) {
                                            const int k = dy * 2 + dx;
                                            const float4 h = hv[dx + dj];
                                            acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[di][4 * dj + 0], h.x, acc[k], 0, 0, 0);
                                            acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[di][4 * dj + 1], h.y, acc[k], 0, 0, 0);
                                            acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[di][4 * dj + 2], h.z, acc[k], 0, 0, 0);
                                            acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[di][4 * dj + 3], h.w, acc[k], 0, 0, 0);
                                        }
                                }
                            }
                        } else {
                        NF_PRIO_UP();
#pragma unroll
                        for (int di = 0; di < 3; ++di) {
                            float w3[12];   // this lane's A operands of one filter row: W3[(di,dj,i)][j4]
#pragma unroll
                            for (int q = 0; q < 3; ++q) {
                                const float4 w = *reinterpret_cast<const float4 *>(wb + NF2_CPL_W3T + 36 * j4 + 12 * di + 4 * q);
                                w3[4 * q + 0] = w.x; w3[4 * q + 1] = w.y; w3[4 * q + 2] = w.z; w3[4 * q + 3] = w.w;
                            }
#pragma unroll
                            for (int dj = 0; dj < 3; ++dj) {
                                const int doff = (di - 1) * Wp + (dj - 1);
#pragma unroll
                                for (int k = 0; k < PX; ++k) {
                                    const float4 hv = *reinterpret_cast<const float4 *>(th + (size_t)(lidx[k] + doff) * 4);
                                    acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[4 * dj + 0], hv.x, acc[k], 0, 0, 0);
                                    acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[4 * dj + 1], hv.y, acc[k], 0, 0, 0);
                                    acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[4 * dj + 2], hv.z, acc[k], 0, 0, 0);
                                    acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(w3[4 * dj + 3], hv.w, acc[k], 0, 0, 0);
                                }
                            }
                        }
                        }
                        NF_PRIO_DOWN();
#pragma unroll
                        for (int k = 0; k < PX; ++k)
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[k][j] = acc[k][j];
                    } else {
                    const cfloat_p W3 = P + nf_cpl_off_W3(WIDTH);
                    sc = P[nf_cpl_off_S(WIDTH)];
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        // bias + the indicator-channel taps that fall outside the image (per-lane row)
                        const float4 e = *reinterpret_cast<const float4 *>(a.params + coff +
                                                                           nf_cpl_off_E(WIDTH) + 4 * bmask[k]);
                        o[k][0] = e.x; o[k][1] = e.y; o[k][2] = e.z; o[k][3] = e.w;
                    }
#pragma unroll 1
                    for (int di = 0; di < 3; ++di) {
                        const cfloat_p W3r = W3 + di * (3 * WIDTH * 4);
                        const int roff = (di - 1) * Wp - 1;
#pragma unroll
                        for (int dj = 0; dj < 3; ++dj) {
#pragma unroll
                            for (int q = 0; q < WIDTH / 4; ++q) {
#pragma unroll
                                for (int k = 0; k < PX; ++k) {
                                    const float4 hv = *reinterpret_cast<const float4 *>(
                                        th + (size_t)(lidx[k] + roff + dj) * WIDTH + 4 * q);
                                    const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
#pragma unroll
                                        for (int j = 0; j < 4; ++j)
                                            o[k][j] = fmaf(hh[i], W3r[(dj * WIDTH + 4 * q + i) * 4 + j], o[k][j]);
                                }
                            }
                        }
                    }
                    }
                    // shift = o[0:2], raw log-scale = o[2:4]  (tf.split, layers.py:494)
                    if constexpr (MFMA) {
                        // matrix-core layouts (fp32 and fp16-CNN alike): the host pre-scaled the raw columns by 2*log2(e), so
                        //   t = exp2(raw') = exp(2 raw);  ls*log2(e) = scl*tanh(raw) = scl - 2 scl/(t + 1)
                        // and the log-det is accumulated in log2 units (ld2), converted once per patch.
                        const float scl = wl[coff + NF2_CPL_S + 1];     // (NF3_CPL_S == NF2_CPL_S)
                        const float m2scl = wl[coff + NF2_CPL_S + 2];
                        if (type == NF_OP_COUPLING_FWD) {
#pragma unroll
                            for (int k = 0; k < PX; ++k) {
                                const float l0 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[k][2]) + 1.0f), m2scl, scl);
                                const float l1 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[k][3]) + 1.0f), m2scl, scl);
                                z[k][2] = fmaf(z[k][2], __builtin_amdgcn_exp2f(l0), o[k][0]);
                                z[k][3] = fmaf(z[k][3], __builtin_amdgcn_exp2f(l1), o[k][1]);
                                if (own[k]) ld2 += l0 + l1;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < PX; ++k) {
                                const float l0 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[k][2]) + 1.0f), m2scl, scl);
                                const float l1 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[k][3]) + 1.0f), m2scl, scl);
                                z[k][2] = (z[k][2] - o[k][0]) * __builtin_amdgcn_exp2f(-l0);
                                z[k][3] = (z[k][3] - o[k][1]) * __builtin_amdgcn_exp2f(-l1);
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < PX; ++k) {
                            const float ls0 = sc * nf_tanh(o[k][2]);
                            const float ls1 = sc * nf_tanh(o[k][3]);
                            if (type == NF_OP_COUPLING_FWD) {
                                z[k][2] = fmaf(z[k][2], nf_exp(ls0), o[k][0]);
                                z[k][3] = fmaf(z[k][3], nf_exp(ls1), o[k][1]);
                                if (own[k]) ld += ls0 + ls1;
                            } else {
                                z[k][2] = (z[k][2] - o[k][0]) * nf_exp(-ls0);
                                z[k][3] = (z[k][3] - o[k][1]) * nf_exp(-ls1);
                            }
                        }
                    }
                }
#ifdef NF_TIMELINE
                asm volatile("" ::"v"(z[0][2]));
                if (n_cpl < 8) NF_STAMP(3 + n_cpl);
                ++n_cpl;
#endif
                }   // run of (mix, coupling) pairs
                if (patch_done) break;
            } else if (type == NF_OP_SDN_DIV || type == NF_OP_SDN_MUL) {
                const float4 *y4 = reinterpret_cast<const float4 *>(a.y + patch_off);
                float4 yv[PX];
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    yv[k] = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (act[k]) yv[k] = y4[gidx[k]];
                }
                sdn_apply(type, prog.ops[op].off, yv);
            } else if (type == NF_OP_SCALE || type == NF_OP_SCALE_COND) {
                const float s = type == NF_OP_SCALE ? P[0] : a.cond_a[prog.ops[op].off & 3];
#pragma unroll
                for (int k = 0; k < PX; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) z[k][c] *= s;
            } else if (type == NF_OP_STORE) {
                if (a.out) {
                    float4 *out4 = reinterpret_cast<float4 *>(a.out + patch_off);
#pragma unroll
                    for (int k = 0; k < PX; ++k)
                        if (own[k]) out4[gidx[k]] = make_float4(z[k][0], z[k][1], z[k][2], z[k][3]);
                }
                if constexpr (!MFMA || BS) {
                    if (a.flags & NF_K_CARRY_OUT) {   // the log-det of what ran so far goes with the stored tensor
                        float *cp = a.ld_carry + (size_t)b * THREADS + t;
                        float v = fmaf(ld2, 0.6931471805599453f, ld);
                        if (a.flags & NF_K_CARRY_ADD) v += *cp;
                        *cp = v;
                    }
                }
            }
        }

        // next patch's x: in flight during the epilogue
        if (!PHILOX && !tiled) {
            const int64_t nb = b + gridDim.x;
            const bool more = nb < a.B;
            const float4 *in4 = reinterpret_cast<const float4 *>(a.in) + (size_t)(more ? nb : b) * (size_t)HW;
#pragma unroll
            for (int k = 0; k < PX; ++k) {   // assigned on every path: zin is dead across the body of the loop
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (more && act[k]) v = in4[gidx[k]];
                zin[k] = v;
            }
            if constexpr (YPF) {
                const float4 *y4n = reinterpret_cast<const float4 *>(a.y) + (size_t)(more ? nb : b) * (size_t)HW;
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (more && sdn_first) v = y4n[gidx[k]];
                    yin[k] = v;
                }
            }
        }

        if constexpr (!MFMA || BS) {
            if (a.stats) continue;   // batch-statistics pass: no outputs
        }

        // ---- epilogue ----
        if (a.out) {
            float4 *out4 = reinterpret_cast<float4 *>(a.out + patch_off);
#pragma unroll
            for (int k = 0; k < PX; ++k)
                if (own[k]) out4[gidx[k]] = make_float4(z[k][0], z[k][1], z[k][2], z[k][3]);
        }
        if (a.nll_out || a.sd_out || a.ld_out || a.sums || (tiled && a.tile_part)) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < PX; ++k)
                if (own[k]) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        s1 += z[k][c];
                        s2 = fmaf(z[k][c], z[k][c], s2);
                    }
                }
            float r0 = wave_sum(fmaf(ld2, 0.6931471805599453f, ld)), r1 = wave_sum(s1), r2 = wave_sum(s2);
            constexpr int NW = THREADS / 64;
            if constexpr (NW > 1 && DEFER) {
                // Two scratch sets, alternating by patch.  The wavefronts leave their partials of THIS patch in one set and go on;
                // the other set holds the PREVIOUS patch's partials, complete since the barriers of this patch's body, and wavefront
                // 0 finishes that patch now — the epilogue itself has no barrier (the last patch is finished behind the loop).  A
                // body without couplings has no barrier: then one stands here, in front of the writes (the set being overwritten
                // was read in the previous epilogue).
                if (!synced) __syncthreads();
                float *const rd = red + (red_sel ? 3 * NW : 0);
                const float *const prev = red + (red_sel ? 0 : 3 * NW);
                red_sel ^= 1;
                const int wv = t >> 6;
                if ((t & 63) == 0) {
                    rd[wv] = r0;
                    rd[NW + wv] = r1;
                    rd[2 * NW + wv] = r2;
                }
                if (pend_b >= 0) finish_patch(prev, pend_b, 0.f, 0.f, 0.f);
                pend_b = b;
            } else if constexpr (NW > 1) {
                // two scratch sets, alternating by patch: wavefront 0 finishes this patch while the others are already in the next
                // one — the next barrier any of them reaches comes after its reads
                float *const rd = red + (red_sel ? 3 * NW : 0);
                red_sel ^= 1;
                const int wv = t >> 6;
                if ((t & 63) == 0) {
                    rd[wv] = r0;
                    rd[NW + wv] = r1;
                    rd[2 * NW + wv] = r2;
                }
                __syncthreads();
                finish_patch(rd, b, 0.f, 0.f, 0.f);
            } else {
                finish_patch(nullptr, b, r0, r1, r2);
            }
        }
        NF_STAMP(11);
    }
    if constexpr (DEFER) {
        if (pend_b >= 0) {   // the last patch
            __syncthreads();
            finish_patch(red + (red_sel ? 0 : 3 * (THREADS / 64)), pend_b, 0.f, 0.f, 0.f);
        }
    }

    if (a.sums && t == 0 && !tiled) {
        // device-scope atomics on ONE cache line serialise at ~10 ns each (2 per workgroup = 5.6 us of a
        // 55 us launch at B = 1024); the slotted layout spreads them over NF_SUMS_SLOTS lines
        double *sp = a.sums;
        if (a.flags & NF_K_SUMS_WIDE) sp += (size_t)(blockIdx.x & (NF_SUMS_SLOTS - 1)) * NF_SUMS_STRIDE;
        atomicAdd(&sp[0], acc_nll);
        atomicAdd(&sp[1], acc_sd);
        if (blockIdx.x == 0) atomicAdd(&sp[2], (double)a.B);
    }
#ifdef NF_TIMELINE
    stamp_on = true;
    NF_STAMP(12);
#endif
}

template <int WIDTH, int THREADS, int PX, bool PHILOX, bool MFMA, bool FULL, int PREC, bool BS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(NF_MIN_WAVES(THREADS, PX, MFMA)))) void nf_flow_kernel(const NfProgram prog, const NfLaunch a)
{
    nf_flow_body<WIDTH, THREADS, PX, PHILOX, MFMA, FULL, PREC, BS, false>(prog, a);
}

// NF_K_TILED launches over full 64x64 tiles: the full-patch matrix-core geometry of nf_flow_kernel<4, 1024, 4, ., true, true, PREC, false>
// (PREC = 0: fp32, 2x2-blocked lanes; PREC = 2: fp16 CNN on v_mfma_f32_16x16x32_f16)
template <bool PHILOX, int PREC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(NF_MIN_WAVES(1024, 4, true)))) void nf_flow_tile64_kernel(const NfProgram prog, const NfLaunch a)
{
    nf_flow_body<4, 1024, 4, PHILOX, true, true, PREC, false, true>(prog, a);
}

// Batch-statistics mode, device side of the re-fold (layers.py:388-391 + the BN-eval folding of fold_coupling):
// the moments of one normalisation from the slotted sums, then  W[r][j] *= 1/sqrt(var_j + eps),
// B[j] = (B[j] - mean_j)/sqrt(var_j + eps)  in place on the working copy of the folded block (which holds the
// identity-normalised layer), and the accumulators are cleared for the next pass.  One workgroup.
__global__ __launch_bounds__(64) void nf_bs_finalize_kernel(double *__restrict__ stats, int w, double n, float *__restrict__ Wm, int rows,
                                                            float *__restrict__ Bv, float *__restrict__ mean_out, float *__restrict__ var_out)
{
    __shared__ double sc[64];
    const int j = threadIdx.x;
    if (j < w) {
        double sum = 0.0, sq = 0.0;
        for (int slot = 0; slot < NF_STATS_SLOTS; ++slot) {
            sum += stats[(size_t)slot * 2 * w + j];
            sq += stats[(size_t)slot * 2 * w + w + j];
        }
        const double m = sum / n;
        double v = sq / n - m * m;   // tf.nn.moments: population variance
        if (v < 0.0) v = 0.0;
        const float mf = (float)m, vf = (float)v;
        mean_out[j] = mf;
        var_out[j] = vf;
        const double s = 1.0 / sqrt((double)vf + 1e-4);   // layers.py:378 (eps), as fold_coupling
        sc[j] = s;
        Bv[j] = (float)(((double)Bv[j] - (double)mf) * s);
    }
    __syncthreads();
    for (int i = j; i < rows * w; i += 64) Wm[i] = (float)((double)Wm[i] * sc[i % w]);
    for (int i = j; i < NF_STATS_SLOTS * 2 * w; i += 64) stats[i] = 0.0;
}

// dst[pairs[2i]] = src[pairs[2i+1]]: the normalisation-dependent entries of the matrix-core layout from the working scalar layout
__global__ __launch_bounds__(256) void nf_gather_kernel(float *__restrict__ dst, const float *__restrict__ src, const int32_t *__restrict__ pairs, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[pairs[2 * i]] = src[pairs[2 * i + 1]];
}

// fold the slotted sums into the plain (sum nll, sum sd, count) triple — one wavefront
__global__ __launch_bounds__(64) void nf_sums_reduce_kernel(const double *__restrict__ wide, double *__restrict__ out3,
                                                            int accumulate)
{
    static_assert(NF_SUMS_SLOTS == 64, "one lane per slot");
    const int s = threadIdx.x;
    double v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = wave_sum(wide[(size_t)s * NF_SUMS_STRIDE + k]);
    }
    if (s == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) out3[k] = (accumulate ? out3[k] : 0.0) + v[k];
}

// NF_K_TILED launches (nf_device.h): per image, the sums the tiles of every segment left in `part` -> nll / sd / log-det exactly as
// the fused kernel's epilogue forms them for a patch it holds whole (the log-det over all segments, the moments of z from the
// last one), and the call's (sum nll, sum sd, count) accumulators.  One wavefront per image: lane l adds up tiles l, l + 64, ...
// of every segment in double, then the fixed-order wavefront sum — the same bits whatever else is in the batch.
__global__ __launch_bounds__(64) void nf_tile_combine_kernel(const float *__restrict__ part, const NfTileParts tp, int64_t B, double n,
                                                             double ld_const, uint32_t flags, float *__restrict__ nll_out,
                                                             float *__restrict__ sd_out, float *__restrict__ ld_out, double *__restrict__ sums)
{
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    for (int sgm = 0; sgm < tp.n_seg; ++sgm) {
        const int nt = tp.nt[sgm];
        const float4 *p4 = reinterpret_cast<const float4 *>(part) + tp.off[sgm] + (size_t)i * nt;
        const bool last = sgm == tp.n_seg - 1;
        for (int k = lane; k < nt; k += 64) {
            const float4 v = p4[k];
            r0 += (double)v.x;
            if (last) {
                r1 += (double)v.y;
                r2 += (double)v.z;
            }
        }
    }
    r0 = wave_sum(r0);
    r1 = wave_sum(r1);
    r2 = wave_sum(r2);
    if (lane == 0) {
        const double logdet = r0 + ld_const;
        double nll = -logdet;
        if (flags & NF_K_PRIOR) nll += 0.5 * n * 1.8378770664093453 + 0.5 * r2;
        const double mean = r1 / n;
        double var = r2 / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double sd = (double)__builtin_amdgcn_sqrtf((float)var);
        if (nll_out) nll_out[i] = (float)nll;
        if (sd_out) sd_out[i] = (float)sd;
        if (ld_out) ld_out[i] = (float)logdet;
        if (sums) {
            double *sp = sums;
            if (flags & NF_K_SUMS_WIDE) sp += (size_t)(blockIdx.x & (NF_SUMS_SLOTS - 1)) * NF_SUMS_STRIDE;
            atomicAdd(&sp[0], (double)(float)nll);
            atomicAdd(&sp[1], (double)(float)sd);
            if (blockIdx.x == 0) atomicAdd(&sp[2], (double)B);
        }
    }
}

// --------------------------------------------------------------------------
// synthetic SIDD-like patches: y ~ U[0,1), x = eps*sqrt(b1*y+b2)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nf_synth_kernel(uint64_t seed, int64_t patch_base, int64_t B, int HW,
                                                       float beta1, float beta2, float *__restrict__ y_out,
                                                       float *__restrict__ x_out)
{
    const int64_t total = B * (int64_t)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const uint32_t p = (uint32_t)(i - b * HW);
        const uint4 r = philox_pixel(seed, patch_base + b, p, NF_STREAM_Y);
        const float4 y = make_float4(u01_24(r.x), u01_24(r.y), u01_24(r.z), u01_24(r.w));
        if (y_out) reinterpret_cast<float4 *>(y_out)[i] = y;
        if (x_out) {
            float e[4];
            philox_normal4(seed, patch_base + b, p, NF_STREAM_XEPS, e);
            reinterpret_cast<float4 *>(x_out)[i] =
                make_float4(e[0] * sqrtf(fmaf(beta1, y.x, beta2)), e[1] * sqrtf(fmaf(beta1, y.y, beta2)),
                            e[2] * sqrtf(fmaf(beta1, y.z, beta2)), e[3] * sqrtf(fmaf(beta1, y.w, beta2)));
        }
    }
}

// cross-rank batch statistics (nf_set_sync): slotted fp64 sums [NF_STATS_SLOTS][nvals] -> totals in `buf`, and back as
// slot 0 = the all-reduced total, every other slot 0 (the consumers add the slots up)
__global__ __launch_bounds__(64) void nf_stats_compact_kernel(const double *__restrict__ stats, int nvals, double *__restrict__ buf)
{
    const int j = threadIdx.x;
    if (j >= nvals) return;
    double s = 0.0;
    for (int k = 0; k < NF_STATS_SLOTS; ++k) s += stats[(size_t)k * nvals + j];
    buf[j] = s;
}
__global__ __launch_bounds__(64) void nf_stats_scatter_kernel(double *__restrict__ stats, int nvals, const double *__restrict__ buf)
{
    const int j = threadIdx.x;
    if (j >= nvals) return;
    stats[j] = buf[j];
    for (int k = 1; k < NF_STATS_SLOTS; ++k) stats[(size_t)k * nvals + j] = 0.0;
}

// the N(0,1) draw the flow kernels make in-kernel for sampling (stream NF_STREAM_SAMP), written out (nf_sample_eps)
__global__ __launch_bounds__(256) void nf_eps_kernel(uint64_t seed, int64_t patch_base, int64_t B, int HW, float *__restrict__ eps_out)
{
    const int64_t total = B * (int64_t)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        float e[4];
        philox_normal4(seed, patch_base + b, (uint32_t)(i - b * HW), NF_STREAM_SAMP, e);
        reinterpret_cast<float4 *>(eps_out)[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

inline int env_int(const char *name)
{
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}

template <int WIDTH, int THREADS, int PX, bool PHILOX, bool MFMA, bool FULL, int PREC, bool BS = false, bool TF = false>
hipError_t launch_flow_p(const NfProgram &prog, const NfLaunch &a, int n_cu, hipStream_t stream)
{
    const int tile_px = ((a.H + 2) * (a.W + 2) + 1) & ~1;
    size_t lds_f = (size_t)tile_px * (PREC != 0 ? 3 : 2 + WIDTH) + ((6 * (THREADS / 64) + 3) & ~3);
    if (PREC == 1 && a.H == 32) lds_f = (size_t)(34 * 48) * 3 + ((6 * (THREADS / 64) + 3) & ~3);   // padded row pitch
    if (PREC == 2) lds_f = (size_t)((a.H + 2) * nf11_pitch(a.H)) * 3 + ((6 * (THREADS / 64) + 3) & ~3);
    if (MFMA) lds_f += (size_t)((a.n_params + 3) & ~3) + 16;   // + the identity block
    if (BS) lds_f += (size_t)(THREADS / 64) * 8 + 16;
    const size_t lds = sizeof(float) * lds_f;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    void (*const kern)(const NfProgram, const NfLaunch) = [] {
        if constexpr (TF) return &nf_flow_tile64_kernel<PHILOX, PREC>;
        else return &nf_flow_kernel<WIDTH, THREADS, PX, PHILOX, MFMA, FULL, PREC, BS>;
    }();
    const void *fn = reinterpret_cast<const void *>(kern);
    // (device << 40 | lds bytes << 8 | resident workgroups per CU) of the last query; racy but idempotent.  The
    // >64 KiB opt-in is a per-device function attribute, so the device is part of the key.
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<uint64_t> cache{0};
    const uint64_t key = ((uint64_t)(dev & 0xff) << 40) | ((uint64_t)lds << 8);
    uint64_t c = cache.load(std::memory_order_relaxed);
    int occ;
    if ((c & ~(uint64_t)0xff) == key && (c & 0xff) != 0) {
        occ = (int)(c & 0xff);
    } else {
        if (lds > 64 * 1024) {   // opt in to exactly what this launch needs
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
    // persistent grid: exactly the resident capacity, each workgroup strides over patches
    int64_t groups = (int64_t)n_cu * occ;
    if (a.B < groups) groups = a.B;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(THREADS), lds, stream, prog, a);
    return hipGetLastError();
}

template <int WIDTH, int THREADS, int PX, bool PHILOX, bool MFMA, bool FULL>
hipError_t launch_flow_f(const NfProgram &prog, const NfLaunch &a, int n_cu, hipStream_t stream)
{
    if constexpr (MFMA && FULL && PX == 4) {
        if (a.flags & NF_K_FP16_BIG) return launch_flow_p<WIDTH, THREADS, PX, PHILOX, MFMA, FULL, 2>(prog, a, n_cu, stream);
        if (a.flags & NF_K_FP16_CNN) return launch_flow_p<WIDTH, THREADS, PX, PHILOX, MFMA, FULL, 1>(prog, a, n_cu, stream);
    }
    if (a.flags & NF_K_FP16_CNN) return hipErrorInvalidValue;
    if constexpr (MFMA) {   // batch-statistics launches of the width-4 model: the variant with the fused finaliser
        if (a.flags & NF_K_BATCHSTATS) return launch_flow_p<WIDTH, THREADS, PX, PHILOX, MFMA, FULL, 0, true>(prog, a, n_cu, stream);
    }
    return launch_flow_p<WIDTH, THREADS, PX, PHILOX, MFMA, FULL, 0>(prog, a, n_cu, stream);
}

template <int WIDTH, int THREADS, int PX, bool PHILOX, bool MFMA>
hipError_t launch_flow_v(const NfProgram &prog, const NfLaunch &a, int n_cu, hipStream_t stream)
{
    // full-patch specialisation only for the production shapes (32x32, 64x64) to bound code size
    if constexpr (THREADS * PX == 1024 || THREADS * PX == 4096) {
        if (a.H == a.W && a.H * a.W == THREADS * PX) {
            if (!(a.flags & NF_K_TILED)) return launch_flow_f<WIDTH, THREADS, PX, PHILOX, MFMA, true>(prog, a, n_cu, stream);
            // tiled launches: full 64x64 tiles have their own instantiation of the blocked geometry, everything else is masked
            if constexpr (MFMA && WIDTH == 4 && THREADS == 1024 && PX == 4) {
                if (a.flags & NF_K_FP16_BIG) return launch_flow_p<WIDTH, THREADS, PX, PHILOX, MFMA, true, 2, false, true>(prog, a, n_cu, stream);
                static const bool masked = env_int("NF_TILE_MASKED") != 0;   // A/B aid
                if (!masked && !(a.flags & NF_K_BATCHSTATS)) return launch_flow_p<WIDTH, THREADS, PX, PHILOX, MFMA, true, 0, false, true>(prog, a, n_cu, stream);
            }
        }
    }
    return launch_flow_f<WIDTH, THREADS, PX, PHILOX, MFMA, false>(prog, a, n_cu, stream);
}

template <int WIDTH, int THREADS, int PX, bool MFMA>
hipError_t launch_flow(const NfProgram &prog, const NfLaunch &a, int n_cu, hipStream_t stream)
{
    // the in-kernel Philox/Box-Muller prologue is a separate instantiation so that its
    // registers and libm code never burden the likelihood path
    if (a.flags & NF_K_PHILOX_IN) return launch_flow_v<WIDTH, THREADS, PX, true, MFMA>(prog, a, n_cu, stream);
    return launch_flow_v<WIDTH, THREADS, PX, false, MFMA>(prog, a, n_cu, stream);
}

#ifdef NF_ISA_PROBE
// tools/isa_budget.py: only the kernels under study are instantiated (seconds instead of minutes of compile time)
}  // namespace
const void *nf_isa_probe_kernels[] = {
    reinterpret_cast<const void *>(&nf_flow_kernel<4, 1024, 4, false, true, true, 2, false>),
    reinterpret_cast<const void *>(&nf_flow_kernel<4, 256, 4, false, true, true, 2, false>),
    reinterpret_cast<const void *>(&nf_flow_kernel<4, 256, 4, false, true, true, 0, false>),
};
#else
template <int WIDTH, bool MFMA>
hipError_t dispatch_geom(const NfProgram &prog, const NfLaunch &a, int n_cu, hipStream_t stream)
{
    const int hw = a.H * a.W;
    if (hw <= 64) return launch_flow<WIDTH, 64, 1, MFMA>(prog, a, n_cu, stream);
    if (hw <= 256) return launch_flow<WIDTH, 256, 1, MFMA>(prog, a, n_cu, stream);
    if constexpr (WIDTH >= 32) {
        // wide CNNs keep one pixel per lane (the per-pixel hidden vector alone is WIDTH registers)
        if (hw <= 1024) return launch_flow<WIDTH, 1024, 1, MFMA>(prog, a, n_cu, stream);
        return hipErrorInvalidValue;
    } else {
        if (hw <= 1024) {
            // workgroup geometry for the 32x32 patch; NF_GEOM=<threads> overrides (tuning aid)
            static const int geom = env_int("NF_GEOM");
            if (geom == 512) return launch_flow<WIDTH, 512, 2, MFMA>(prog, a, n_cu, stream);
            if (geom == 1024) return launch_flow<WIDTH, 1024, 1, MFMA>(prog, a, n_cu, stream);
            return launch_flow<WIDTH, 256, 4, MFMA>(prog, a, n_cu, stream);
        }
        if (hw <= 4096) return launch_flow<WIDTH, 1024, 4, MFMA>(prog, a, n_cu, stream);
        return hipErrorInvalidValue;
    }
}

}  // namespace

// ---- entry points used by nf_host.hip ----
hipError_t nf_launch_flow(const NfProgram &prog, const NfLaunch &a_in, int n_cu, hipStream_t stream, bool matrix_core)
{
    // what the kernels would otherwise find out with scalar loads, once per workgroup (= once per patch in a one-round launch)
    NfLaunch a = a_in;
    a.run_first = -1;
    a.run_n = a.run_moff = a.run_coff = a.run_stride = a.run_type = a.n_cpl = 0;
    auto is_cpl = [](int t) { return t == NF_OP_COUPLING_FWD || t == NF_OP_COUPLING_REV; };
    for (int q = 0; q < prog.n_ops; ++q) a.n_cpl += is_cpl(prog.ops[q].type) ? 1 : 0;
    const int ct = a.n_cpl < 1 ? 1 : a.n_cpl;
    a.fair_t1 = (ct + 3) / 4;
    a.fair_t2 = (2 * ct + 3) / 4;
    a.fair_t3 = (3 * ct + 3) / 4;
    for (int q = 0; q + 1 < prog.n_ops && a.run_first < 0; ++q) {
        const int t1 = prog.ops[q + 1].type;
        if (prog.ops[q].type != NF_OP_MIX || !is_cpl(t1)) continue;
        a.run_first = q;
        a.run_n = 1;
        a.run_moff = prog.ops[q].off;
        a.run_coff = prog.ops[q + 1].off;
        a.run_type = t1;
        for (int qq = q + 2; qq + 1 < prog.n_ops; qq += 2) {
            if (prog.ops[qq].type != NF_OP_MIX || prog.ops[qq + 1].type != t1) break;
            const int sm = prog.ops[qq].off - a.run_moff, sc = prog.ops[qq + 1].off - a.run_coff;
            if (a.run_n == 1) a.run_stride = sm;
            if (sm != a.run_n * a.run_stride || sc != a.run_n * a.run_stride) break;
            ++a.run_n;
        }
    }
    if (matrix_core) return prog.width == 4 ? dispatch_geom<4, true>(prog, a, n_cu, stream) : hipErrorInvalidValue;
    switch (prog.width) {
    case 4: return dispatch_geom<4, false>(prog, a, n_cu, stream);
    case 8: return dispatch_geom<8, false>(prog, a, n_cu, stream);
    case 16: return dispatch_geom<16, false>(prog, a, n_cu, stream);
    case 32: return dispatch_geom<32, false>(prog, a, n_cu, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t nf_launch_bs_finalize(double *stats, int w, double n, float *Wm, int rows, float *Bv, float *mean_out, float *var_out,
                                 hipStream_t stream)
{
    hipLaunchKernelGGL(nf_bs_finalize_kernel, dim3(1), dim3(64), 0, stream, stats, w, n, Wm, rows, Bv, mean_out, var_out);
    return hipGetLastError();
}

hipError_t nf_launch_gather(float *dst, const float *src, const int32_t *pairs, int n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(nf_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, pairs, n);
    return hipGetLastError();
}

hipError_t nf_launch_sums_reduce(const double *wide, double *out3, bool accumulate, hipStream_t stream)
{
    hipLaunchKernelGGL(nf_sums_reduce_kernel, dim3(1), dim3(64), 0, stream, wide, out3, accumulate ? 1 : 0);
    return hipGetLastError();
}

hipError_t nf_launch_tile_combine(const float *part, const NfTileParts &tp, int64_t B, double n, double ld_const, uint32_t flags,
                                  float *nll_out, float *sd_out, float *ld_out, double *sums, hipStream_t stream)
{
    if (B <= 0) return hipSuccess;
    if (B > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nf_tile_combine_kernel, dim3((unsigned)B), dim3(64), 0, stream, part, tp, B, n, ld_const, flags,
                       nll_out, sd_out, ld_out, sums);
    return hipGetLastError();
}

hipError_t nf_launch_stats_compact(const double *stats, int nvals, double *buf, hipStream_t stream)
{
    hipLaunchKernelGGL(nf_stats_compact_kernel, dim3(1), dim3(64), 0, stream, stats, nvals, buf);
    return hipGetLastError();
}
hipError_t nf_launch_stats_scatter(double *stats, int nvals, const double *buf, hipStream_t stream)
{
    hipLaunchKernelGGL(nf_stats_scatter_kernel, dim3(1), dim3(64), 0, stream, stats, nvals, buf);
    return hipGetLastError();
}

hipError_t nf_launch_eps(uint64_t seed, int64_t patch_base, int64_t B, int HW, float *eps_out, hipStream_t stream)
{
    if (B <= 0) return hipSuccess;
    int64_t blocks = (B * (int64_t)HW + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(nf_eps_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, patch_base, B, HW, eps_out);
    return hipGetLastError();
}

hipError_t nf_launch_synth(uint64_t seed, int64_t patch_base, int64_t B, int HW, float beta1, float beta2,
                           float *y_out, float *x_out, hipStream_t stream)
{
    if (B <= 0) return hipSuccess;
    int64_t blocks = (B * (int64_t)HW + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(nf_synth_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, patch_base, B, HW, beta1,
                       beta2, y_out, x_out);
    return hipGetLastError();
}
#endif   // NF_ISA_PROBE
