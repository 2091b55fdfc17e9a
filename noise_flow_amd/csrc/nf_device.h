// Shared host/device definitions of the folded Noise Flow program.
//
// A model is compiled by the host (nf_host.cpp) into a short straight-line
// "program" of ops over the 4 channel values of every pixel of one patch; the
// fused kernel (nf_kernels.hip) interprets it with one workgroup per patch and
// all intermediate tensors in registers / LDS.
#pragma once
#include <stdint.h>

#define NF_MAX_OPS 64

// device op codes
enum : int32_t {
    NF_OP_MIX          = 1,  // z <- z @ M               (16 floats, row-major [c][k])
    NF_OP_COUPLING_FWD = 2,  // NLL dir:  z1 <- z1*exp(ls) + shift ; ld += sum ls
    NF_OP_COUPLING_REV = 3,  // sampling: z1 <- (z1 - shift)*exp(-ls)
    NF_OP_SDN_DIV      = 4,  // NLL dir:  z <- z / sqrt(k1*y + b2) ; ld -= sum log scale
    NF_OP_SDN_MUL      = 5,  // sampling: z <- z * sqrt(k1*y + b2)
    NF_OP_SCALE        = 6,  // z <- z * s               (1 float; un-folded gain layer)
    NF_OP_SCALE_COND   = 7,  // z <- z * cond_a[slot]    (per-call scalar: plain `gain` layer)
    NF_OP_STORE        = 8,  // out <- z                 (checkpoint between the segments of a batch-statistics call)
};

struct NfOp {
    int32_t type;
    int32_t off;    // offset in floats into the folded parameter block (multiple of 4);
                    // for SDN_* / SCALE_COND ops: the conditioning slot (0..3)
};

struct NfProgram {
    int32_t n_ops;
    int32_t width;  // coupling CNN width of this program (all couplings share it)
    NfOp ops[NF_MAX_OPS];
};

// Folded coupling block layout (floats), w = CNN width; every section starts at a
// multiple of 4 floats so border-table rows can be fetched as one 16-byte load.
//   E   [16][4]        border table: (b3 + edge-channel taps outside the image) * exp(3 logs),
//                      indexed by mask = top | bottom<<1 | left<<2 | right<<3
//   W3  [9][w][4]      l_last weights * exp(3 logs)           (tap-major)
//   W1  [9][2][w]      l_1 weights   / sqrt(var1 + eps)
//   B1  [w]            (b1 - mean1)  / sqrt(var1 + eps)
//   W2  [w][w]         l_2 weights   / sqrt(var2 + eps)
//   B2  [w]            (b2 - mean2)  / sqrt(var2 + eps)
//   S   [4]            rescaling_scale, 0, 0, 0
__host__ __device__ constexpr int nf_cpl_off_E(int)    { return 0; }
__host__ __device__ constexpr int nf_cpl_off_W3(int)   { return 64; }
__host__ __device__ constexpr int nf_cpl_off_W1(int w) { return 64 + 36 * w; }
__host__ __device__ constexpr int nf_cpl_off_B1(int w) { return 64 + 36 * w + 18 * w; }
__host__ __device__ constexpr int nf_cpl_off_W2(int w) { return 64 + 36 * w + 18 * w + w; }
__host__ __device__ constexpr int nf_cpl_off_B2(int w) { return 64 + 36 * w + 18 * w + w + w * w; }
__host__ __device__ constexpr int nf_cpl_off_S(int w)  { return 64 + 36 * w + 18 * w + 2 * w + w * w; }
__host__ __device__ constexpr int nf_cpl_size(int w)   { return 64 + 36 * w + 18 * w + 2 * w + w * w + 4; }

// ---- "matrix-core" layout (width 4 only), used by nf_flow_mfma_kernel ------------------
// Every conv of the width-4 stack has exactly 4 output channels, i.e. it is
//   out[pixel][i] = sum_k W[k][i] * in[pixel][k]          (k = tap x input channel)
// which v_mfma_f32_4x4x1_16b_f32 evaluates for 64 pixels at once with the weights as
// the A operand (lane l supplies W[k][l & 3]) and the lane's own pixel as the B operand.
// Weights are therefore stored j-major ([out channel j][k]) so that a lane reads the A
// operands of consecutive k with 16-byte LDS loads.
//   MIX       Mt [4][4]            Mt[j][c] = M[c][j]
//   COUPLING  E  [16][4]  @0       border table (as above)
//             B1 [4]      @64
//             B2 [4]      @68
//             S  [4]      @72      rescaling_scale, 0, 0, 0
//             W1t[4][3][8] @76     W1t[j][di][dj*2+c]  (6 of 8 used per filter row)
//             W2t[4][4]   @172     W2t[j][i]
//             W3t[4][36]  @188     W3t[j][tap*4+i]
#define NF2_MIX_SIZE 16
#define NF2_CPL_E 0
#define NF2_CPL_B1 64
#define NF2_CPL_B2 68
#define NF2_CPL_S 72
#define NF2_CPL_W1T 76
#define NF2_CPL_W2T 172
#define NF2_CPL_W3T 188
#define NF2_CPL_SIZE 332
#define NF2_MAX_FLOATS 6144   // 24 KiB of LDS for the whole model's weights

// ---- fp16-CNN layout (matrix-core path, BASELINE configs[4]) ---------------------------
// Same ops; the three convs of the coupling CNN run on v_mfma_f32_4x4x4_16b_f16 (K = 4 per
// instruction, fp32 accumulate) with fp16 weights and fp16 activations; everything else
// (1x1 mixes, biases, border table, tanh/exp, log-det, prior) stays fp32.  Offsets in
// 32-bit words; one A operand = 4 halves = 2 words.
//   MIX       Mt [4][4] fp32 (as above)
//   COUPLING  E  [16][4] fp32 @0, B1[4] @64, B2[4] @68, S[4] @72 (sc, sc*log2e, -2 sc*log2e, 0)
//             W1h[4][3][4][2w] @76    per out channel j, filter row di, group g: 4 halves
//                                     g0 = dx0:(dj0c0,dj0c1,dj1c0,dj1c1)  g1 = dx0:(dj2c0,dj2c1,0,0)
//                                     g2 = dx1:(0,0,dj0c0,dj0c1)          g3 = dx1:(dj1c0,dj1c1,dj2c0,dj2c1)
//                                     (the B operand is always the 8-byte pair of horizontally
//                                      adjacent z0 pixels the lane already holds)
//             W2h[4][2w]    @172      per j: (i0,i1,i2,i3)
//             W3h[4][9][2w] @180      per j, tap: (i0,i1,i2,i3)
#define NF3_CPL_E 0
#define NF3_CPL_B1 64
#define NF3_CPL_B2 68
#define NF3_CPL_S 72
#define NF3_CPL_W1H 76
#define NF3_CPL_W2H 172
#define NF3_CPL_W3H 180
#define NF3_W3H_STRIDE 20      // words per lane-of-four: 9 taps x 2 words, padded so that every lane's run is 16-byte aligned
#define NF3_CPL_SIZE 260

// ---- fp16-CNN layout on the 1-issue matrix instruction (width 4, full 32x32 / 64x64 patches) ---------------------------
// v_mfma_f32_4x4x4_16b_f16 is a 2-pass instruction that owns the SIMD's issue port, so the 64 of them a lane needs per
// coupling and 4 pixels add to its VALU work instead of hiding under it.  v_mfma_f32_16x16x32_f16 issues once for 4 passes
// (16 cycles) and fits the width-4 CNN when the 2x2 OUTPUT BLOCK of a lane column sits on M and its 4x4 INPUT WINDOW on K:
//   unit   = 2 image rows x 32 columns = 64 pixels, one per lane: lane l = 16 g + n owns pixel (row 2u + (g >> 1), column
//            32 q + 2 n + (g & 1)); g = 2 a + p is the M slot of the output pixel (a, p) of the 2x2 block of lane column n
//   D      : lane (g, n), register v = output channel v of that pixel            (M row 4 g + v, N column n)
//   l_1    : K = 32 = 4 window rows x 4 window columns x 2 channels: ONE instruction per unit.  K slot gk = lane >> 4 of
//            the B operand reads window row nf11_l1_row(gk), columns 2n .. 2n+3 of the half2 tile (two 8-byte reads);
//            element e = 2 wc + c
//   l_last : K = 64 = 4 x 4 x 4 channels: TWO chained instructions m3 = 0, 1.  K slot gk reads window row
//            nf11_l3_row(gk, m3), column pair gk >> 1 (columns 2n + 2 (gk >> 1) + {0, 1}) of the 4 x half tile — one aligned
//            16-byte read; element e = 4 px + c
//   A      : lane (gk = l >> 4, m = l & 15) holds the 8 halves A[m][8 gk .. 8 gk + 7], m = 4 (2a + p) + j:
//            W[di = window row - a][dj = window column - p][c][j], zero where a tap index falls outside 0 .. 2
//   (56 % of the multiplies are useful; the window-row order of the K slots is chosen so that the lane groups that share an
//    LDS access cycle sit 128 B (8-byte reads) / 0 B (16-byte reads) apart modulo 256 B at the tile pitches below)
// Offsets in 32-bit words:
//   COUPLING  E [16][4] fp32 @0, B1 [4] @64, B2 [4] @68, S [4] @72 (as NF3_*), W2h [4][2w] @76 (as NF3_CPL_W2H),
//             A1 [64][4w] @84, A3 [2][64][4w] @340
#define NF11_CPL_E 0
#define NF11_CPL_B1 64
#define NF11_CPL_B2 68
#define NF11_CPL_S 72
#define NF11_CPL_W2H 76
#define NF11_CPL_A1 84
#define NF11_CPL_A3 340
#define NF11_CPL_SIZE 852
// 64x64 patches (one 16-wavefront workgroup per CU: LDS to spare) carry l_2 as v_mfma_f32_16x16x32_f16 operands too, so that the
// 2-pass v_mfma_f32_4x4x4_16b_f16 (owns the issue port for 2.85 plain-VALU slots) leaves the coupling: the B operand of one
// instruction is the PAIR {relu(h1) of unit 2u, relu(h1) of unit 2u + 1} (8 halves of the lane's own two pixels = K slot
// gk = lane >> 4), A2[half][lane (gk, m = 4 g + j)][e] = W2[e - 4 half][j] for gk == g and 4 half <= e < 4 half + 4, else 0:
// instruction `half` evaluates unit 2u + half.  12.5 % of the multiplies useful, one issue slot each.
//             A2 [2][64][4w] @852
#define NF11_CPL_A2 852
#define NF11_CPL_SIZE_L2 1364
#define NF11_MAX_FLOATS 12288   // 48 KiB of LDS for the whole model's weights in this layout
__host__ __device__ constexpr int nf11_l1_row(int gk) { return 2 * (gk & 1) + (gk >> 1); }
__host__ __device__ constexpr int nf11_l3_row(int gk, int m3) { return 2 * (gk & 1) + m3; }
#ifndef NF11_PITCH64
#define NF11_PITCH64 80
#endif
__host__ __device__ constexpr int nf11_pitch(int side) { return side == 64 ? NF11_PITCH64 : 48; }   // tile row pitch in pixels

// ---- wide-CNN layout (coupling width 32, nf_wide.hip) ------------------------------------------
// The three convs of a width-32 coupling CNN run on v_mfma_f32_32x32x2_f32 with the PIXELS on the N
// axis (a tile = 32 consecutive pixels of one image row, lanes n = lane & 31; the two lane halves
// g = lane >> 5 are the instruction's two K slices) and channels / taps on the M axis.  The D
// register v of lane half g then holds row  i = 8 (v >> 2) + 4 g + (v & 3)  — call it c(v, g) — so
// the output of one layer is, register by register, the B operand of the next: K step s of the next
// layer consumes D[s], its two K slices being the channels c(s, 0) and c(s, 1).  All weights below
// are stored in exactly the order the lanes fetch them (one A operand per lane per step):
//   MIX       M [4][4]          row-major [c][k] (scalar loads)
//   COUPLING  E  [16][4]  @0    border table, raw columns pre-scaled by 2 log2(e)  (as NF2_CPL_E)
//             S  [4]      @64   rescaling_scale, sc*log2e, -2 sc*log2e, 0
//             IMG         @68   the LDS image the workgroup copies per coupling:
//               A1 [3][64][4]   l_1: step = tap (9 used of 12), lane l: W1[tap][ch = l>>5][i = l&31]
//               B1 [2][16]      l_1 bias by (g, v): b1[c(v, g)]
//               A2 [4][64][4]   l_2: step s, lane l: W2[in = c(s, l>>5)][out = l&31]
//               B2 [2][16]
//               A3 [4][64][4]   l_last as P = W3^T h2: step s, lane l: W3[tap(i)][in = c(s, l>>5)][j = i&3], i = l&31,
//                               where row i of P is (a = i>>3, g' = (i>>2)&1, j): taps (di,dj) by (a, g'):
//                               a=0: (0,0)|(2,0)   a=1: (0,2)|(2,2)   a=2: (0,1)|(2,1)   a=3: (1,0)|(1,2)
//               A3C[4][8][4]    centre tap (1,1) on v_mfma_f32_4x4x1: step s, (g, j): W3[centre][c(s, g)][j]
#define NF4_CPL_E 0
#define NF4_CPL_S 64
#define NF4_CPL_IMG 68
#define NF4_IMG_A1 0
#define NF4_IMG_B1 768
#define NF4_IMG_A2 800
#define NF4_IMG_B2 1824
#define NF4_IMG_A3 1856
#define NF4_IMG_A3C 2880
#define NF4_IMG_SIZE 3008
#define NF4_CPL_SIZE (NF4_CPL_IMG + NF4_IMG_SIZE)
__host__ __device__ constexpr int nf4_chan(int v, int g) { return 8 * (v >> 2) + 4 * g + (v & 3); }

// ---- width-16 layout (nf_wide16.hip, v_mfma_f32_16x16x4_f32) ----------------------------------------
// Tile = 16 pixels (lane n = lane & 15), four K slices g = lane >> 4; D register v of group g = channel 4 g + v.
//   COUPLING  E [16][4] @0, S [4] @64 (as NF4), IMG6 @68:
//     A1  [64][4] + [64]   l_1: K step s, lane l: W1[(tap, ch)][i = l&15] with 2 tap + ch = 4 s + (l>>4) (< 18, else 0)
//     B1  [4][4]           bias by (g, v)
//     A2  [64][4]          l_2: step s, lane l: W2[in = 4 (l>>4) + s][out = l&15]
//     B2  [4][4]
//     A3A [64][4]          P chain A: row i = l&15 = (g' = i>>2, j = i&3), taps (0,0) (1,0) (2,0) (0,1) by g'; in = 4 (l>>4) + s
//     A3B [64][4]          P chain B: taps (0,2) (1,2) (2,2) (2,1)
//     A3C [16][4]          centre tap on v_mfma_f32_4x4x1: (g, j): W3[centre][4 g + s][j], s = 0..3
#define NF6_IMG_A1 0
#define NF6_IMG_B1 320
#define NF6_IMG_A2 336
#define NF6_IMG_B2 592
#define NF6_IMG_A3A 608
#define NF6_IMG_A3B 864
#define NF6_IMG_A3C 1120
#define NF6_IMG_SIZE 1184
#define NF6_CPL_SIZE (NF4_CPL_IMG + NF6_IMG_SIZE)

// ---- wide-CNN fp16 layout (NF_CFG_FP16_CNN at coupling width 32) --------------------------------
// Same kernel structure on v_mfma_f32_32x32x16_f16 (K = 16 per instruction, fp32 accumulate): an A / B operand is 8
// halves = 4 dwords per lane, the K slice of lane half g being elements 8g .. 8g+7.  Folded weights and the three CNN
// inputs (z0, relu(h1), relu(h2)) are rounded to half; biases, border table, tanh/exp, log-det stay fp32 (as NF3_*).
//   COUPLING  E [16][4] @0 (raw columns pre-scaled by 2 log2 e), S [4] @64, IMG16 @68 (dwords):
//     A1H [2][64][4]  l_1, 2 instructions: #0 element q of lane half g = (tap 4g + q/2, ch q&1); #1: g=0, q<2 = (tap 8, ch q)
//     B1  [2][16]     fp32, as NF4
//     A2H [2][64][4]  l_2: instruction m, element q = input channel c(8m + q, g)
//     B2  [2][16]
//     A3H [2][64][4]  P rows as NF4_IMG_A3 (NOT pre-scaled: the kernel scales the raw columns in fp32)
//     A3CH[4][8][2]   centre tap on v_mfma_f32_4x4x4_16b_f16: instruction q, (g, j): channels c(4q .. 4q+3, g)
#define NF5_IMG_A1H 0
#define NF5_IMG_B1 512
#define NF5_IMG_A2H 544
#define NF5_IMG_B2 1056
#define NF5_IMG_A3H 1088
#define NF5_IMG_A3CH 1600
#define NF5_IMG_SIZE 1664
#define NF5_CPL_SIZE (NF4_CPL_IMG + NF5_IMG_SIZE)

// ---- GEMM layout (coupling widths 33 .. 512, nf_gemm.hip) -------------------------------------------
// WP = the width zero-padded to 64 / 128 / 256 / 512, MT = WP / 32 channel tiles.  Same tile convention as NF4_* (a tile =
// 32 pixels on the N axis of v_mfma_f32_32x32x2_f32, channels on M, D register v of lane half g = channel c(v, g) of its
// tile); the hidden activations of a band of NB = 32768 / WP pixels live in LDS, the weights are streamed from L2 by the
// wavefront that consumes them, in exactly its fetch order:
//   COUPLING  E [16][4] @0, S [4] @64 (as NF4), IMG7 @68:
//     A1 [MT][3][64][4]      l_1 of output tile m: step = tap (9 used of 12), lane l: W1[tap][ch = l>>5][32 m + (l&31)]
//     B1 [MT][2][16]         l_1 bias by (m, g, v): b1[32 m + c(v, g)]
//     B2 [MT][2][16]
//     A2 [MT][WP/8][64][4]   l_2 of output tile m: K step kk = 4 kc + s consumes input tile kk / 16, register kk % 16:
//                            lane l: W2[in = 32 (kk/16) + c(kk%16, l>>5)][out = 32 m + (l&31)]
//     A3 [MT][4][64][4]      l_last as P = W3^T h2, taps 0 .. 7: row i = l&31 = 4 tap + j; step v of input tile mi:
//                            W3[tap][in = 32 mi + c(v, l>>5)][j]   (raw columns j >= 2 pre-scaled by 2 log2 e)
//     A3C[MT][4][8][4]       tap 8 on v_mfma_f32_4x4x1: step v = 4 grp + s of input tile mi, (g, j): W3[8][32 mi + c(v, g)][j]
#define NF7_CPL_E 0
#define NF7_CPL_S 64
#define NF7_CPL_IMG 68
__host__ __device__ constexpr int nf7_pad_width(int w) { return w <= 64 ? 64 : w <= 128 ? 128 : w <= 256 ? 256 : 512; }
__host__ __device__ constexpr int nf7_img_A1(int) { return 0; }
__host__ __device__ constexpr int nf7_img_B1(int wp) { return (wp / 32) * 768; }
__host__ __device__ constexpr int nf7_img_B2(int wp) { return (wp / 32) * 800; }
__host__ __device__ constexpr int nf7_img_A2(int wp) { return (wp / 32) * 832; }
__host__ __device__ constexpr int nf7_img_A3(int wp) { return (wp / 32) * 832 + wp * wp; }
__host__ __device__ constexpr int nf7_img_A3C(int wp) { return (wp / 32) * 832 + wp * wp + (wp / 32) * 1024; }
__host__ __device__ constexpr int nf7_img_size(int wp) { return (wp / 32) * 832 + wp * wp + (wp / 32) * (1024 + 128); }
#define NF7_P_STRIDE 44          // floats per pixel of a partial P tile in LDS: 8 taps x 4, tap 8 of lane half 0 / 1, pad (conflict-free)
__host__ __device__ constexpr int nf7_cpl_size(int wp) { return NF7_CPL_IMG + nf7_img_size(wp); }
#define NF7_BAND_FLOATS 32768   // hidden activations of one band: WP channels x NB pixels (128 KiB of LDS)
#define NF7_MAX_PIXELS 4096     // pixels per patch the GEMM kernels hold (8 per thread): up to 64x64

// launch flags
enum : uint32_t {
    NF_K_PRIOR     = 1u,   // nll = -(logdet + logp(z)); otherwise nll = -logdet
    NF_K_PHILOX_IN = 2u,   // input = Philox normal draw (sampling with in-kernel eps)
    NF_K_FP16_CNN  = 4u,   // parameter block is the fp16-CNN layout (NF3_*)
    NF_K_SUMS_WIDE = 8u,   // `sums` is the slotted layout of NF_SUMS_WIDE (include/noiseflow_hip.h)
    NF_K_BATCHSTATS = 16u, // matrix-core launch of a batch-statistics call: honours fix_* and stats
    // batch-statistics calls: the log-det of the segments already evaluated in their final form travels with the resident
    // tensor, one float per thread of the patch's workgroup (ld_carry), so that the last launch only runs the last segment
    NF_K_CARRY_OUT = 32u,  // at NF_OP_STORE: ld_carry <- this thread's log-det so far (+ ld_carry with NF_K_CARRY_ADD)
    NF_K_CARRY_ADD = 64u,
    NF_K_CARRY_IN  = 128u, // start from ld_carry instead of 0
    // images larger than one workgroup's tile (H or W > 64): every "patch" of the launch is one H x W TILE of an
    // img_H x img_W image, see NfLaunch::tile_* below
    NF_K_TILED     = 256u,
    NF_K_FP16_BIG  = 512u, // with NF_K_FP16_CNN: the parameter block is the NF11_* layout (v_mfma_f32_16x16x32_f16)
};

// ---- images beyond 64 x 64: overlapping tiles -------------------------------------------------------
// A coupling reads a 5 x 5 neighbourhood (3x3, 1x1, 3x3 convs), so a tile evaluated on its own (zero padding at ITS
// border) is exact `halo` = 2 x (number of couplings) pixels away from every tile border that is not an image border.
// Along one axis of extent S, tiles of t pixels start at origin(i) = min(i * (t - 2 halo), S - t), i = 0 .. n-1, and
// tile i REPORTS (stores its output, adds its log-det / prior terms for) the pixels [core0(i), core1(i)):
// core0(0) = 0, core0(i) = origin(i) + halo, core1(i) = core0(i + 1), core1(n - 1) = S — a partition of [0, S).
// Deep stacks would leave a small core (8 couplings: 32 of 64 pixels per axis, i.e. 4 x the work on a large image), so the host
// may cut the program after a coupling into up to NF_MAX_TILE_SEGS SEGMENTS, each its own tiled launch with its own, smaller
// halo, the tensor between two segments resident in HBM (32 B per pixel and segment: nothing next to the recomputation saved).
#define NF_MAX_TILE_SEGS 16
struct NfTileParts {            // where the per-tile sums of a call's segments sit (nf_tile_combine_kernel)
    int32_t n_seg;
    int32_t nt[NF_MAX_TILE_SEGS];    // tiles per image in segment s
    int64_t off[NF_MAX_TILE_SEGS];   // first float4 of segment s in the array ([image][tile] inside a segment)
};
__host__ __device__ constexpr int nf_tile_count(int S, int t, int halo)
{
    return S <= t ? 1 : (S - t + (t - 2 * halo) - 1) / (t - 2 * halo) + 1;
}
__host__ __device__ constexpr int nf_tile_origin(int i, int S, int t, int halo)
{
    return i * (t - 2 * halo) < S - t ? i * (t - 2 * halo) : (S - t > 0 ? S - t : 0);
}
__host__ __device__ constexpr int nf_tile_core0(int i, int S, int t, int halo)
{
    return i == 0 ? 0 : nf_tile_origin(i, S, t, halo) + halo;
}
__host__ __device__ constexpr int nf_tile_core1(int i, int n, int S, int t, int halo)
{
    return i == n - 1 ? S : nf_tile_origin(i + 1, S, t, halo) + halo;
}

struct NfLaunch {
    const float *params;   // folded parameter block (device)
    const float *in;       // [B,H,W,4] input tensor (x or eps); unused with NF_K_PHILOX_IN
    const float *y;        // [B,H,W,4] clean image (SDN ops) or null
    float *out;            // [B,H,W,4] output tensor (z or x) or null
    float *nll_out;        // [B] or null
    float *sd_out;         // [B] or null
    float *ld_out;         // [B] or null
    double *sums;          // double[3] accumulators or null
    int64_t B;
    int64_t patch_base;    // global index of patch 0 (Philox key)
    uint64_t seed;
    double ld_const;       // constant part of the log-det sum
    float in_scale;        // input multiplier (sampling temperature)
    float cond_a[4];       // per-call scalars of the conditional ops, indexed by NfOp::off (slot):
    float cond_b[4];       //   SDN_*: scale = sqrt(cond_a*y + cond_b);  SCALE_COND: z *= cond_a
    int32_t H, W;
    uint32_t flags;
    int32_t n_params;      // floats in the parameter block (matrix-core kernel stages it in LDS)
    // batch-statistics pass (scalar-weight kernel only): when `stats` is set the kernel stops at
    // coupling op `stats_op` and adds the per-channel sum / sum of squares of that layer's first
    // (stage 1) or second (stage 2) pre-normalisation activation to stats[slot][2*width];
    // an NF_OP_STORE before it writes the tensor at that point to `out`
    double *stats;
    int32_t stats_op;
    int32_t stats_stage;
    // matrix-core batch-statistics launches (NF_K_BATCHSTATS, width 4): the re-fold the PREVIOUS statistics pass makes
    // due — every workgroup applies it to its LDS weight image, workgroup 0 persists it for the next launch
    const double *fix_stats;   // [NF_STATS_SLOTS][8] sums of that pass, or null
    double fix_n;              // number of values behind each sum (B*H*W)
    int32_t fix_off;           // offset of the coupling's block in the matrix-core layout
    int32_t fix_stage;         // 1: l_1 / BN_1, 2: l_2 / BN_2
    float *fix_params_out;     // parameter block the next launch reads (n_params floats)
    float *fix_mom_out;        // mean[4], var[4] of that normalisation
    float *ld_carry;           // [B][threads per workgroup] (NF_K_CARRY_*), or null
    // NF_K_TILED (fused kernel, masked instantiations): in / y / out are [B / (tile_ny tile_nx)][img_H][img_W][4]; patch b
    // of the launch is tile (b % (tile_ny tile_nx)) of image b / (tile_ny tile_nx), H x W pixels at
    // (nf_tile_origin(ty, img_H, H, tile_halo), nf_tile_origin(tx, img_W, W, tile_halo)).  Border masks follow the IMAGE
    // border; outputs, log-det and prior sums cover the tile's core window only and go, per tile, to tile_part[b][4] =
    // (data log-det of THIS launch's ops, sum z, sum z^2, -) instead of nll_out / sd_out / ld_out / sums (nf_tile_combine_kernel
    // adds them up: the log-det over every segment, the moments of z from the last one)
    int32_t img_H, img_W;
    int32_t tile_ny, tile_nx;
    int32_t tile_halo;
    float *tile_part;
    // matrix-core kernels (filled by nf_launch_flow from the program, not by callers): the first run of `mix, coupling` pairs whose
    // parameter blocks sit a constant stride apart — walked as a counted loop without scalar loads — and the number of couplings
    int32_t run_first, run_n, run_moff, run_coff, run_stride, run_type, n_cpl;
    int32_t fair_t1, fair_t2, fair_t3;   // progress-based wave priority (NF_FAIR): the smallest coupling count c with 4 c / n_cpl >= 1, 2, 3
};

#define NF_STATS_SLOTS 64   // power of two

// Philox stream ids (4th counter word)
#define NF_STREAM_Y    0u   // synthetic clean image
#define NF_STREAM_XEPS 1u   // synthetic noise draw
#define NF_STREAM_SAMP 2u   // sampling-direction base draw
