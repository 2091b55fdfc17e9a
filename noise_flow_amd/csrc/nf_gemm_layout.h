// fp16-CNN GEMM layout (NF_CFG_FP16_CNN at coupling widths 33 .. 512, nf_gemm16.hip) — kept out of nf_device.h on purpose:
// profiles/traffic.json is stamped with a content hash of the headers the HEADLINE kernel includes.
//
// Same structure as NF7_* (nf_device.h) on v_mfma_f32_32x32x16_f16 (K = 16 per instruction, fp32 accumulate): an A / B operand
// is 8 halves = 4 dwords per lane, the K slice of lane half g being elements 8g .. 8g+7.  Folded weights and the three CNN
// inputs (z0, relu(h1), relu(h2)) are rounded to half once; biases, border table, tanh / exp, log-det stay fp32 (the rounding
// points of NF3_* / NF5_* and of the oracle's cnn_dtype='fp16').  WP = the width zero-padded to 64 / 128 / 256 / 512,
// MT = WP / 32; a band = 65536 / WP pixels (h1 of the band in half precision = 128 KiB of LDS).  Offsets in dwords:
//   COUPLING  E [16][4] @0 (raw columns pre-scaled by 2 log2 e), S [4] @64, IMG8 @68:
//     A1H [MT][2][64][4]       l_1 of output tile m, 2 instructions: #0 element q of lane half g = (tap 4g + q/2, ch q&1);
//                              #1: g = 0, q < 2 = (tap 8, ch q)
//     B1  [MT][2][16]          fp32 bias by (m, g, v): b1[32 m + c(v, g)]
//     B2  [MT][2][16]
//     A2H [MT][WP/16][64][4]   l_2 of output tile m: K step ks consumes registers 8 (ks & 1) .. +7 of input tile ks / 2:
//                              element q of lane l: W2[in = 32 (ks/2) + c(8 (ks&1) + q, l>>5)][out = 32 m + (l&31)]
//     A3H [MT][2][64][4]       P = W3^T h2, taps 0 .. 7: row i = l&31 = 4 tap + j; instruction m2 of input tile mi, element q:
//                              W3[tap][in = 32 mi + c(8 m2 + q, l>>5)][j]        (NOT pre-scaled: the kernel scales in fp32)
//     A3CH[MT][4][8][2]        tap 8 on v_mfma_f32_4x4x4_16b_f16: instruction q4 of input tile mi, (g, j):
//                              W3[8][32 mi + c(4 q4 .. 4 q4 + 3, g)][j]
#pragma once
#define NF8_CPL_E 0
#define NF8_CPL_S 64
#define NF8_CPL_IMG 68
__host__ __device__ constexpr int nf8_img_A1H(int) { return 0; }
__host__ __device__ constexpr int nf8_img_B1(int wp) { return (wp / 32) * 512; }
__host__ __device__ constexpr int nf8_img_B2(int wp) { return (wp / 32) * 544; }
__host__ __device__ constexpr int nf8_img_A2H(int wp) { return (wp / 32) * 576; }
__host__ __device__ constexpr int nf8_img_A3H(int wp) { return (wp / 32) * 576 + wp * wp / 2; }
__host__ __device__ constexpr int nf8_img_A3CH(int wp) { return (wp / 32) * 576 + wp * wp / 2 + (wp / 32) * 512; }
__host__ __device__ constexpr int nf8_img_size(int wp) { return (wp / 32) * (576 + 512 + 64) + wp * wp / 2; }
__host__ __device__ constexpr int nf8_cpl_size(int wp) { return NF8_CPL_IMG + nf8_img_size(wp); }
#define NF8_BAND_HALVES 65536   // hidden activations of one band: WP channels x NB pixels, half precision (128 KiB of LDS)

// ---- variant B (pixel tiles per wavefront, weights through LDS): one contiguous SLAB per channel tile m ----------------------------
//   IMG9 @68:  A1H [MT][2][64][4] and B1 [MT][2][16] as NF8 (same offsets), then
//     SLAB [MT] x { A2H(m) [WP/16][64][4]   the l_2 rows of OUTPUT tile m, K step ks at + 64 ks (16 bytes per lane)
//                   A3H(m) [2][64][4]        the l_last columns (taps 0 .. 7) of INPUT tile m
//                   A3CH(m)[4][8][2]         tap 8
//                   B2(m)  [2][16] }         fp32 bias of l_2's output tile m
__host__ __device__ constexpr int nf9_slab_dwords(int wp) { return (wp / 16) * 256 + 512 + 64 + 32; }
__host__ __device__ constexpr int nf9_slab_A3H(int wp) { return (wp / 16) * 256; }
__host__ __device__ constexpr int nf9_slab_A3CH(int wp) { return (wp / 16) * 256 + 512; }
__host__ __device__ constexpr int nf9_slab_B2(int wp) { return (wp / 16) * 256 + 512 + 64; }
__host__ __device__ constexpr int nf9_img_SLAB(int wp) { return (wp / 32) * 544; }
__host__ __device__ constexpr int nf9_img_size(int wp) { return (wp / 32) * 544 + (wp / 32) * nf9_slab_dwords(wp); }
__host__ __device__ constexpr int nf9_cpl_size(int wp) { return NF8_CPL_IMG + nf9_img_size(wp); }

// ---- exact-fp32 variant B (nf_gemm.hip, widths <= 128): the NF7_* values, one contiguous SLAB per channel tile m ---------------------
//   IMG10 @68:  A1 [MT][3][64][4], B1 [MT][2][16] as NF7 (same offsets; NF7's B2 block stays where it is, unused), then at
//   nf10_img_SLAB:  SLAB [MT] x { A2(m) [WP/8][64][4]   l_2 rows of OUTPUT tile m, chunk kc (4 K steps) at + 256 kc floats
//                                 A3(m) [4][64][4]       l_last columns (taps 0 .. 7) of INPUT tile m
//                                 A3C(m)[4][8][4]        tap 8 (v_mfma_f32_4x4x1)
//                                 B2(m) [2][16] }
__host__ __device__ constexpr int nf10_slab_floats(int wp) { return 32 * wp + 1024 + 128 + 32; }
__host__ __device__ constexpr int nf10_slab_A3(int wp) { return 32 * wp; }
__host__ __device__ constexpr int nf10_slab_A3C(int wp) { return 32 * wp + 1024; }
__host__ __device__ constexpr int nf10_slab_B2(int wp) { return 32 * wp + 1024 + 128; }
__host__ __device__ constexpr int nf10_img_SLAB(int wp) { return (wp / 32) * 832; }
__host__ __device__ constexpr int nf10_img_size(int wp) { return (wp / 32) * 832 + (wp / 32) * nf10_slab_floats(wp); }
__host__ __device__ constexpr int nf10_cpl_size(int wp) { return 68 + nf10_img_size(wp); }
