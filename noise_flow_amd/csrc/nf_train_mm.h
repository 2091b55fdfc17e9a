// Dense products of the training step at coupling widths without stage kernels of their own (1 .. 512 except 4 / 8 / 16 / 32):
// hand-written fp32 GEMMs on the matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation), with
// what a library GEMM forces into separate passes over the [pixel][w] tensors fused into their prologues and epilogues.
//
//   k_mm_pix    C[pixel][n] = sum_k pro(A)[pixel][k] Bt[n][k]        pixels on the M axis (l_2, the transposed l_2 / l_last / l_1,
//               l_last as 36 columns, l_1 at widths that are not a multiple of 4)
//               prologue  APRO 1: A = relu(bn(h + bias)) formed from the pre-BN activation h while the tile is staged — the
//                         normalised activations a1 / a2 never exist in memory
//               epilogue  EPI 1: per-channel batch sums (sum, sum of squares) of C + bias       -> the slots k_bn_fin adds up
//                         EPI 2: the two batch sums of BN's backward (sum gx, sum gx * xhat; gx = C where xhat(h) > 0)
//   k_mm_kpix   part[s][m][n] = sum_{pixel in chunk s} pro(A)[pixel][m] B[pixel][n]     the filter gradients: K = the PIXELS of
//               the minibatch, cut into chunks whose partial products k_g_store_grad adds in fp64 (bit-reproducible: no atomics)
//
// Tiles.  256 threads = 4 wavefronts; a wavefront owns up to 2 x 2 accumulator tiles of 32 x 32 (64 VGPRs) and per K step of 2
// feeds 4 MFMAs (256 cycles of the matrix pipe) from 2 + 2 operand registers — one ds_read_b128 per operand tile per 4 K steps
// in k_mm_pix, one ds_read_b64 per operand pair per step in k_mm_kpix.  Both operands go global -> registers -> (prologue) -> LDS,
// double-buffered: the loads of K tile t + 1 are issued before the MFMAs of tile t and parked after them, one barrier per tile;
// 2 workgroups per CU (66 KiB of LDS, <= 128 VGPRs) cover each other's barrier and epilogue.
//   k_mm_pix LDS layout [k / 4][row (+1)][4]: a lane's float4 = 4 consecutive k of its row = its A / B registers of 4 MFMA steps
//   (lane half g takes the k4 group 2 c + g of chunk c: the K order inside a chunk is permuted identically on both operands).
//   The odd row pitch makes the 8-lane groups of ds_write_b128 and the 16-lane groups of ds_read_b128 conflict-free.
//   k_mm_kpix LDS layout [pixel][channel] = the global layout; tile t of a wavefront takes the channels base + T r + t (r = MFMA
//   row), so ONE 8-byte read serves both tiles of an operand.
// Work split.  k_mm_pix: workgroup (n tile, m slot) walks the pixel tiles m slot, m slot + gm, ... and keeps the epilogue's batch
// sums in registers until the end: one partial per channel and SLOT (the trainer's slotted reductions; every slot < nslot is
// written).  Workgroup ids are dealt to the 8 XCDs round-robin by the hardware: ids that share an XCD share the A tile (all n
// tiles of a pixel tile) resp. the pixel chunk (all output tiles of a chunk), so the operand crosses HBM -> L2 once.
//
// Replaces (reference, /root/reference): train_noise_flow.py:64-66 (one sess.run of train_op) with borealisflows/layers.py:452-498
// (real_nvp_conv_template: conv2d -> batch_norm -> relu, twice, conv2d_zeros) and TF's gradients of it, at hps.width
// (sidd/ArgParser.py:43: default 512).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <atomic>

#ifndef MM_ABL
// tools/probes/mm_probe.hip only (timing ablations of k_mm_pix; results are wrong): 1 = no barrier in the K loop, 2 = no global
// loads in it, 4 = no LDS writes in it, 8 = every pixel tile reads tile 0 of A (cache hits), 16 = no B loads, 32 = no A loads, 64 = no epilogue, 128 = every load hits one L1-resident 4 KiB
#define MM_ABL 0
#endif
#ifndef MM_DEFAULT_MODE
#define MM_DEFAULT_MODE 2   // pix_mode() without NF_MM_MODE in the environment
#endif
#ifndef MM_INTERLEAVE
// bf16 x 6: N VALU instructions of the next tile's prologue + split scheduled behind each matrix instruction of a tile's second K = 16
// step (sched_group_barrier), 0: prologue + split behind the tile's last matrix instruction.  Measured (width 512, 141 312 pixels, 4
// wavefronts): 6 -> plain 394 us, l_2 forward 453, its transpose 647 (48 more live registers: the epilogue variants spill into AGPRs);
// 0 -> 381 / 466 / 437.  Left at 0.
#define MM_INTERLEAVE 0
#endif
#ifndef MM_KREP
#define MM_KREP 1   // with MM_ABL 7: the K loop runs this many times (loop time apart from the tile boundaries)
#endif

namespace {
namespace mm {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kSlotStride = 1024;   // floats between two values of a slotted accumulator group (= NSLOT of nf_train.hip)
constexpr int kT = 256;             // threads per workgroup
constexpr int kBK = 32;             // K extent of one staged tile (k_mm_pix: channels; k_mm_kpix: pixels)
constexpr int kBM = 128;            // pixels per tile of k_mm_pix

// normalised activation (layers.py:378-401 with the batch moments) of a pre-BN value h stored WITHOUT its bias:
// ((h + b) - mean) rstd as ONE fma, h rstd + c with c = (b - mean) rstd per channel.  The ONE expression every site shares — the
// forward operand, the ReLU masks the backward pass re-derives from h, the BN backward — so a re-derived mask is the forward's, bit
// for bit.  (fp32 VALU work and the MFMA pipe do not overlap on a SIMD: 2 instead of 4 instructions per staged element is
// matrix-pipe time)
__device__ __forceinline__ float xhat_c(float b, float m, float rs) { return (b - m) * rs; }
__device__ __forceinline__ float xhat(float h, float rs, float c) { return fmaf(h, rs, c); }

// BN backward of one value (k_g_bn_bwd of nf_train_gemm.h): g = d loss / d relu output, h = the pre-BN activation; ba, bq = the batch
// means of gx and gx * xhat (gx = g where the forward's relu kept the value)
__device__ __forceinline__ float bn_bwd(float g, float h, float rs, float c, float ba, float bq)
{
    const float xh = xhat(h, rs, c);
    return rs * ((xh > 0.0f ? g : 0.0f) - ba - xh * bq);
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// 4 floats from a 4-byte aligned address (parameter blocks)
__device__ __forceinline__ float4 ld4u(const float *p) { return make_float4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ float f4c(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// 4 consecutive floats of a row, zero beyond `n`.  V = 4: the row base and `at` are 16-byte aligned and n % 4 == 0
template <int V>
__device__ __forceinline__ float4 row4(const float *row, int at, int n)
{
    if constexpr (V == 4) {
        return at < n ? ld4(row + at) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float4 v;
        v.x = at + 0 < n ? row[at + 0] : 0.0f;
        v.y = at + 1 < n ? row[at + 1] : 0.0f;
        v.z = at + 2 < n ? row[at + 2] : 0.0f;
        v.w = at + 3 < n ? row[at + 3] : 0.0f;
        return v;
    }
}

// pixels per tile of k_mm_pix by workgroup size (4 or 8 wavefronts)
constexpr int pix_bm(int NW) { return NW == 8 ? 256 : kBM; }

// floats of LDS k_mm_pix shares between its operand tiles (fp32: 2 x 8 x (BM + 1 + BN + 1) float4; bf16 x 6: 2 x 12 x (BM + 2 + BN + 2)
// 16-byte units), the finished tile on its way out ([BM][BN + 4]) and the final sums
constexpr int pix_region0_floats(int BN, int BM = kBM, int prec = 0)
{
    const int nbuf = prec && BM == kBM ? 1 : 2;   // (bf16 x 6 on 4 wavefronts: one operand buffer, see k_mm_pix)
    const int ops = prec ? nbuf * 12 * (BM + 2 + BN + 2) * 4 : 2 * 8 * (BM + 1 + BN + 1) * 4, out = BM * (BN + 4);
    return ops > out ? ops : out;
}

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

// v = v1 + v2 + v3 per element, each part a bf16 (round to nearest even), packed 4 to a uint2 per part
__device__ __forceinline__ void split_bf16x3(const float4 v, uint2 (&pl)[3])
{
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const v2bf h01 = {(__bf16)r[0], (__bf16)r[1]}, h23 = {(__bf16)r[2], (__bf16)r[3]};
        pl[q] = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
        if (q < 2) {
            r[0] -= (float)h01[0];
            r[1] -= (float)h01[1];
            r[2] -= (float)h23[0];
            r[3] -= (float)h23[1];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct PixArgs {
    int64_t P;              // pixels (rows of A and C)
    int N, K;               // output channels, reduction length
    const float *A;         // [P][lda]
    int lda;
    const float *Bt;        // packed weights [N][ldb]: 16-byte aligned, ldb % 4 == 0, zero in the columns K .. ldb - 1
    int ldb;
    float *C;               // [P][ldc]
    int ldc;
    const float *abias, *abn;        // APRO 1 / 2: bias [K] and (mean [K], rstd [K]) of the batch normalisation in front of A
    const float *A2, *abb;           // APRO 2: the pre-BN activation [P][lda] beside A = d loss / d relu output, and BN backward's two batch means (ba [K], bq [K])
    const float *ebias, *ebn, *eh;   // EPI 1: ebias [N].  EPI 2 / 3 / 4: ebias [N], ebn = (mean [N], rstd [N]), eh [P][ldh] = pre-BN activation
    const float *ebb;                // EPI 3: BN backward's two batch means (ba [N], bq [N])
    int ldh;
    float *stats;           // EPI 1 / 2 / 4: value j (j < 2 N: the N sums, then the N second sums) of slot s at stats[j * kSlotStride + s]; EPI 3: N sums
    int gm, nslot;          // m slots that take tiles (<= nslot); slots gm .. nslot - 1 are cleared
    int n_tiles, m_tiles;
};

//   WN     wavefronts along the channel axis (1 or 2); NW / WN along the pixel axis
//   TN     32-channel tiles per wavefront: the workgroup's tile is BM pixels x (32 WN TN) channels
//   NW     wavefronts per workgroup: 4 (BM = 128 pixels, two workgroups per CU) or 8 (BM = 256, one per CU: 25 % fewer operand bytes per
//          flop through the staging path — the tile DESIGN.md §4.7's ablation asked for)
//   PREC   0: exact fp32 products on v_mfma_f32_32x32x2_f32.
//          1: "bf16 x 6" on v_mfma_f32_32x32x16_bf16 — every operand element is split while its tile is staged into three bf16 numbers
//             a = a1 + a2 + a3 (round-to-nearest each: |a - a1 - a2 - a3| <= 2^-27 |a|), and a b is taken as the six exact products
//             a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1), added in fp32 smallest first; the three dropped products are
//             <= 2^-26 |a b| together — a quarter of the rounding of ONE fp32 product — so the result carries fp32 round-off, not bf16's.
//             6 matrix instructions of 32 cycles per 32 x 32 x 16 block instead of 8 of 64: 2.67 x the fp32 matrix rate.
//   APRO   0: A as stored;  1: A = relu(xhat(A + bias));  2: A = BN backward of (A masked by xhat(A2) > 0)
//   EPI    0: store;  1: + batch sums of C + bias;  2: + masked batch sums against xhat(eh);  4: those sums WITHOUT the store;
//          3: store BN backward of the masked C (the batch means are known: second pass over a cheap product) + its column sums
//   AV     4: A rows are 16-byte aligned and K % 4 == 0 (one global_load_dwordx4 per 4 k);  1: any width / alignment
//   CV     4: C (and eh) rows are 16-byte aligned and N % 4 == 0: the tile leaves through LDS as whole 16-byte pieces of its rows
//          (one global_store_dwordx4 / global_load_dwordx4 per 4 channels, 512 contiguous bytes per 32 lanes);  1: D registers
//          straight to memory, 4 bytes per lane
template <int WN, int TN, int APRO, int EPI, int AV, int CV, int NW = 4, int PREC = 0>
__global__ __launch_bounds__(64 * NW) void k_mm_pix(const PixArgs a)
{
    constexpr int kT = 64 * NW, kBM = pix_bm(NW);
    constexpr int WM = NW / WN, TM = kBM / (32 * WM), BN = WN * TN * 32;
    constexpr int SR8 = kT / 8;                // rows one staging pass covers (a thread stages 4 consecutive k of a row)
    constexpr int NA = kBM / SR8;              // A-tile float4s per thread
    constexpr int NB = BN / SR8;               // B-tile float4s per thread
    static_assert(NA == 4 && NB >= 1 && TM >= 1, "tile / workgroup shape");
    // PREC 0: float4 units per k4 row (odd pitch).  PREC 1: 16-byte units (8 bf16 = one lane's operand of a K = 16 step) per
    // (plane, k8 group) row — pitch = 2 mod 8 units keeps the 8-byte staging stores of a 16-lane group on distinct banks
    constexpr int AP = PREC ? kBM + 2 : kBM + 1, BP = PREC ? BN + 2 : BN + 1;
    extern __shared__ __attribute__((aligned(16))) float mm_smem[];
    constexpr int SP = BN + 4;                 // row pitch (floats) of the tile on its way out (CV 4)
    constexpr int R0 = pix_region0_floats(BN, kBM, PREC); // the operand tiles; between two K loops the finished tile [BM][SP]; at the end the sums
    constexpr int KG = PREC ? 12 : 8;          // rows of AP / BP units per buffer: 8 k4 groups, or 3 planes x 4 k8 groups
    // bf16 x 6 on 4 wavefronts: ONE operand buffer (50 KiB; two would leave one 4-wavefront workgroup per CU) and two barriers per K
    // tile — two workgroups per CU cover each other's barriers and epilogues, as the fp32 kernel's do
    constexpr bool SB = PREC == 1 && NW == 4;
    constexpr int NBUF = SB ? 1 : 2;
    float4 *const sA = reinterpret_cast<float4 *>(mm_smem);   // [NBUF][KG][AP]
    float4 *const sB = sA + NBUF * KG * AP;                    // [NBUF][KG][BP]
    float *const stg = mm_smem;                                // [kBM][SP]
    float *const red = mm_smem;                                // [2][WM or kT / (BN / 4)][BN]
    float *const cst = mm_smem + R0;                           // APRO 1: [2][Kc] (rstd, c = (bias - mean) rstd); APRO 2: [4][Kc] (+ ba, bq)
    const int Kc = (a.K + kBK - 1) / kBK * kBK;

    // ids that share an XCD (id % 8) and are neighbours there share the pixel tile
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    const int nt = q % a.n_tiles, ms = (q / a.n_tiles) * 8 + xcd;
    if (ms >= a.gm) return;

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 31, g = lane >> 5;
    const int wm = wv / WN, wn = wv % WN;
    const int j = tid & 7, r0 = tid >> 3;      // staging: k4 group j of the rows r0 + SR8 i
    const int n0 = nt * BN;

    // (Measured and dropped: starting the second workgroup of every CU half a tile late, so that one's epilogue falls into the
    // other's K loop — 670 vs 672 us for l_2 at width 512: the two are not in lockstep to begin with.)
    if constexpr (APRO != 0) {
        for (int i = tid; i < Kc; i += kT) {
            const float rs = i < a.K ? a.abn[a.K + i] : 0.0f;
            cst[i] = rs;
            cst[Kc + i] = i < a.K ? xhat_c(a.abias[i], a.abn[i], rs) : 0.0f;
            if constexpr (APRO == 2) {
                cst[2 * Kc + i] = i < a.K ? a.abb[i] : 0.0f;
                cst[3 * Kc + i] = i < a.K ? a.abb[a.K + i] : 0.0f;
            }
        }
        __syncthreads();
    }

    // epilogue constants: a lane's channels are n0 + wn TN 32 + tn 32 + n for every tile it ever computes
    [[maybe_unused]] float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};   // CV 4: this thread's 4 channels (n0 + 4 (tid % (BN / 4)) ...)
    [[maybe_unused]] float eb[TN], er[TN], ec[TN], ea[TN], eq[TN], ssum[TN], qsum[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int ch = n0 + (wn * TN + tn) * 32 + n;
        eb[tn] = (EPI != 0 && ch < a.N) ? a.ebias[ch] : 0.0f;
        er[tn] = (EPI >= 2 && ch < a.N) ? a.ebn[a.N + ch] : 0.0f;
        ec[tn] = (EPI >= 2 && ch < a.N) ? xhat_c(eb[tn], a.ebn[ch], er[tn]) : 0.0f;
        ea[tn] = (EPI == 3 && ch < a.N) ? a.ebb[ch] : 0.0f;
        eq[tn] = (EPI == 3 && ch < a.N) ? a.ebb[a.N + ch] : 0.0f;
        ssum[tn] = 0.0f;
        qsum[tn] = 0.0f;
    }

    // Operand addresses = a wavefront-UNIFORM base (tile origin + K position: scalar registers, advanced by scalar adds) + a per-lane
    // 32-bit byte offset that never changes (row within the tile, k4 group): the loads take the scalar-base form and the K loop
    // spends no vector instruction on addresses.
    // B rows of this workgroup's tile (fixed over the pixel tiles); rows beyond N: a valid row is loaded and multiplied away
    const float *const bbase = a.Bt + (size_t)n0 * a.ldb;
    uint32_t boff[NB];
    float bmask[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = r0 + SR8 * i;
        bmask[i] = n0 + r < a.N ? 1.0f : 0.0f;
        boff[i] = (MM_ABL & 128) ? (uint32_t)(tid * 16) : (uint32_t)((n0 + r < a.N ? r : 0) * a.ldb + 4 * j) * 4u;
    }
    const bool bragged = n0 + BN > a.N;   // workgroup-uniform
    const int nkt = Kc / kBK;
    const int nkt_full = a.K / kBK;   // K tiles every load of which is in range (ldb >= K)
    // (Measured and dropped: every workgroup starting its K loop at a K tile of its own, so that workgroups in step do not pull the
    // same 128 bytes of every 2 KiB row at a time — 649 vs 651 us for the plain product at width 512: the L2 channel hash already
    // spreads them.  What the feed costs is issue, not the memory system: with every load hitting one L1-resident 4 KiB the kernel
    // still runs 618 us against 587 us without loads; K loop alone, no tile boundaries: 0.975 of the matrix-pipe rate.)

    // the A side of the tile being staged: origin and per-lane offsets (set by `origin`).  (Issuing a tile's FIRST loads before the
    // previous tile's epilogue — so that the round trip at a tile boundary hides behind the stores — was measured: no gain, the
    // other workgroup of the CU already covers it, and 38 more live registers)
    const float *abase = a.A;
    [[maybe_unused]] const float *hbase = a.A2;
    uint32_t aoff[4];
    auto origin = [&](int64_t mt) {
        const int64_t m0 = mt * kBM;
        abase = a.A + ((MM_ABL & (8 | 128)) ? 0 : m0) * a.lda;
        if constexpr (APRO == 2) hbase = a.A2 + m0 * a.lda;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = r0 + SR8 * i;
            r = m0 + r < a.P ? r : (int)(a.P - 1 - m0);   // rows past the end: loaded (valid memory), never stored nor summed
            aoff[i] = (MM_ABL & 128) ? (uint32_t)(tid * 16) : (uint32_t)(r * a.lda + 4 * j) * 4u;
        }
    };
    // Staging registers (ONE set: the loads of K tile t + 1 are issued before the MFMAs of tile t and parked after them.  A
    // second set with two tiles of lead was measured SLOWER, 630 -> 654 us for the plain 512 product: the parking wait is not
    // what the feed costs — removing the loads altogether gains 10 %, removing the barrier 2 %)
    struct Stage {
        float4 ra[4], rb[NB], rh[APRO == 2 ? 4 : 1];
    };
    Stage SR;
    auto at = [](const float *base, uint32_t off) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off); };
    auto fetch = [&](int kt, Stage &R) {
        const float *const ak = abase + ((MM_ABL & 128) ? 0 : kt * kBK), *const bk = bbase + ((MM_ABL & 128) ? 0 : kt * kBK);
        [[maybe_unused]] const float *const hk = hbase + kt * kBK;
        if (kt < nkt_full) {     // workgroup-uniform: the whole tile is inside K, no predicates around the loads
            if (!(MM_ABL & 32)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) R.ra[i] = row4<AV>(at(ak, aoff[i]), 0, 4);
            }
            if constexpr (APRO == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) R.rh[i] = row4<AV>(at(hk, aoff[i]), 0, 4);
            }
            if (!(MM_ABL & 16)) {
#pragma unroll
            for (int i = 0; i < NB; ++i) R.rb[i] = ld4(at(bk, boff[i]));
            }
        } else {
            const int k = kt * kBK + 4 * j;
#pragma unroll
            for (int i = 0; i < 4; ++i) R.ra[i] = row4<AV>(at(ak, aoff[i]), 0, a.K - k);
            if constexpr (APRO == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) R.rh[i] = row4<AV>(at(hk, aoff[i]), 0, a.K - k);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) R.rb[i] = k < a.ldb ? ld4(at(bk, boff[i])) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // PREC 1: the staged operands split into their three bf16 parts (registers), between the prologue and the LDS stores — pure VALU
    // work that `compute` interleaves with the matrix instructions of the tile before
    struct Parts {
        uint2 a[PREC ? 4 : 1][3], b[PREC ? NB : 1][3];
    };
    [[maybe_unused]] Parts PK;
    auto prologue = [&](int kt, Stage &R) {
        if constexpr (APRO == 1) {
            const int k = kt * kBK + 4 * j;
            const float4 cr = ld4(cst + k), cc = ld4(cst + Kc + k);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                R.ra[i].x = fmaxf(xhat(R.ra[i].x, cr.x, cc.x), 0.0f);
                R.ra[i].y = fmaxf(xhat(R.ra[i].y, cr.y, cc.y), 0.0f);
                R.ra[i].z = fmaxf(xhat(R.ra[i].z, cr.z, cc.z), 0.0f);
                R.ra[i].w = fmaxf(xhat(R.ra[i].w, cr.w, cc.w), 0.0f);
            }
        } else if constexpr (APRO == 2) {
            const int k = kt * kBK + 4 * j;
            const float4 cr = ld4(cst + k), cc = ld4(cst + Kc + k), ca = ld4(cst + 2 * Kc + k), cq = ld4(cst + 3 * Kc + k);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                R.ra[i].x = bn_bwd(R.ra[i].x, R.rh[i].x, cr.x, cc.x, ca.x, cq.x);
                R.ra[i].y = bn_bwd(R.ra[i].y, R.rh[i].y, cr.y, cc.y, ca.y, cq.y);
                R.ra[i].z = bn_bwd(R.ra[i].z, R.rh[i].z, cr.z, cc.z, ca.z, cq.z);
                R.ra[i].w = bn_bwd(R.ra[i].w, R.rh[i].w, cr.w, cc.w, ca.w, cq.w);
            }
        }
        if constexpr (PREC == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) split_bf16x3(R.ra[i], PK.a[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float4 v = R.rb[i];
                if (bragged) v = make_float4(v.x * bmask[i], v.y * bmask[i], v.z * bmask[i], v.w * bmask[i]);
                split_bf16x3(v, PK.b[i]);
            }
        }
    };
    auto store = [&](int buf, Stage &R) {
        if constexpr (PREC == 1) {
            // plane p, k8 group j >> 1, row, half j & 1 of the 16-byte unit: one 8-byte store per plane and staged float4
            uint2 *da = reinterpret_cast<uint2 *>(sA + (buf * KG + (j >> 1)) * AP + r0) + (j & 1);
            uint2 *db = reinterpret_cast<uint2 *>(sB + (buf * KG + (j >> 1)) * BP + r0) + (j & 1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) da[2 * ((q * 4) * AP + SR8 * i)] = PK.a[i][q];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) db[2 * ((q * 4) * BP + SR8 * i)] = PK.b[i][q];
        } else {
        float4 *da = sA + (buf * 8 + j) * AP + r0, *db = sB + (buf * 8 + j) * BP + r0;
#pragma unroll
        for (int i = 0; i < 4; ++i) da[SR8 * i] = R.ra[i];
        if (bragged) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                db[SR8 * i] = make_float4(R.rb[i].x * bmask[i], R.rb[i].y * bmask[i], R.rb[i].z * bmask[i], R.rb[i].w * bmask[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) db[SR8 * i] = R.rb[i];
        }
        }
    };
    for (int64_t mt = ms; mt < a.m_tiles; mt += a.gm) {
        const int64_t m0 = mt * kBM;
        origin(mt);
        fetch(0, SR);
        v16f acc[TM][TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[tm][tn][v] = 0.0f;

        // the 4 chunks of 8 k of one staged tile.  Every tile runs all of them: beyond K both operands hold zeros (a ragged K costs
        // MFMA time in the last tile only; guarding the chunks with uniform branches was measured SLOWER — the wait-count pass then
        // also waits for the prefetched operands at every join).  Operand registers in ping-pong: the LDS reads of chunk c + 1 are
        // ISSUED before the 16 MFMAs of chunk c (left alone, the machine scheduler sinks them to just before their first use and
        // every chunk starts with an exposed LDS round trip)
        // `next`: the K tile whose staged operands (already in SR) get their prologue + split between this tile's matrix instructions
        auto compute = [&](int buf, int next) {
            if constexpr (PREC == 1) {
                // K = 16 step s: lane half g holds the 8 k of k8 group 2 s + g of its row — in each of the three planes
                const float4 *pa = sA + (buf * KG + g) * AP + wm * TM * 32 + n;
                const float4 *pb = sB + (buf * KG + g) * BP + wn * TN * 32 + n;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    v8bf af[3][TM], bf[3][TN];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) af[q][tm] = __builtin_bit_cast(v8bf, pa[(q * 4 + 2 * st) * AP + tm * 32]);
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) bf[q][tn] = __builtin_bit_cast(v8bf, pb[(q * 4 + 2 * st) * BP + tn * 32]);
                    }
                    // the next tile's operands arrived while step 0 ran: their prologue and split are VALU work for the 28 idle issue
                    // cycles behind every matrix instruction of step 1 (left in front of the LDS stores they would run with the matrix pipe idle)
                    if (MM_INTERLEAVE && st == 1 && next >= 0) prologue(next, SR);
                    // smallest products first: (a1 b3, a3 b1, a2 b2), (a1 b2, a2 b1), a1 b1
                    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t6]][tm], bf[PB[t6]][tn], acc[tm][tn], 0, 0, 0);
#if MM_INTERLEAVE
                    if (st == 1 && next >= 0) {
#pragma unroll
                        for (int t6 = 0; t6 < 6 * TM * TN; ++t6) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // one matrix instruction
                            __builtin_amdgcn_sched_group_barrier(0x002, MM_INTERLEAVE, 0);  // ... then VALU of the split
                        }
                    }
#endif
                }
                if (!MM_INTERLEAVE && next >= 0) prologue(next, SR);
                return;
            }
            const float4 *pa = sA + (buf * 8 + g) * AP + wm * TM * 32 + n;
            const float4 *pb = sB + (buf * 8 + g) * BP + wn * TN * 32 + n;
            float4 af[2][TM], bf[2][TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[0][tm] = pa[tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[0][tn] = pb[tn * 32];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c + 1 < 4) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) af[(c + 1) & 1][tm] = pa[2 * (c + 1) * AP + tm * 32];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bf[(c + 1) & 1][tn] = pb[2 * (c + 1) * BP + tn * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(af[c & 1][tm], s), f4c(bf[c & 1][tn], s), acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (next >= 0) prologue(next, SR);   // (fp32: behind the matrix instructions, where it always was)
        };

        prologue(0, SR);
        store(0, SR);
        __syncthreads();
        for (int kt = 0; kt < nkt * MM_KREP; ++kt) {
            const int buf = SB ? 0 : kt & 1;
            const bool more = kt + 1 < nkt;
            if (more && !(MM_ABL & 2)) fetch(kt + 1, SR);
            compute(buf, more && !(MM_ABL & 4) ? kt + 1 : -1);
            if constexpr (SB) __syncthreads();   // every wavefront has read the tile
            if (more && !(MM_ABL & 4)) store(SB ? 0 : buf ^ 1, SR);
            if (!(MM_ABL & 1)) __syncthreads();
        }

        // ---- epilogue: D register v of lane (n, g) = pixel 8 (v >> 2) + 4 g + (v & 3) of the tile, channel n ----
        if ((MM_ABL & 64) && acc[0][0][0] != 12345.0f) continue;
        if constexpr (CV == 4) {
            // through LDS (the operand tiles are dead: every wavefront is past the K loop's last barrier): D registers -> [pixel][SP]
            // (32 lanes = 32 consecutive channels: conflict-free), then every thread takes 16-byte pieces of the rows — its 4 channels
            // are the same for every row and every tile, so the batch sums stay in 8 registers until the kernel ends
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    float *d = stg + ((wm * TM + tm) * 32 + 4 * g) * SP + (wn * TN + tn) * 32 + n;
#pragma unroll
                    for (int v = 0; v < 16; ++v) d[(8 * (v >> 2) + (v & 3)) * SP] = acc[tm][tn][v];
                }
            __syncthreads();
            constexpr int Q = BN / 4, RS = kT / Q, PASS = kBM / RS;   // float4 per row, rows per pass, passes
            const int c4 = tid % Q, rr = tid / Q, ch = n0 + 4 * c4;
            if (ch < a.N) {
                [[maybe_unused]] float4 kb, kr, kc, ka, kq;
                if constexpr (EPI != 0) kb = ld4u(a.ebias + ch);
                if constexpr (EPI >= 2) {
                    const float4 km = ld4u(a.ebn + ch);
                    kr = ld4u(a.ebn + a.N + ch);
                    kc = make_float4(xhat_c(kb.x, km.x, kr.x), xhat_c(kb.y, km.y, kr.y), xhat_c(kb.z, km.z, kr.z), xhat_c(kb.w, km.w, kr.w));
                }
                if constexpr (EPI == 3) {
                    ka = ld4u(a.ebb + ch);
                    kq = ld4u(a.ebb + a.N + ch);
                }
                // (loads and stores share one in-order counter: the pre-BN activations of HALF of the thread's pieces are requested
                // together, before their stores — two memory round trips per tile; all 16 at once need 64 registers beside the next
                // tile's staged operands and cost the kernel its second wavefront per SIMD)
                constexpr int GRP = PASS >= 8 ? PASS / 2 : PASS;
#pragma unroll
                for (int i0 = 0; i0 < PASS; i0 += GRP) {
                [[maybe_unused]] float4 hv[EPI >= 2 ? GRP : 1];
                if constexpr (EPI >= 2) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) {
                        const int64_t p = m0 + rr + RS * (i0 + i);
                        hv[i] = p < a.P ? ld4(a.eh + p * a.ldh + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int r = rr + RS * (i0 + i);
                    const int64_t p = m0 + r;
                    if (p >= a.P) continue;
                    float4 v = *reinterpret_cast<const float4 *>(stg + r * SP + 4 * c4);
                    if constexpr (EPI == 1) {
                        const float x0 = v.x + kb.x, x1 = v.y + kb.y, x2 = v.z + kb.z, x3 = v.w + kb.w;
                        s4[0] += x0; s4[1] += x1; s4[2] += x2; s4[3] += x3;
                        q4[0] = fmaf(x0, x0, q4[0]); q4[1] = fmaf(x1, x1, q4[1]); q4[2] = fmaf(x2, x2, q4[2]); q4[3] = fmaf(x3, x3, q4[3]);
                    } else if constexpr (EPI == 2 || EPI == 4) {
                        const float h0 = xhat(hv[i].x, kr.x, kc.x), h1 = xhat(hv[i].y, kr.y, kc.y), h2 = xhat(hv[i].z, kr.z, kc.z), h3 = xhat(hv[i].w, kr.w, kc.w);
                        const float g0 = h0 > 0.f ? v.x : 0.f, g1 = h1 > 0.f ? v.y : 0.f, g2 = h2 > 0.f ? v.z : 0.f, g3 = h3 > 0.f ? v.w : 0.f;
                        s4[0] += g0; s4[1] += g1; s4[2] += g2; s4[3] += g3;
                        q4[0] = fmaf(g0, h0, q4[0]); q4[1] = fmaf(g1, h1, q4[1]); q4[2] = fmaf(g2, h2, q4[2]); q4[3] = fmaf(g3, h3, q4[3]);
                    } else if constexpr (EPI == 3) {
                        v.x = bn_bwd(v.x, hv[i].x, kr.x, kc.x, ka.x, kq.x);
                        v.y = bn_bwd(v.y, hv[i].y, kr.y, kc.y, ka.y, kq.y);
                        v.z = bn_bwd(v.z, hv[i].z, kr.z, kc.z, ka.z, kq.z);
                        v.w = bn_bwd(v.w, hv[i].w, kr.w, kc.w, ka.w, kq.w);
                        s4[0] += v.x; s4[1] += v.y; s4[2] += v.z; s4[3] += v.w;
                    }
                    if constexpr (EPI != 4) *reinterpret_cast<float4 *>(a.C + p * a.ldc + ch) = v;
                }
                }
            }
            __syncthreads();   // the next tile's operands overwrite the region
        } else {
        const bool whole = m0 + kBM <= a.P && n0 + BN <= a.N;   // workgroup-uniform: no per-element predicates
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int ch = n0 + (wn * TN + tn) * 32 + n;
                const bool chok = ch < a.N;
                const int64_t pb0 = m0 + (wm * TM + tm) * 32 + 4 * g;
                float *const cp = a.C + pb0 * a.ldc + ch;
                [[maybe_unused]] const float *const hp = EPI >= 2 ? a.eh + pb0 * a.ldh + ch : nullptr;
                // the 16 pre-BN activations this lane needs, requested together and BEFORE any store of the tile: loads and stores
                // share one in-order counter (vmcnt), so a load issued behind a store waits for that store's round trip
                [[maybe_unused]] float hv[EPI >= 2 ? 16 : 1];
                if constexpr (EPI >= 2) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int dp = 8 * (v >> 2) + (v & 3);
                        hv[v] = (whole || (chok && pb0 + dp < a.P)) ? hp[(int64_t)dp * a.ldh] : 0.0f;
                    }
                }
                auto one = [&](int v) {
                    const int dp = 8 * (v >> 2) + (v & 3);
                    const float val = acc[tm][tn][v];
                    if constexpr (EPI <= 2) cp[(int64_t)dp * a.ldc] = val;
                    if constexpr (EPI == 1) {
                        const float x = val + eb[tn];
                        ssum[tn] += x;
                        qsum[tn] = fmaf(x, x, qsum[tn]);
                    } else if constexpr (EPI == 2 || EPI == 4) {
                        const float xh = xhat(hv[v], er[tn], ec[tn]);
                        const float gx = xh > 0.0f ? val : 0.0f;
                        ssum[tn] += gx;
                        qsum[tn] = fmaf(gx, xh, qsum[tn]);
                    } else if constexpr (EPI == 3) {
                        const float o = bn_bwd(val, hv[v], er[tn], ec[tn], ea[tn], eq[tn]);
                        cp[(int64_t)dp * a.ldc] = o;
                        ssum[tn] += o;
                    }
                };
                if (whole) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) one(v);
                } else if (chok) {
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (pb0 + 8 * (v >> 2) + (v & 3) < a.P) one(v);
                }
            }
            }
    }

    if constexpr (EPI != 0 && CV == 4) {
        // one partial per channel and slot: the kT / (BN / 4) threads that share 4 channels meet in LDS
        constexpr int Q = BN / 4, RS = kT / Q;
        const int c4 = tid % Q, rr = tid / Q;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[rr * BN + 4 * c4 + i] = s4[i];
            red[(RS + rr) * BN + 4 * c4 + i] = q4[i];
        }
        __syncthreads();
        for (int c = tid; c < BN; c += kT) {
            const int ch = n0 + c;
            if (ch >= a.N) continue;
            float s = 0.0f, qq = 0.0f;
#pragma unroll
            for (int w = 0; w < RS; ++w) {
                s += red[w * BN + c];
                qq += red[(RS + w) * BN + c];
            }
            float *ps = a.stats + (size_t)ch * kSlotStride, *pq = a.stats + (size_t)(a.N + ch) * kSlotStride;
            ps[ms] = s;
            if (EPI != 3) pq[ms] = qq;
            for (int k = ms + a.gm; k < a.nslot; k += a.gm) {   // the slots nobody owns
                ps[k] = 0.0f;
                if (EPI != 3) pq[k] = 0.0f;
            }
        }
    }
    if constexpr (EPI != 0 && CV != 4) {
        // one partial per channel and slot: lane halves, then the WM wavefronts that share the channels
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            ssum[tn] += __shfl_xor(ssum[tn], 32);
            qsum[tn] += __shfl_xor(qsum[tn], 32);
        }
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                red[wm * BN + (wn * TN + tn) * 32 + n] = ssum[tn];
                red[(WM + wm) * BN + (wn * TN + tn) * 32 + n] = qsum[tn];
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += kT) {
            const int ch = n0 + c;
            if (ch >= a.N) continue;
            float s = 0.0f, qq = 0.0f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s += red[w * BN + c];
                qq += red[(WM + w) * BN + c];
            }
            float *ps = a.stats + (size_t)ch * kSlotStride, *pq = a.stats + (size_t)(a.N + ch) * kSlotStride;
            ps[ms] = s;
            if (EPI != 3) pq[ms] = qq;
            for (int k = ms + a.gm; k < a.nslot; k += a.gm) {   // the slots nobody owns
                ps[k] = 0.0f;
                if (EPI != 3) pq[k] = 0.0f;
            }
        }
    }
}

template <int WN, int TN, int NW = 4, int PREC = 0>
constexpr size_t pix_lds_bytes(int K, int apro)
{
    constexpr int BN = WN * TN * 32;
    return ((size_t)pix_region0_floats(BN, pix_bm(NW), PREC) + (apro == 2 ? 4 : apro ? 2 : 0) * (size_t)((K + kBK - 1) / kBK * kBK)) * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct KpixArgs {
    int64_t npix;
    int M, N;               // output rows (channels of A), columns (channels of B)
    const float *A;         // [npix][lda]
    int lda;
    const float *B;         // [npix][ldb]
    int ldb;
    float *part;            // [S][M][N]
    int64_t part_cap;       // floats `part` holds (0: kGradPartFloats) — the launch never leaves more partial products than fit
    int64_t chunk;          // pixels per partial product (a multiple of kBK)
    int S;                  // partial products
    int m_tiles, n_tiles;
    const float *abias, *abn;   // APRO 1: bias [M], (mean [M], rstd [M]): A = relu(xhat(A + bias))
    const float *B2, *bbias, *bbn, *bbb;   // BPRO 2: B = BN backward of (B masked by xhat(B2) > 0): pre-BN activation [npix][ldb], bias [N], (mean, rstd), (ba, bq)
    float *dbias;               // BPRO 2: the column sums of that B (= d bias) as slotted partials: value n of slot s at dbias[n * kSlotStride + s]
    int nslot;                  // BPRO 2: S <= nslot; the slots S .. nslot - 1 are cleared
};

//   WMv       wavefronts along M (1, 2, 4); 4 / WMv along N
//   TM, TN    32-channel tiles per wavefront along M / N (1 or 2): the workgroup's tile is (32 WMv TM) x (32 (4 / WMv) TN)
//   AV, BV    4: the operand's rows are 16-byte aligned and its width is a multiple of 4;  1: anything
//   BPRO      0: B as stored;  2: B = BN backward of the masked B, formed while the tile is staged (and its column sums = d bias)
//   APRO      0: A as stored;  1: A = relu(xhat(A + bias));  3: TWO products from one pass over A — part[s][0] with A = relu(xhat)
//             and part[s][1] with A = the ReLU mask (xhat > 0): with B = G36 the first is d l_last/W and, contracted with the
//             filter, sum gx xhat of BN2's backward; the second contracted with the filter is sum gx (k_g_bnb_from_parts) — the
//             batch sums of BN2's backward without a pass over a [pixel][w] tensor of their own
template <int WMv, int TM, int TN, int APRO, int AV, int BV, int BPRO = 0>
__global__ __launch_bounds__(kT) void k_mm_kpix(const KpixArgs a)
{
    constexpr int WNv = 4 / WMv, BM = WMv * TM * 32, BN = WNv * TN * 32;
    constexpr int NA = APRO == 3 ? 2 : 1;      // A tiles staged / accumulator sets
    extern __shared__ __attribute__((aligned(16))) float mm_smem[];
    float *const sA = mm_smem;                 // [2][NA][kBK][BM]
    float *const sB = sA + 2 * NA * kBK * BM;  // [2][kBK][BN]

    const int tiles = a.m_tiles * a.n_tiles;
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    const int tile = q % tiles, s = (q / tiles) * 8 + xcd;   // the tiles of one pixel chunk share an XCD
    if (s >= a.S) return;
    const int m0 = (tile / a.n_tiles) * BM, n0 = (tile % a.n_tiles) * BN;
    const int64_t p0 = (int64_t)s * a.chunk, p1 = p0 + a.chunk < a.npix ? p0 + a.chunk : a.npix;

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 31, g = lane >> 5;
    const int wm = wv / WNv, wn = wv % WNv;

    // staging: thread -> a fixed channel group (so the prologue's constants stay in registers), RS pixel rows apart
    constexpr int AQ = AV == 4 ? BM / 4 : BM, ARS = kT / AQ, APASS = kBK / ARS;
    constexpr int BQ = BV == 4 ? BN / 4 : BN, BRS = kT / BQ, BPASS = kBK / BRS;
    static_assert(kT % AQ == 0 && kT % BQ == 0 && kBK % ARS == 0 && kBK % BRS == 0 && APASS >= 1 && BPASS >= 1, "staging split");
    const int acq = tid % AQ, apr = tid / AQ, bcq = tid % BQ, bpr = tid / BQ;
    const int ac = m0 + AV * acq, bc = n0 + BV * bcq;      // first channel this thread stages
    float4 cr = make_float4(0.f, 0.f, 0.f, 0.f), cc = cr;      // APRO 1 / 3: rstd and c = (bias - mean) rstd of this thread's channels
    if constexpr (APRO == 1 || APRO == 3) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = i < AV && ac + i < a.M;
            t[i] = ok ? a.abn[a.M + ac + i] : 0.0f;
            t[4 + i] = ok ? xhat_c(a.abias[ac + i], a.abn[ac + i], t[i]) : 0.0f;
        }
        cr = make_float4(t[0], t[1], t[2], t[3]);
        cc = make_float4(t[4], t[5], t[6], t[7]);
    }

    [[maybe_unused]] float kr[4], kc[4], ka[4], kq[4], dsum[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BPRO == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = i < BV && bc + i < a.N;
            kr[i] = ok ? a.bbn[a.N + bc + i] : 0.0f;
            kc[i] = ok ? xhat_c(a.bbias[bc + i], a.bbn[bc + i], kr[i]) : 0.0f;
            ka[i] = ok ? a.bbb[bc + i] : 0.0f;
            kq[i] = ok ? a.bbb[a.N + bc + i] : 0.0f;
        }
    }
    const bool sums = BPRO == 2 && tile / a.n_tiles == 0;   // one workgroup row adds up d bias (every m tile stages the same B)

    v16f acc[NA][TM][TN];
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[na][tm][tn][v] = 0.0f;

    // Operand addresses = a wavefront-uniform base (chunk position: scalar registers) + a per-lane 32-bit byte offset that never
    // changes (pixel row within the staged tile, channel): scalar-base loads, no vector address arithmetic in the K loop
    uint32_t aoff[APASS], boff[BPASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) aoff[i] = (uint32_t)((apr + ARS * i) * a.lda + AV * acq) * 4u;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) boff[i] = (uint32_t)((bpr + BRS * i) * a.ldb + BV * bcq) * 4u;
    auto at = [](const float *base, uint32_t off) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off); };
    const bool a_in = m0 + BM <= a.M, b_in = n0 + BN <= a.N;   // workgroup-uniform: every staged channel of the operand exists

    float4 ra[AV == 4 ? APASS : (APASS + 3) / 4], rb[BV == 4 ? BPASS : (BPASS + 3) / 4];
    float sa1[AV == 4 ? 1 : APASS], sb1[BV == 4 ? 1 : BPASS];
    [[maybe_unused]] float4 rh[(BPRO == 2 && BV == 4) ? BPASS : 1];
    [[maybe_unused]] float sh1[(BPRO == 2 && BV == 1) ? BPASS : 1];
    [[maybe_unused]] int64_t bpk = 0;   // first pixel of the tile the B registers hold
    auto fetch = [&](int64_t pk) {
        const float *const ab = a.A + pk * a.lda + m0, *const bb = a.B + pk * a.ldb + n0;
        [[maybe_unused]] const float *const hb = BPRO == 2 ? a.B2 + pk * a.ldb + n0 : nullptr;
        const bool rows_in = pk + kBK <= p1;                // workgroup-uniform: no predicates around the loads of a whole tile
        bpk = pk;
        if (rows_in && a_in && b_in) {                      // the common case as ONE straight block (measured: split into a block per
            if constexpr (AV == 4) {                        // operand the 512 x 512 product lost 8 %)
#pragma unroll
                for (int i = 0; i < APASS; ++i) ra[i] = ld4(at(ab, aoff[i]));
            } else {
#pragma unroll
                for (int i = 0; i < APASS; ++i) sa1[i] = *at(ab, aoff[i]);
            }
            if constexpr (BV == 4) {
#pragma unroll
                for (int i = 0; i < BPASS; ++i) {
                    rb[i] = ld4(at(bb, boff[i]));
                    if constexpr (BPRO == 2) rh[i] = ld4(at(hb, boff[i]));
                }
            } else {
#pragma unroll
                for (int i = 0; i < BPASS; ++i) {
                    sb1[i] = *at(bb, boff[i]);
                    if constexpr (BPRO == 2) sh1[i] = *at(hb, boff[i]);
                }
            }
            return;
        }
        if (rows_in && a_in) {
            if constexpr (AV == 4) {
#pragma unroll
                for (int i = 0; i < APASS; ++i) ra[i] = ld4(at(ab, aoff[i]));
            } else {
#pragma unroll
                for (int i = 0; i < APASS; ++i) sa1[i] = *at(ab, aoff[i]);
            }
        } else {
            if constexpr (AV == 4) {
#pragma unroll
                for (int i = 0; i < APASS; ++i) ra[i] = (pk + apr + ARS * i < p1 && ac < a.M) ? ld4(at(ab, aoff[i])) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int i = 0; i < APASS; ++i) sa1[i] = (pk + apr + ARS * i < p1 && ac < a.M) ? *at(ab, aoff[i]) : 0.0f;
            }
        }
        if constexpr (BV == 4) {
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const bool ok = pk + bpr + BRS * i < p1 && bc < a.N;
                rb[i] = ok ? ld4(at(bb, boff[i])) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (BPRO == 2) rh[i] = ok ? ld4(at(hb, boff[i])) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const bool ok = pk + bpr + BRS * i < p1 && bc < a.N;
                sb1[i] = ok ? *at(bb, boff[i]) : 0.0f;
                if constexpr (BPRO == 2) sh1[i] = ok ? *at(hb, boff[i]) : 0.0f;
            }
        }
    };
    auto park = [&](int buf) {
        float *da = sA + buf * NA * kBK * BM, *db = sB + buf * kBK * BN;
        if constexpr (AV == 4) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                float4 v = ra[i];
                if constexpr (APRO == 1 || APRO == 3) {
                    v.x = fmaxf(xhat(v.x, cr.x, cc.x), 0.0f);
                    v.y = fmaxf(xhat(v.y, cr.y, cc.y), 0.0f);
                    v.z = fmaxf(xhat(v.z, cr.z, cc.z), 0.0f);
                    v.w = fmaxf(xhat(v.w, cr.w, cc.w), 0.0f);
                }
                *reinterpret_cast<float4 *>(da + (apr + ARS * i) * BM + 4 * acq) = v;
                if constexpr (APRO == 3)      // the ReLU mask: relu(xhat) > 0 <=> xhat > 0 (rows past the chunk: a zero row of B beside them)
                    *reinterpret_cast<float4 *>(da + kBK * BM + (apr + ARS * i) * BM + 4 * acq) =
                        make_float4(v.x > 0.f ? 1.f : 0.f, v.y > 0.f ? 1.f : 0.f, v.z > 0.f ? 1.f : 0.f, v.w > 0.f ? 1.f : 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {
                float v = sa1[i];
                if constexpr (APRO == 1 || APRO == 3) v = fmaxf(xhat(v, cr.x, cc.x), 0.0f);
                da[(apr + ARS * i) * BM + acq] = v;
                if constexpr (APRO == 3) da[kBK * BM + (apr + ARS * i) * BM + acq] = v > 0.f ? 1.f : 0.f;
            }
        }
        if constexpr (BV == 4) {
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                float4 v = rb[i];
                if constexpr (BPRO == 2) {
                    const bool in = bpk + bpr + BRS * i < p1 && bc < a.N;     // rows past the chunk stay zero (and out of d bias)
                    v.x = in ? bn_bwd(v.x, rh[i].x, kr[0], kc[0], ka[0], kq[0]) : 0.0f;
                    v.y = in ? bn_bwd(v.y, rh[i].y, kr[1], kc[1], ka[1], kq[1]) : 0.0f;
                    v.z = in ? bn_bwd(v.z, rh[i].z, kr[2], kc[2], ka[2], kq[2]) : 0.0f;
                    v.w = in ? bn_bwd(v.w, rh[i].w, kr[3], kc[3], ka[3], kq[3]) : 0.0f;
                    dsum[0] += v.x; dsum[1] += v.y; dsum[2] += v.z; dsum[3] += v.w;
                }
                *reinterpret_cast<float4 *>(db + (bpr + BRS * i) * BN + 4 * bcq) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                float v = sb1[i];
                if constexpr (BPRO == 2) {
                    const bool in = bpk + bpr + BRS * i < p1 && bc < a.N;
                    v = in ? bn_bwd(v, sh1[i], kr[0], kc[0], ka[0], kq[0]) : 0.0f;
                    dsum[0] += v;
                }
                db[(bpr + BRS * i) * BN + bcq] = v;
            }
        }
    };

    const int nkt = (int)((p1 - p0 + kBK - 1) / kBK);
    if (nkt > 0) {
        fetch(p0);
        park(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) fetch(p0 + (int64_t)(kt + 1) * kBK);
        const float *pa = sA + buf * NA * kBK * BM + g * BM + wm * TM * 32 + TM * n;
        const float *pb = sB + buf * kBK * BN + g * BN + wn * TN * 32 + TN * n;
        if constexpr (TM * TN == 4 && NA == 1) {
            // operand registers in a ring of 4 steps: the LDS reads of step st + 3 are ISSUED before the MFMAs of step st (left alone
            // the compiler reads each step's pair right before its MFMAs and waits for it: an exposed LDS round trip per 4 MFMAs).
            // d l_2/W at width 512: 610 -> 578 us
            constexpr int NST = kBK / 2, LEAD = 3;
            float av[4][TM], bv[4][TN];
            auto rd = [&](int st) {
                const float2 ta = *reinterpret_cast<const float2 *>(pa + 2 * st * BM), tb = *reinterpret_cast<const float2 *>(pb + 2 * st * BN);
                av[st & 3][0] = ta.x;
                av[st & 3][1] = ta.y;
                bv[st & 3][0] = tb.x;
                bv[st & 3][1] = tb.y;
            };
#pragma unroll
            for (int st = 0; st < LEAD; ++st) rd(st);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + LEAD < NST) rd(st + LEAD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[0][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 3][tm], bv[st & 3][tn], acc[0][tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // (with 1 or 2 MFMAs per step the pinned order LOSES — 100 -> 143 us for d l_last/W: the scheduler's own order stays)
#pragma unroll
            for (int st = 0; st < kBK / 2; ++st) {
                float av[NA][TM], bv[TN];
#pragma unroll
                for (int na = 0; na < NA; ++na) {
                    if constexpr (TM == 2) {
                        const float2 t = *reinterpret_cast<const float2 *>(pa + na * kBK * BM + 2 * st * BM);
                        av[na][0] = t.x;
                        av[na][1] = t.y;
                    } else {
                        av[na][0] = pa[na * kBK * BM + 2 * st * BM];
                    }
                }
                if constexpr (TN == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(pb + 2 * st * BN);
                    bv[0] = t.x;
                    bv[1] = t.y;
                } else {
                    bv[0] = pb[2 * st * BN];
                }
#pragma unroll
                for (int na = 0; na < NA; ++na)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[na][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[na][tm], bv[tn], acc[na][tm][tn], 0, 0, 0);
            }
        }
        if (kt + 1 < nkt) park(buf ^ 1);
        __syncthreads();
    }

    // D register v of lane (n, g): MFMA row r = 8 (v >> 2) + 4 g + (v & 3) -> channel m0 + wm TM 32 + TM r + tm; column n -> n0 + wn TN 32 + TN n + tn
    float *const out = a.part + (size_t)s * NA * a.M * a.N;     // [NA][M][N] per chunk
#pragma unroll
    for (int na = 0; na < NA; ++na)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = n0 + wn * TN * 32 + TN * n + tn;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = m0 + wm * TM * 32 + TM * (8 * (v >> 2) + 4 * g + (v & 3)) + tm;
                    if (row < a.M && col < a.N) out[((size_t)na * a.M + row) * a.N + col] = acc[na][tm][tn][v];
                }
            }

    if constexpr (BPRO == 2) {
        if (!sums) return;                       // workgroup-uniform
        // the BRS threads that staged the same channels meet in LDS (the tiles are dead); one partial per channel and chunk = slot
        float *const red = mm_smem;              // [BRS][BN]
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BV; ++i) red[bpr * BN + BV * bcq + i] = dsum[i];
        __syncthreads();
        for (int c = tid; c < BN; c += kT) {
            const int ch = n0 + c;
            if (ch >= a.N) continue;
            float tot = 0.0f;
            for (int r = 0; r < BRS; ++r) tot += red[r * BN + c];
            float *pd = a.dbias + (size_t)ch * kSlotStride;
            pd[s] = tot;
            for (int k = s + a.S; k < a.nslot; k += a.S) pd[k] = 0.0f;   // the slots nobody owns
        }
    }
}

// k_mm_kpix at M, N multiples of 4 on the bf16 matrix pipe ("bf16 x 6", see k_mm_pix PREC 1): part[s][m][n] = sum over the pixels of
// chunk s of pro(A)[pixel][m] B[pixel][n] with the error of fp32 products.  128 x 128 channels per workgroup (4 wavefronts, 2 x 2
// tiles of 32 x 32 each), 32 pixels per staged tile, ONE operand buffer (50 KiB: two workgroups per CU cover each other's barriers).
// The reduction index is the pixel, which is the SLOW index of both operands in memory: a staging thread therefore takes 4 channels
// x 8 consecutive pixels (8 x 16-byte loads, 512 contiguous bytes per 32 lanes each), so that after the split it holds, per channel
// and part, exactly one lane operand of a K = 16 step (8 bf16 of consecutive pixels = 16 bytes) and stores it with ds_write_b128
// into k_mm_pix's layout [part][k8 group][row][16 B].  Channel 4 q + c of a tile sits in LDS row 32 c + q (consecutive lanes ->
// consecutive rows: conflict-free stores), i.e. the 32 x 32 accumulator tile c of a wavefront holds the channels = c (mod 4).
//   APRO  0: A as stored;  1: A = relu(xhat(A + bias))
template <int APRO>
__global__ __launch_bounds__(kT) void k_mm_kpix6(const KpixArgs a)
{
    constexpr int BM = 128, BN = 128, AP = BM + 2, BP = BN + 2, KG = 12;
    extern __shared__ __attribute__((aligned(16))) float mm_smem[];
    float4 *const sA = reinterpret_cast<float4 *>(mm_smem);   // [KG][AP]
    float4 *const sB = sA + KG * AP;                           // [KG][BP]
    const int tiles = a.m_tiles * a.n_tiles;
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    const int tile = q % tiles, s = (q / tiles) * 8 + xcd;   // the tiles of one pixel chunk share an XCD
    if (s >= a.S) return;
    const int m0 = (tile / a.n_tiles) * BM, n0 = (tile % a.n_tiles) * BN;
    const int64_t p0 = (int64_t)s * a.chunk, p1 = p0 + a.chunk < a.npix ? p0 + a.chunk : a.npix;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, n = lane & 31, g = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    // staging task of this thread: operand (A: threads 0 .. 127, B: 128 .. 255), channel quad cq, k8 group kg
    const bool isA = tid < 128;
    const int cq = tid & 31, kg = (tid >> 5) & 3;
    const int ch = (isA ? m0 : n0) + 4 * cq;
    const bool ch_ok = ch < (isA ? a.M : a.N);                // (widths are multiples of 4: the whole quad exists or none of it)
    const float *const src = isA ? a.A : a.B;
    const int ld = isA ? a.lda : a.ldb;
    float4 cr = make_float4(0.f, 0.f, 0.f, 0.f), cc = cr;
    if constexpr (APRO == 1) {
        if (isA && ch_ok) {
            cr = ld4u(a.abn + a.M + ch);
            const float4 mb = ld4u(a.abias + ch), mm_ = ld4u(a.abn + ch);
            cc = make_float4(xhat_c(mb.x, mm_.x, cr.x), xhat_c(mb.y, mm_.y, cr.y), xhat_c(mb.z, mm_.z, cr.z), xhat_c(mb.w, mm_.w, cr.w));
        }
    }
    v16f acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[tm][tn][v] = 0.0f;

    float4 rg[8];
    auto fetch = [&](int64_t pk) {
        const float *const b0 = src + (pk + 8 * kg) * ld + ch;
#pragma unroll
        for (int i = 0; i < 8; ++i) rg[i] = (ch_ok && pk + 8 * kg + i < p1) ? ld4(b0 + (int64_t)i * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    uint4 pk4[4][3];   // [channel of the quad][part]: 8 pixels x bf16
    auto prep = [&]() {
        if constexpr (APRO == 1) {
            if (isA) {       // (rows past the chunk were loaded as zeros: relu(xhat(0)) need not be zero, so they are masked again below)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    rg[i].x = fmaxf(xhat(rg[i].x, cr.x, cc.x), 0.0f);
                    rg[i].y = fmaxf(xhat(rg[i].y, cr.y, cc.y), 0.0f);
                    rg[i].z = fmaxf(xhat(rg[i].z, cr.z, cc.z), 0.0f);
                    rg[i].w = fmaxf(xhat(rg[i].w, cr.w, cc.w), 0.0f);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = f4c(rg[i], c);
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                uint32_t w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v2bf h = {(__bf16)r[2 * i], (__bf16)r[2 * i + 1]};
                    w[i] = __builtin_bit_cast(uint32_t, h);
                    if (part < 2) {
                        r[2 * i] -= (float)h[0];
                        r[2 * i + 1] -= (float)h[1];
                    }
                }
                pk4[c][part] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    };
    [[maybe_unused]] int64_t fetched = 0;
    auto store = [&]() {
        float4 *const d = (isA ? sA : sB) + kg * (isA ? AP : BP) + cq;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int part = 0; part < 3; ++part)
                d[(part * 4) * (isA ? AP : BP) + 32 * c] = __builtin_bit_cast(float4, pk4[c][part]);
    };
    const int nkt = (int)((p1 - p0 + kBK - 1) / kBK);
    // APRO 1 turns the zeros loaded for pixels past the chunk into relu(c): those pixels' B rows are zero, which is what keeps them out
    if (nkt > 0) {
        fetch(p0);
        prep();
        store();
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) fetch(p0 + (int64_t)(kt + 1) * kBK);
        const float4 *pa = sA + g * AP + wm * 64 + n;
        const float4 *pb = sB + g * BP + wn * 64 + n;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            v8bf af[3][2], bf[3][2];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[part][t] = __builtin_bit_cast(v8bf, pa[(part * 4 + 2 * st) * AP + t * 32]);
                    bf[part][t] = __builtin_bit_cast(v8bf, pb[(part * 4 + 2 * st) * BP + t * 32]);
                }
            }
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // smallest products first
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t6]][tm], bf[PB[t6]][tn], acc[tm][tn], 0, 0, 0);
        }
        if (more) prep();
        __syncthreads();   // every wavefront has read the tile
        if (more) store();
        __syncthreads();
    }
    // D register v of lane (n, g): MFMA row r = 8 (v >> 2) + 4 g + (v & 3) of tile (wm, tm) = channel m0 + 4 r + (2 wm + tm);
    // column n of tile (wn, tn) = channel n0 + 4 n + (2 wn + tn): a lane's two column tiles are neighbours -> one 8-byte store
    float *const out = a.part + (size_t)s * a.M * a.N;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + 4 * (8 * (v >> 2) + 4 * g + (v & 3)) + 2 * wm + tm;
            const int col = n0 + 4 * n + 2 * wn;
            if (row < a.M && col < a.N) *reinterpret_cast<float2 *>(out + (size_t)row * a.N + col) = make_float2(acc[tm][0][v], acc[tm][1][v]);
        }
}
constexpr size_t kpix6_lds_bytes() { return (size_t)12 * (130 + 130) * 16; }

template <int WMv, int TM, int TN>
constexpr size_t kpix_lds_bytes(int na = 1)
{
    return (size_t)2 * kBK * (na * WMv * TM * 32 + (4 / WMv) * TN * 32) * sizeof(float);   // (>= the [BRS][BN] floats of the d-bias reduction)
}

// ---- weights into the layout k_mm_pix reads: dst [rows][ld], ld % 4 == 0, zero beyond `cols` ----------------------------------
//   mode 0  dst[r][c] = src[r * cols + c]                         (Bt = the matrix as stored)
//   mode 1  dst[r][c] = src[c * rows + r]                         (Bt = its transpose)
//   mode 2  dst[tap*4 + k][i] = l_last/W[tap][i][k], i < w        (rows = 36, cols = w: the 36 columns of l_last's transposed evaluation)
//   mode 3  dst[i][tap*4 + k] = l_last/W[tap][i][k]               (rows = w, cols = 36)
__global__ void k_mm_pack(int mode, int rows, int cols, int ld, int w, const float *__restrict__ src, float *__restrict__ dst)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * ld) return;
    const int r = e / ld, c = e - r * ld;
    float v = 0.0f;
    if (c < cols) {
        if (mode == 0) v = src[(size_t)r * cols + c];
        else if (mode == 1) v = src[(size_t)c * rows + r];
        else if (mode == 2) v = src[((size_t)(r >> 2) * (w + 1) + c) * 4 + (r & 3)];
        else v = src[((size_t)(c >> 2) * (w + 1) + r) * 4 + (c & 3)];
    }
    dst[e] = v;
}

// ---- launches (host) ----------------------------------------------------------------------------------------------------------
struct Ctx {
    int n_cu, device;
};

// floats of partial products one filter gradient may leave (k_mm_kpix: S x M x N)
constexpr int64_t kGradPartFloats = (int64_t)1 << 24;   // 64 MiB per filter

// dynamic LDS beyond 64 KiB has to be enabled per kernel (once per device; racy but idempotent).  `cur` = the caller's static
// per-instantiation record of what was enabled so far
inline bool mm_enable_lds(const void *fn, size_t lds, int device, std::atomic<size_t> (&cur)[16])
{
    std::atomic<size_t> &c = cur[device & 15];
    if (lds > c.load(std::memory_order_relaxed) || device > 15) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        c.store(lds, std::memory_order_relaxed);
    }
    return true;
}

// C[P x N] = pro(A) . Bt^T with the prologue / epilogue of the template (nf_train_mm.h); `a` carries everything but the work split
template <int WN, int TN, int APRO, int EPI, int AV, int CV, int NW = 4, int PREC = 0>
inline bool mm_pix_launch(const Ctx &cx, hipStream_t st, PixArgs a)
{
    constexpr int BN = WN * TN * 32, BM = pix_bm(NW);
    a.n_tiles = (a.N + BN - 1) / BN;
    a.m_tiles = (int)((a.P + BM - 1) / BM);
    // without batch sums: one workgroup per tile, the dispatcher balances (persistent workgroups, 2 or 4 per CU: measured, no difference).  With them: one SLOT per workgroup, each walking its
    // share of the pixel tiles — two workgroups per CU in flight
    int gm = a.m_tiles;
    if (EPI != 0) {
        // (measured: 3 or 4 per CU lose 2 - 5 % at widths 64 - 256; so does trimming the count to the fewest workgroups with the same
        // longest share — 368 instead of 512 at width 128 leave the CUs unevenly filled: 4.53 vs 4.33 ms per step)
        const int want = std::max(1, (NW == 8 ? 1 : 2) * cx.n_cu / a.n_tiles);
        gm = std::max(1, std::min(std::min(a.nslot, a.m_tiles), want));
    }
    a.gm = gm;
    const size_t lds = pix_lds_bytes<WN, TN, NW, PREC>(a.K, APRO);
    auto fn = &k_mm_pix<WN, TN, APRO, EPI, AV, CV, NW, PREC>;
    static std::atomic<size_t> enabled[16];
    if (!mm_enable_lds(reinterpret_cast<const void *>(fn), lds, cx.device, enabled)) return false;
    const unsigned grid = (unsigned)((gm + 7) / 8 * 8 * a.n_tiles);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * NW), lds, st, a);
    return true;
}

// How the 128-column products (widths beyond 64: the l_2 products) run.  0: fp32, 4 wavefronts on 128 x 128 (two workgroups per CU);
// 1: fp32, 8 wavefronts on 256 x 128; 2: bf16 x 6 (k_mm_pix, PREC 1), 8 wavefronts on 256 x 128; 3: bf16 x 6, 4 wavefronts on 128 x 128, one
// operand buffer.  NF_MM_MODE overrides (A/B aid, read once).
inline int pix_mode(int64_t P, int n_cu)
{
    static const int forced = [] { const char *e = getenv("NF_MM_MODE"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced;                                       // (every size: the probe checks small products this way)
    if (MM_DEFAULT_MODE == 1 || MM_DEFAULT_MODE == 2) return P >= (int64_t)pix_bm(8) * n_cu / 2 ? MM_DEFAULT_MODE : 0;   // 256-pixel tiles want enough of them to fill the chip
    return MM_DEFAULT_MODE;
}
// the tile width follows the channel count: 128 / 64 / 32 columns
template <int APRO, int EPI, int AV, int CV>
inline bool mm_pix_cv(const Ctx &cx, hipStream_t st, const PixArgs &a)
{
    if (a.N > 64) {
        const int mode = pix_mode(a.P, cx.n_cu);
        if (mode == 3) return mm_pix_launch<2, 2, APRO, EPI, AV, CV, 4, 1>(cx, st, a);
        if (mode == 2) return mm_pix_launch<2, 2, APRO, EPI, AV, CV, 8, 1>(cx, st, a);
        if (mode == 1) return mm_pix_launch<2, 2, APRO, EPI, AV, CV, 8, 0>(cx, st, a);
        return mm_pix_launch<2, 2, APRO, EPI, AV, CV>(cx, st, a);
    }
    if (a.N > 32) return mm_pix_launch<1, 2, APRO, EPI, AV, CV>(cx, st, a);
    return mm_pix_launch<1, 1, APRO, EPI, AV, CV>(cx, st, a);
}
// ... and the way the tile leaves follows the alignment of C (and of the pre-BN activation beside it)
template <int APRO, int EPI, int AV>
inline bool mm_pix(const Ctx &cx, hipStream_t st, const PixArgs &a)
{
    // (measured: a tile that is only STORED leaves faster straight from the D registers — stores are fire-and-forget, the LDS pass
    // costs two barriers; one that also READS the pre-BN activation gains from one 16-byte request per 4 channels)
    const bool vec = EPI >= 2 && a.N % 4 == 0 && (EPI == 4 || (a.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0)) && a.ldh % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(a.eh) & 15) == 0;
    return vec ? mm_pix_cv<APRO, EPI, AV, 4>(cx, st, a) : mm_pix_cv<APRO, EPI, AV, 1>(cx, st, a);
}

// part[s][M][N] = sum over the pixels of chunk s of pro(A)^T . B; returns the number of partial products (0 on failure)
template <int WMv, int TM, int TN, int APRO, int AV, int BV, int BPRO = 0>
inline int mm_kpix_launch(const Ctx &cx, hipStream_t st, KpixArgs a)
{
    constexpr int BM = WMv * TM * 32, BN = (4 / WMv) * TN * 32;
    a.m_tiles = (a.M + BM - 1) / BM;
    a.n_tiles = (a.N + BN - 1) / BN;
    const int tiles = a.m_tiles * a.n_tiles;
    // workgroups over the whole launch: ~4 per CU, ~2 per CU for the 2 x 2-tile products (64 accumulator registers, 64 KiB of LDS:
    // two are resident, and more chunks are only more partial products to write and add up — width 512, 138 patches: d l_2/W on
    // 32 instead of 64 chunks and the dual product on 128 instead of 256 took the step from 22.7 to 22.3 ms)
    const int per_cu = TM * TN == 4 ? 2 : 4;
    int64_t S = std::max<int64_t>(1, (per_cu * (int64_t)cx.n_cu + tiles - 1) / tiles);
    // at most 256 partial products (128 of the dual product); 512 where one or two tiles cover the output (widths <= 128: the launch
    // is short and wants the CUs 2 - 4 workgroups deep — measured at 138 patches: width 64 2.93 -> 2.75 ms, 128 4.72 -> 4.55; width
    // 512 loses 1.5 % with it)
    const int scap = APRO == 3 ? 128 : (a.M <= 128 && a.N <= 128) ? 512 : 256;
    const int64_t cap = a.part_cap > 0 ? std::min<int64_t>(a.part_cap, kGradPartFloats) : kGradPartFloats;
    if (cap < (APRO == 3 ? 2 : 1) * (int64_t)a.M * a.N) return 0;
    S = std::min<int64_t>(S, std::min<int64_t>(scap, cap / ((APRO == 3 ? 2 : 1) * (int64_t)a.M * a.N)));
    S = std::min<int64_t>(S, std::max<int64_t>(1, a.npix / (4 * kBK)));            // chunks of at least 128 pixels
    if (BPRO == 2) S = std::min<int64_t>(S, std::max(1, a.nslot));                     // one d-bias slot per chunk
    a.chunk = ((a.npix + S - 1) / S + kBK - 1) / kBK * kBK;
    a.S = (int)((a.npix + a.chunk - 1) / a.chunk);
    if constexpr (WMv == 2 && TM == 2 && TN == 2 && APRO <= 1 && AV == 4 && BV == 4 && BPRO == 0) {
        if (pix_mode(a.npix, cx.n_cu) >= 2 && a.M % 4 == 0 && a.N % 4 == 0) {   // the bf16 x 6 twin (same tiles, same chunks)
            const unsigned grid6 = (unsigned)((a.S + 7) / 8 * 8 * tiles);
            hipLaunchKernelGGL(k_mm_kpix6<APRO>, dim3(grid6), dim3(kT), kpix6_lds_bytes(), st, a);
            return a.S;
        }
    }
    const size_t lds = kpix_lds_bytes<WMv, TM, TN>(APRO == 3 ? 2 : 1);
    auto fn = &k_mm_kpix<WMv, TM, TN, APRO, AV, BV, BPRO>;
    static std::atomic<size_t> enabled[16];
    if (!mm_enable_lds(reinterpret_cast<const void *>(fn), lds, cx.device, enabled)) return 0;
    const unsigned grid = (unsigned)((a.S + 7) / 8 * 8 * tiles);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kT), lds, st, a);
    return a.S;
}

// weights into the packed layout k_mm_pix reads (nf_train_mm.h: k_mm_pack); returns the row pitch
inline int mm_pack(hipStream_t st, int mode, int rows, int cols, int w, const float *src, float *dst)
{
    const int ld = (cols + 3) & ~3;
    hipLaunchKernelGGL(k_mm_pack, dim3((unsigned)((rows * ld + 255) / 256)), dim3(256), 0, st, mode, rows, cols, ld, w, src, dst);
    return ld;
}


// every packed layout of one coupling in ONE launch (the weights do not change between a step's forward and backward pass):
//   [w2t: w x w4 | w2: w x w4 | w3a: 36 x w4 | w3b: w x 36 | w1: 18 x w4 | w1t: w x 20]      (w4 = w rounded up to 4)
struct PackAll {
    size_t o_w2t, o_w2, o_w3a, o_w3b, o_w1, o_w1t, total;
    int w4;
};
inline PackAll pack_layout(int w)
{
    PackAll L;
    L.w4 = (w + 3) & ~3;
    L.o_w2t = 0;
    L.o_w2 = L.o_w2t + (size_t)w * L.w4;
    L.o_w3a = L.o_w2 + (size_t)w * L.w4;
    L.o_w3b = L.o_w3a + 36 * (size_t)L.w4;
    L.o_w1 = L.o_w3b + (size_t)w * 36;
    L.o_w1t = L.o_w1 + 18 * (size_t)L.w4;
    L.total = L.o_w1t + (size_t)w * 20;
    return L;
}
// blockIdx.y = the coupling: its parameter block starts at P + C.off[y] (l_1/W, then l_2/W at + 21 w, l_last/W at + 24 w + w w), its
// packed weights at dst + y * dst_stride
struct PackCpl {
    static constexpr int kMax = 32;
    int off[kMax];
};
__global__ void k_mm_pack_all(int w, PackAll L, const float *__restrict__ P, PackCpl C, float *__restrict__ dst0, size_t dst_stride)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.total) return;
    const float *const W1 = P + C.off[blockIdx.y], *const W2 = W1 + 21 * (size_t)w, *const W3 = W1 + 24 * (size_t)w + (size_t)w * w;
    float *const dst = dst0 + blockIdx.y * dst_stride;
    const int w4 = L.w4;
    float v = 0.0f;
    if (e < L.o_w2) {                     // Bt[j][i] = W2[i][j]
        const int r = (int)(e / w4), c = (int)(e - (size_t)r * w4);
        if (c < w) v = W2[(size_t)c * w + r];
    } else if (e < L.o_w3a) {             // Bt[i][j] = W2[i][j]
        const size_t f = e - L.o_w2;
        const int r = (int)(f / w4), c = (int)(f - (size_t)r * w4);
        if (c < w) v = W2[(size_t)r * w + c];
    } else if (e < L.o_w3b) {             // Bt[tap*4+k][i] = l_last/W[tap][i][k]
        const size_t f = e - L.o_w3a;
        const int r = (int)(f / w4), c = (int)(f - (size_t)r * w4);
        if (c < w) v = W3[((size_t)(r >> 2) * (w + 1) + c) * 4 + (r & 3)];
    } else if (e < L.o_w1) {              // Bt[i][tap*4+k] = l_last/W[tap][i][k]
        const size_t f = e - L.o_w3b;
        const int r = (int)(f / 36), c = (int)(f - (size_t)r * 36);
        v = W3[((size_t)(c >> 2) * (w + 1) + r) * 4 + (c & 3)];
    } else if (e < L.o_w1t) {             // Bt[tap*2+c][j] = W1[tap*2+c][j]
        const size_t f = e - L.o_w1;
        const int r = (int)(f / w4), c = (int)(f - (size_t)r * w4);
        if (c < w) v = W1[(size_t)r * w + c];
    } else {                              // Bt[j][tap*2+c] = W1[tap*2+c][j]
        const size_t f = e - L.o_w1t;
        const int r = (int)(f / 20), c = (int)(f - (size_t)r * 20);
        if (c < 18) v = W1[(size_t)c * w + r];
    }
    dst[e] = v;
}

}  // namespace mm
}  // namespace
