// nf_train_pr.h — part of nf_train.hip (included inside its anonymous namespace; not a standalone header).
// Patch-resident training stages for the paper-scale coupling CNN (width 32, job_noise_flow.sh:18) on 32x32 patches.
// ---------------------------------------------------------------------------------------------------------------------------
// The stage kernels of nf_train_wide.h keep every [pixel][32] tensor of a coupling in HBM (h1, h2, the masked gradients t1 / t2:
// ~17 tensor passes, 2.4 GB per coupling at 1 024 patches) and cut the step wherever one of them changes hands.  Here the step is
// cut ONLY where batch normalisation forces it (two batch-statistics barriers per coupling and direction), and between two cuts
// one workgroup holds a whole patch: the coupling CNN is RE-COMPUTED from the 16 bytes per pixel of its input (l_1 and l_2 are
// 1.9 kMAC per pixel — cheaper than reading one [pixel][32] tensor back), activations and gradients live in registers in the
// matrix instruction's own result layout, and the only tensors in HBM are 16-byte-per-pixel ones (z, u, gu, dz).
//
//   forward   F0: (mix,) l_1                      -> batch sums of h1
//             F1: l_1, BN1, ReLU, l_2             -> batch sums of h2
//             F2: ... BN2, ReLU, l_last, affine   -> z', u (l_last's raw output), log-det share
//   backward  A : affine / tanh backward -> gu;  ... transposed l_last, ReLU mask -> the two batch sums of BN2's backward;  d l_last/W
//             B : ... BN2 backward, transposed l_2, ReLU mask                    -> the two batch sums of BN1's backward;  d l_2/W, d l_2/b
//             C : ... BN1 backward, transposed l_1 (+ the folded Conv2d1x1 backward) -> d loss / d z;  d l_1/W, d l_1/b, d A
//
// Layout (as nf_wide.hip): NW = 8 (or 4) wavefronts, a wavefront owns a strip of 32 / NW image rows, a TILE is one row (32 pixels on
// the N axis of v_mfma_f32_32x32x2_f32, lane n = lane & 31, lane half g = lane >> 5 = the instruction's K slice).  D register v of
// lane half g holds channel c(v, g) = 8 (v >> 2) + 4 g + (v & 3) of the lane's pixel and IS the B operand of K step v of the next
// layer, forward and transposed alike.  The filter gradients are products over the PIXELS: the two operands go through a
// wavefront-private [pixel][36] LDS tile (4 x 16-byte writes, 16 x 4-byte reads per operand and tile, no conflicts).
// l_last forward and the transposed l_1 are evaluated as P[pixel][(tap, j)] + shift-add, with the taps of the transposed
// convolution mirrored so that both use the same code.
//
// Reference: layers.py:251-375 (AffineCoupling), :378-401 (batch norm under is_training), :463-497 (the CNN),
// train_noise_flow.py:50-77,187-198 (the step).
constexpr int PR_WP = 34, PR_PL = 34 * 34, PR_RP = 36;
// -DNF_PR_TIMELINE (a tools/build_variant.sh build, OBJ=nf_train): thread 0 of every workgroup of stage A stamps the 100 MHz counter at
// its phase boundaries into PrBwdArgs::dz_out (unused by that stage) as int64[grid][16]; pr_coupling_backward prints their means
#ifdef NF_PR_TIMELINE
#define PR_TL(i) do { if (STAGE == 0 && threadIdx.x == 0) reinterpret_cast<long long *>(a.dz_out)[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PR_TL(i) do { } while (0)
#endif
// per-coupling weight image (floats), every A operand in the order the lanes fetch it (k_pr_pack)
constexpr int PR_A1 = 0;        // [3][64][4]  l_1: step = tap (9 of 12), lane l: W1[tap][ch = l >> 5][i = l & 31]
constexpr int PR_B1 = 768;      // [2][16]     b1[c(v, g)]
constexpr int PR_A2 = 800;      // [4][64][4]  l_2: step s, lane l: W2[in = c(s, l >> 5)][out = l & 31]
constexpr int PR_B2 = 1824;     // [2][16]
constexpr int PR_A3 = 1856;     // [4][64][4]  l_last as P = W3^T a2 (rows (a, g', j) as NF4_IMG_A3, raw weights)
constexpr int PR_A3C = 2880;    // [4][8][4]   its centre tap on v_mfma_f32_4x4x1
constexpr int PR_E = 3008;      // [16][4]     b3 + the indicator-channel weights of the taps outside the image, by border mask
constexpr int PR_FWD = 3072;
constexpr int PR_A3T = 3072;    // [5][64][4]  transposed l_last: step s (18 of 20) = (tap = s >> 1, q = 2 (l >> 5) + (s & 1)), W3[tap][ch = l & 31][q]
constexpr int PR_A2T = 4352;    // [4][64][4]  transposed l_2: step s, lane l: W2[in = l & 31][out = c(s, l >> 5)]
constexpr int PR_A1Q = 5376;    // [4][64][4]  transposed l_1 as Q = W1 g_h1, rows as PR_A3 with MIRRORED taps, j < 2
constexpr int PR_A1QC = 6400;   // [4][8][4]
constexpr int PR_SIZE = 6528;
__host__ __device__ constexpr int pr_chan(int v, int g) { return 8 * (v >> 2) + 4 * g + (v & 3); }

struct PrOffs {
    int off[kMaxLayers];   // raw parameter offset of every coupling, by TLayer::aux
};

__global__ __launch_bounds__(256) void k_pr_pack(const float *__restrict__ P, PrOffs offs, float *__restrict__ imgs)
{
    constexpr int W = 32;
    const int off = offs.off[blockIdx.x];
    const float *W1 = P + off, *b1 = W1 + 18 * W, *W2 = P + off + 21 * W, *b2 = W2 + W * W, *W3 = P + off + 24 * W + W * W,
                *b3 = W3 + 36 * (W + 1);
    float *img = imgs + (size_t)blockIdx.x * PR_SIZE;
    auto tap_of = [](int a, int gp) { return a == 0 ? (gp ? 6 : 0) : a == 1 ? (gp ? 8 : 2) : a == 2 ? (gp ? 7 : 1) : (gp ? 5 : 3); };
    for (int e = blockIdx.y * 256 + threadIdx.x; e < PR_SIZE; e += 256 * gridDim.y) {
        float v = 0.0f;
        if (e < PR_B1) {
            const int q = e, l = (q >> 2) & 63, step = (q >> 8) * 4 + (q & 3);
            if (step < 9) v = W1[(step * 2 + (l >> 5)) * W + (l & 31)];
        } else if (e < PR_A2) {
            const int i = e - PR_B1;
            v = b1[pr_chan(i & 15, i >> 4)];
        } else if (e < PR_B2) {
            const int q = e - PR_A2, l = (q >> 2) & 63, s = (q >> 8) * 4 + (q & 3);
            v = W2[pr_chan(s, l >> 5) * W + (l & 31)];
        } else if (e < PR_A3) {
            const int i = e - PR_B2;
            v = b2[pr_chan(i & 15, i >> 4)];
        } else if (e < PR_A3C) {
            const int q = e - PR_A3, l = (q >> 2) & 63, s = (q >> 8) * 4 + (q & 3), i = l & 31;
            v = W3[tap_of(i >> 3, (i >> 2) & 1) * (W + 1) * 4 + pr_chan(s, l >> 5) * 4 + (i & 3)];
        } else if (e < PR_E) {
            const int q = e - PR_A3C, idx = q >> 2, s = (idx >> 3) * 4 + (q & 3), gg = (idx >> 2) & 1, j = idx & 3;
            v = W3[4 * (W + 1) * 4 + pr_chan(s, gg) * 4 + j];
        } else if (e < PR_FWD) {
            const int q = e - PR_E, bm = q >> 2, j = q & 3;
            v = b3[j];
            for (int tap = 0; tap < 9; ++tap) {
                const int di = tap / 3, dj = tap % 3;
                const bool out = (di == 0 && (bm & 1)) || (di == 2 && (bm & 2)) || (dj == 0 && (bm & 4)) || (dj == 2 && (bm & 8));
                if (out) v += W3[tap * (W + 1) * 4 + W * 4 + j];
            }
        } else if (e < PR_A2T) {
            const int q = e - PR_A3T, l = (q >> 2) & 63, s = (q >> 8) * 4 + (q & 3);
            if (s < 18) v = W3[(s >> 1) * (W + 1) * 4 + (l & 31) * 4 + 2 * (l >> 5) + (s & 1)];
        } else if (e < PR_A1Q) {
            const int q = e - PR_A2T, l = (q >> 2) & 63, s = (q >> 8) * 4 + (q & 3);
            v = W2[(l & 31) * W + pr_chan(s, l >> 5)];
        } else if (e < PR_A1QC) {
            const int q = e - PR_A1Q, l = (q >> 2) & 63, s = (q >> 8) * 4 + (q & 3), i = l & 31, j = i & 3;
            if (j < 2) v = W1[((8 - tap_of(i >> 3, (i >> 2) & 1)) * 2 + j) * W + pr_chan(s, l >> 5)];
        } else {
            const int q = e - PR_A1QC, idx = q >> 2, s = (idx >> 3) * 4 + (q & 3), gg = (idx >> 2) & 1, j = idx & 3;
            if (j < 2) v = W1[(4 * 2 + j) * W + pr_chan(s, gg)];
        }
        img[e] = v;
    }
}

// tanh and exp of the affine transform on the hardware's exp2 / rcp (as the evaluation kernels: |error| ~ 1e-7 absolute / 2 ulp; the
// library calls cost ~40 instructions each, 8 per pixel in the forward's last stage and in the backward's first)
__device__ __forceinline__ float pr_tanh(float x)
{
    return fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f), -2.0f, 1.0f);
}
__device__ __forceinline__ float pr_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// value of the lane one pixel to the left / right (DPP wave_shr:1 / wave_shl:1; callers multiply the tile ends away)
__device__ __forceinline__ float pr_from_prev(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float pr_from_next(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true)); }
// a(lanes 0-31) + a(lanes 32-63) in the low half, b(lanes 0-31) + b(lanes 32-63) in the high half
__device__ __forceinline__ float pr_half_sums(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// l_1 of one tile: zb = z tile + g * PL + r * 34 + n (tap (di, dj) at + di * 34 + dj); bias included
__device__ __forceinline__ v16f pr_l1(const float4 *wb4, const float *zb, int lane, int g)
{
    v16f d;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bb = wb4[PR_B1 / 4 + g * 4 + q];
        d[4 * q + 0] = bb.x; d[4 * q + 1] = bb.y; d[4 * q + 2] = bb.z; d[4 * q + 3] = bb.w;
    }
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
        const float4 aw = wb4[PR_A1 / 4 + grp * 64 + lane];
        const float as[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tap = grp * 4 + s;
            if (tap < 9) d = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], zb[(tap / 3) * PR_WP + tap % 3], d, 0, 0, 0);
        }
    }
    return d;
}
// x <- xhat = x * rstd + c, c = -mean * rstd: ONE fma per element and the one expression every stage shares, so that a ReLU mask
// re-derived in a later stage is the forward's.  Constants by (g, v): bnc[g * 16 + v] = c, bnc[32 + g * 16 + v] = rstd
__device__ __forceinline__ void pr_bn(v16f &x, const float *bnc, int g)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 m = *reinterpret_cast<const float4 *>(bnc + g * 16 + 4 * q), r = *reinterpret_cast<const float4 *>(bnc + 32 + g * 16 + 4 * q);
        x[4 * q + 0] = fmaf(x[4 * q + 0], r.x, m.x);
        x[4 * q + 1] = fmaf(x[4 * q + 1], r.y, m.y);
        x[4 * q + 2] = fmaf(x[4 * q + 2], r.z, m.z);
        x[4 * q + 3] = fmaf(x[4 * q + 3], r.w, m.w);
    }
}
// D = init + sum_s A[s] * f(B[s]) over 16 K steps, A image [4][64][4] at wa4
template <bool RELU>
__device__ __forceinline__ v16f pr_mm16(const float4 *wa4, const v16f &B, v16f D, int lane)
{
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        const float4 aw = wa4[grp * 64 + lane];
        const float as[4] = {aw.x, aw.y, aw.z, aw.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) D = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], RELU ? fmaxf(B[grp * 4 + s], 0.0f) : B[grp * 4 + s], D, 0, 0, 0);
    }
    return D;
}
__device__ __forceinline__ v16f pr_bias(const float4 *wb4, int off4, int g)
{
    v16f e;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bb = wb4[off4 + g * 4 + q];
        e[4 * q + 0] = bb.x; e[4 * q + 1] = bb.y; e[4 * q + 2] = bb.z; e[4 * q + 3] = bb.w;
    }
    return e;
}
__device__ __forceinline__ v16f pr_zero16()
{
    v16f e;
#pragma unroll
    for (int v = 0; v < 16; ++v) e[v] = 0.0f;
    return e;
}
// P / Q of one tile: 16 steps on 32x32x2 (8 off-centre taps x 4 columns) + the centre tap on 4x4x1; RELU: B = relu(h)
template <bool RELU>
__device__ __forceinline__ void pr_taps(const float4 *wa4, const float4 *wc4, const v16f &h, int lane, int g, v16f &p, v4f &pc)
{
    p = pr_zero16();
    pc = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        const float4 aw = wa4[grp * 64 + lane];
        const float4 ac = wc4[grp * 8 + g * 4 + (lane & 3)];
        const float as[4] = {aw.x, aw.y, aw.z, aw.w};
        const float cs[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float hv = RELU ? fmaxf(h[grp * 4 + s], 0.0f) : h[grp * 4 + s];
            p = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], hv, p, 0, 0, 0);
            pc = __builtin_amdgcn_mfma_f32_4x4x1f32(cs[s], hv, pc, 0, 0, 0);
        }
    }
}
// the strip-local part of the shift-add (nf_wide.hip, phase B): tile k of the strip adds its taps to cp[k - 1 .. k + 1]
struct PrLaneMasks {
    float ml, mr, mg0, mg1, ml0, mr1;
};
template <int TPW, int NW>
__device__ __forceinline__ void pr_shift_add(const v16f &p, const v4f &pc, int k, float (&cp)[TPW][4], const PrLaneMasks &lm, float *exch, int w,
                                             int n, int g)
{
    float rm[4];   // g' = 0: the di = 0 taps (go one row down), g' = 1: the di = 2 taps (go one row up)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rm[j] = fmaf(pr_from_next(p[4 + j]), lm.mr, fmaf(pr_from_prev(p[j]), lm.ml, p[8 + j]));
        cp[k][j] += fmaf(pr_from_next(p[12 + j]), lm.mr1, fmaf(pr_from_prev(p[12 + j]), lm.ml0, pc[j]));
    }
    if (k + 1 < TPW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cp[k + 1][j] = fmaf(rm[j], lm.mg0, cp[k + 1][j]);
    }
    if (k > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cp[k - 1][j] = fmaf(rm[j], lm.mg1, cp[k - 1][j]);
    }
    if (k == 0 && g == 1 && w > 0) *reinterpret_cast<float4 *>(exch + (w * 2 + 1) * 128 + n * 4) = make_float4(rm[0], rm[1], rm[2], rm[3]);
    if (k == TPW - 1 && g == 0 && w < NW - 1) *reinterpret_cast<float4 *>(exch + (w * 2 + 0) * 128 + n * 4) = make_float4(rm[0], rm[1], rm[2], rm[3]);
}
// the rows that crossed a strip boundary (call after the barrier behind the strip loop)
template <int TPW, int NW>
__device__ __forceinline__ void pr_shift_join(float (&cp)[TPW][4], const float *exch, int w, int n, int g)
{
    if (w > 0 && g == 0) {
        const float4 v = *reinterpret_cast<const float4 *>(exch + ((w - 1) * 2 + 0) * 128 + n * 4);
        cp[0][0] += v.x; cp[0][1] += v.y; cp[0][2] += v.z; cp[0][3] += v.w;
    }
    if (w < NW - 1 && g == 1) {
        const float4 v = *reinterpret_cast<const float4 *>(exch + ((w + 1) * 2 + 1) * 128 + n * 4);
        cp[TPW - 1][0] += v.x; cp[TPW - 1][1] += v.y; cp[TPW - 1][2] += v.z; cp[TPW - 1][3] += v.w;
    }
}

// Reductions over the workgroup, in two phases with ONE barrier between them: every *_put parks its partials in a region of its
// own (the patch tiles are dead by then: the caller puts a barrier in front), every *_get adds them up and stores this workgroup's
// slot of the value (and clears the slots no workgroup owns).
// per-channel sums: vals[v] of lane (n, g) is a partial of channel c(v, g).  red: [NW][4][16]
template <int NW>
__device__ __forceinline__ void pr_chan_put(const float (&vals)[16], float *red)
{
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const float r = row_sum16(vals[v]);
        if ((lane & 15) == 0) red[(w * 4 + (lane >> 4)) * 16 + v] = r;
    }
}
template <int NW>
__device__ __forceinline__ void pr_chan_get(const float *red, Acc dst, int nslot, int t)   // t: 0 .. 31
{
    if (t >= 0 && t < 32) {
        const int g = t >> 4, v = t & 15;
        float tot = 0.0f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tot += red[(ww * 4 + 2 * g) * 16 + v] + red[(ww * 4 + 2 * g + 1) * 16 + v];
        float *d = dst.p + (size_t)pr_chan(v, g) * NSLOT;
        d[blockIdx.x] = tot;
        for (int q = blockIdx.x + gridDim.x; q < nslot; q += gridDim.x) d[q] = 0.0f;
    }
}
// N per-thread values -> one slot each.  red: [NW][4][N]
template <int N, int NW>
__device__ __forceinline__ void pr_acc_put(const float (&v)[N], float *red)
{
    const int t = threadIdx.x, row = t >> 4;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float sv = row_sum16(v[k]);
        if ((t & 15) == 0) red[row * N + k] = sv;
    }
}
template <int N, int NW>
__device__ __forceinline__ void pr_acc_get(const float *red, Acc dst, int nslot, int t)   // t: 0 .. N-1
{
    if (t >= 0 && t < N) {
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < NW; ++i) tot += (red[(4 * i) * N + t] + red[(4 * i + 1) * N + t]) + (red[(4 * i + 2) * N + t] + red[(4 * i + 3) * N + t]);
        float *d = dst.p + (size_t)t * NSLOT;
        d[blockIdx.x] = tot;
        for (int q = blockIdx.x + gridDim.x; q < nslot; q += gridDim.x) d[q] = 0.0f;
    }
}
// the 32x32 accumulators of the NW wavefronts -> one slot per value.  red: [NW][16][64]
__device__ __forceinline__ void pr_tile_put(const v16f &D, float *red)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
#pragma unroll
    for (int v = 0; v < 16; ++v) red[(wv * 16 + v) * 64 + ln] = D[v];
}
template <int NW, typename F>
__device__ __forceinline__ void pr_tile_get(const float *red, Acc dst, int nslot, F value)
{
    for (int e = threadIdx.x; e < 1024; e += 64 * NW) {
        const int v = e >> 6, l = e & 63;
        float tot = 0.0f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tot += red[ww * 1024 + e];
        const int idx = value(8 * (v >> 2) + 4 * (l >> 5) + (v & 3), l & 31);
        if (idx >= 0) {
            float *d = dst.p + (size_t)idx * NSLOT;
            d[blockIdx.x] = tot;
            for (int q = blockIdx.x + gridDim.x; q < nslot; q += gridDim.x) d[q] = 0.0f;
        }
    }
}
// the lane's 16 channel values of its pixel -> its row of the wavefront's [pixel][36] tile (channels c(4q .. 4q+3, g) are consecutive)
template <bool RELU>
__device__ __forceinline__ void pr_park(float *tile, const v16f &x, int n, int g)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 o = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        if (RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        *reinterpret_cast<float4 *>(tile + n * PR_RP + 8 * q + 4 * g) = o;
    }
}
// permuted copy of a [mean W][rstd W] pair (or the two BN-backward means) into (g, v) order: dst[kind * 32 + g * 16 + v]
template <bool BN = false>   // BN: src = [mean][rstd] -> [c = -mean rstd][rstd] (pr_bn)
__device__ __forceinline__ void pr_load_bn(float *dst, const float *__restrict__ src, int t0)
{
    const int t = (int)threadIdx.x - t0;
    if (t >= 0 && t < 64) {
        const int ch = pr_chan(t & 15, (t >> 4) & 1);
        dst[t] = (BN && t < 32) ? -src[ch] * src[32 + ch] : src[(t >> 5) * 32 + ch];
    }
}

// The batch moments are finalised by their CONSUMER: every workgroup adds the producer's `nred` slots up itself (2 NW threads per
// channel, fp64, a fixed order: every workgroup gets the same bits), keeps (mean, 1/sqrt(var + eps)) in LDS in (g, v) order, and
// workgroup 0 publishes them for the later stages and moves the running statistics (layers.py:388-393) — what k_bn_fin / k_bnb_fin
// do in a launch of their own (4 launches of ~5.5 us per coupling at 138 patches).
template <int NW>
__device__ __forceinline__ void pr_slot_sums(Acc acc, int nred, double &s, double &q)
{
    constexpr int TPC = 2 * NW;    // threads per channel: 32 channels x TPC = the workgroup
    const int c = threadIdx.x / TPC, sub = threadIdx.x % TPC;
    const float *ps = acc.p + (size_t)c * NSLOT, *pq = acc.p + (size_t)(32 + c) * NSLOT;
    s = 0.0;
    q = 0.0;
    for (int i = sub; i < nred; i += TPC) {
        s += (double)ps[i];
        q += (double)pq[i];
    }
#pragma unroll
    for (int o = TPC / 2; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
}
template <int NW>   // run_mean != null: a training step (the running statistics move); mom != null: an evaluation call ([mean W][var W] left there)
__device__ __forceinline__ void pr_bn_finalize(Acc stats, int nred, double n, float *bnc, float *__restrict__ bn_out, float *__restrict__ run_mean,
                                               float *__restrict__ run_var, float *__restrict__ mom = nullptr)
{
    double sm, sq;
    pr_slot_sums<NW>(stats, nred, sm, sq);
    if (threadIdx.x % (2 * NW) == 0) {
        const int c = threadIdx.x / (2 * NW), gv = ((c >> 2) & 1) * 16 + 4 * (c >> 3) + (c & 3);
        const double m = sm / n;
        double v = sq / n - m * m;
        if (v < 0.0) v = 0.0;
        const float mf = (float)m, rf = (float)(1.0 / sqrt(v + (double)kBnEps));
        bnc[gv] = -mf * rf;
        bnc[32 + gv] = rf;
        if (blockIdx.x == 0) {
            bn_out[c] = mf;
            bn_out[32 + c] = rf;
            if (run_mean) {
                run_mean[c] -= kBnDecay * (run_mean[c] - mf);
                run_var[c] -= kBnDecay * (run_var[c] - (float)v);
            }
            if (mom) {
                mom[c] = mf;
                mom[32 + c] = (float)v;
            }
        }
    }
}
template <int NW>
__device__ __forceinline__ void pr_bnb_finalize(Acc bstats, int nred, double n, float *bnc, float *__restrict__ bb_out)
{
    double sa, sb;
    pr_slot_sums<NW>(bstats, nred, sa, sb);
    if (threadIdx.x % (2 * NW) == 0) {
        const int c = threadIdx.x / (2 * NW), gv = ((c >> 2) & 1) * 16 + 4 * (c >> 3) + (c & 3);
        const float af = (float)(sa / n), bf = (float)(sb / n);
        bnc[gv] = af;
        bnc[32 + gv] = bf;
        if (blockIdx.x == 0) {
            bb_out[c] = af;
            bb_out[32 + c] = bf;
        }
    }
}

struct PrFwdArgs {
    const float *zsrc;       // stage 0: the tensor in front of the folded Conv2d1x1 (MIX) / the coupling input; stages 1, 2: the coupling input
    const float *A;          // MIX: the Conv2d1x1 matrix [4][4]
    float *zmixed;           // MIX: where the coupling input goes
    const float *img;        // this coupling's packed weights (k_pr_pack)
    float *bn1, *bn2;        // [mean W][rstd W] of this step: stage 1 finalises bn1 from stats_in, stage 2 bn2 (and reads bn1)
    Acc stats_in;            // stage 1: the sums stage 0 left; stage 2: the sums stage 1 left
    int nred;                // how many slots they occupy (the producer's grid)
    double n;                // pixels of the (global) minibatch
    float *run_mean, *run_var;   // the running statistics the finalising stage moves
    const float *tail3;      // raw b3 (4), logs (4), rescale
    Acc stats;               // stage 0: sums of h1, h1^2; stage 1: of h2
    float *zout, *u_out;     // stage 2
    Acc ldacc;
    // stage 2 with NEXT: stage 0 of the coupling above (behind one folded Conv2d1x1) in the same launch
    const float *next_A;     // that Conv2d1x1's matrix
    float *next_zmixed;      // where that coupling's input goes
    const float *next_img;   // its packed weights (A1, B1 are used)
    Acc next_stats;          // its sums of h1, h1^2
    // evaluation under batch statistics (nf_bs_wide_run; EVAL 1 = NLL direction, 2 = sampling direction): the moments of the finalised
    // layer for the caller's EMA, the per-patch data-dependent log-det
    float *mom;
    double *ldp;
};
constexpr size_t pr_fwd_lds(int stage, int nw)
{
    return (size_t)(2 * PR_PL + (stage == 0 ? PR_A2 : stage == 1 ? PR_A3 : PR_FWD + PR_A2) + 128 + nw * 256 + nw * 192) * sizeof(float);
}

// NW wavefronts (4 or 8) of 32 / NW rows each: 8 wavefronts put two on every SIMD, so that one's LDS / VALU phases run under the
// other's matrix instructions
// NEXT (stage 2 only): the launch goes on with stage 0 of the coupling above — its Conv2d1x1 applied to the pixels this stage has
// just produced, its l_1 on the tile they are written back to, its batch sums — one launch and one trip of z through HBM less
template <int STAGE, bool MIX, int NW, bool NEXT = false, int EVAL = 0>
__global__ __launch_bounds__(64 * NW) void k_pr_fwd(Geo g, PrFwdArgs a)
{
    static_assert(!NEXT || (STAGE == 2 && !MIX), "only stage 2 continues into the next coupling");
    static_assert(!EVAL || !NEXT, "the evaluator walks the layers one by one");
    constexpr int TPW = 32 / NW, OWN = TPW / 2, NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NIMG = STAGE == 0 ? PR_A2 : STAGE == 1 ? PR_A3 : PR_FWD + PR_A2;   // (stage 2: + A1, B1 of the coupling above)
    float *const z0s = smem;              // [2][PL] zero-bordered planes of the pass-through half
    float *const img = z0s + 2 * PR_PL;
    float *const bnc = img + NIMG;        // [2 layers][mean | rstd][g][v]
    float *const exch = bnc + 128;        // [NW][2][32][4] strip-boundary rows
    float *const red = exch + NW * 256;   // 3 x [NW][4][16]
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, n = lane & 31, gh = lane >> 5, row0 = w * TPW;
    const float4 *const wb4 = reinterpret_cast<const float4 *>(img);
    const int npatch = (int)(g.npix / g.HW);
    // the first patch is requested ahead of the set-up (its latency runs under the image copy and the moment finalisation)
    float4 zr[OWN];
    auto load_patch = [&](int b) {
#pragma unroll
        for (int m = 0; m < OWN; ++m) zr[m] = reinterpret_cast<const float4 *>(a.zsrc)[(int64_t)b * 1024 + (row0 + 2 * m + gh) * 32 + n];
    };
    if ((int)blockIdx.x < npatch) load_patch(blockIdx.x);
    constexpr int NOWN = STAGE == 2 ? PR_FWD : NIMG;
    for (int i = t; i < NOWN / 4; i += NTH) reinterpret_cast<float4 *>(img)[i] = reinterpret_cast<const float4 *>(a.img)[i];
    if (NEXT)
        for (int i = t; i < PR_A2 / 4; i += NTH) reinterpret_cast<float4 *>(img + PR_FWD)[i] = reinterpret_cast<const float4 *>(a.next_img)[i];
    for (int i = t; i < 2 * PR_PL; i += NTH) z0s[i] = 0.0f;
    if (STAGE == 1) pr_bn_finalize<NW>(a.stats_in, a.nred, a.n, bnc, a.bn1, a.run_mean, a.run_var, a.mom);
    if (STAGE == 2) {
        pr_load_bn<true>(bnc, a.bn1, 0);
        pr_bn_finalize<NW>(a.stats_in, a.nred, a.n, bnc + 64, a.bn2, a.run_mean, a.run_var, a.mom);
    }
    PrLaneMasks lm;
    lm.ml = n > 0 ? 1.0f : 0.0f;
    lm.mr = n < 31 ? 1.0f : 0.0f;
    lm.mg0 = gh == 0 ? 1.0f : 0.0f;
    lm.mg1 = gh == 1 ? 1.0f : 0.0f;
    lm.ml0 = lm.ml * lm.mg0;
    lm.mr1 = lm.mr * lm.mg1;
    float mm[16];
    if (MIX) {
#pragma unroll
        for (int i = 0; i < 16; ++i) mm[i] = a.A[i];
    }
    if (NEXT)   // the row sums of the coupling above are added up in LDS patch by patch (its 32 running sums do not fit the registers)
        for (int i = t; i < 2 * NW * 64; i += NTH) red[NW * 64 + i] = 0.0f;
    float s1[16], q1[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) s1[v] = q1[v] = 0.0f;
    float lsum = 0.0f;
    float e3[4] = {1.f, 1.f, 1.f, 1.f}, sc = 0.0f;
    if (STAGE == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) e3[k] = expf(kLogscale * a.tail3[4 + k]);
        sc = a.tail3[8];
    }
    __syncthreads();
    for (int b = blockIdx.x; b < npatch; b += gridDim.x) {
        const int64_t pb = (int64_t)b * 1024;
        float z[OWN][4];
#pragma unroll
        for (int m = 0; m < OWN; ++m) {
            const int r = row0 + 2 * m + gh;
            const float4 u = zr[m];
            if (MIX) {
                z[m][0] = u.x * mm[0] + u.y * mm[4] + u.z * mm[8] + u.w * mm[12];
                z[m][1] = u.x * mm[1] + u.y * mm[5] + u.z * mm[9] + u.w * mm[13];
                z[m][2] = u.x * mm[2] + u.y * mm[6] + u.z * mm[10] + u.w * mm[14];
                z[m][3] = u.x * mm[3] + u.y * mm[7] + u.z * mm[11] + u.w * mm[15];
                reinterpret_cast<float4 *>(a.zmixed)[pb + r * 32 + n] = make_float4(z[m][0], z[m][1], z[m][2], z[m][3]);
            } else {
                z[m][0] = u.x; z[m][1] = u.y; z[m][2] = u.z; z[m][3] = u.w;
            }
        }
        if (b != (int)blockIdx.x) __syncthreads();   // the previous patch's tile has been read
#pragma unroll
        for (int m = 0; m < OWN; ++m) {
            const int r = row0 + 2 * m + gh;
            z0s[(r + 1) * PR_WP + n + 1] = z[m][0];
            z0s[PR_PL + (r + 1) * PR_WP + n + 1] = z[m][1];
        }
        if (b + (int)gridDim.x < npatch) load_patch(b + gridDim.x);   // the next patch of this workgroup: in flight during the strip loop
        __syncthreads();
        float cp[TPW][4];
        if (STAGE == 2) {
#pragma unroll
            for (int k = 0; k < TPW; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) cp[k][j] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            const int r = row0 + k;
            v16f d = pr_l1(wb4, z0s + gh * PR_PL + r * PR_WP + n, lane, gh);
            if (STAGE == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    s1[v] += d[v];
                    q1[v] = fmaf(d[v], d[v], q1[v]);
                }
                continue;
            }
            pr_bn(d, bnc, gh);
            v16f e = pr_mm16<true>(wb4 + PR_A2 / 4, d, pr_bias(wb4, PR_B2 / 4, gh), lane);
            if (STAGE == 1) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    s1[v] += e[v];
                    q1[v] = fmaf(e[v], e[v], q1[v]);
                }
                continue;
            }
            pr_bn(e, bnc + 64, gh);
            v16f p;
            v4f pc;
            pr_taps<true>(wb4 + PR_A3 / 4, wb4 + PR_A3C / 4, e, lane, gh, p, pc);
            pr_shift_add<TPW, NW>(p, pc, k, cp, lm, exch, w, n, gh);
        }
        if (STAGE == 2) {
            __syncthreads();
            pr_shift_join<TPW, NW>(cp, exch, w, n, gh);
            if (NEXT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) mm[i] = a.next_A[i];
            }
            [[maybe_unused]] double lpatch = 0.0;   // EVAL 1: this lane's share of the patch's log-det
#pragma unroll
            for (int m = 0; m < OWN; ++m) {
                const int r = row0 + 2 * m + gh;
                const int bm = (r == 0 ? 1 : 0) | (r == 31 ? 2 : 0) | (n == 0 ? 4 : 0) | (n == 31 ? 8 : 0);
                const float4 eb = *reinterpret_cast<const float4 *>(img + PR_E + 4 * bm);
                float u[4];
                u[0] = pr_half_sums(cp[2 * m][0], cp[2 * m + 1][0]) + eb.x;
                u[1] = pr_half_sums(cp[2 * m][1], cp[2 * m + 1][1]) + eb.y;
                u[2] = pr_half_sums(cp[2 * m][2], cp[2 * m + 1][2]) + eb.z;
                u[3] = pr_half_sums(cp[2 * m][3], cp[2 * m + 1][3]) + eb.w;
                const int64_t p = pb + r * 32 + n;
                if (!EVAL) reinterpret_cast<float4 *>(a.u_out)[p] = make_float4(u[0], u[1], u[2], u[3]);
                const float sh0 = u[0] * e3[0], sh1 = u[1] * e3[1];
                const float ls0 = sc * pr_tanh(u[2] * e3[2]), ls1 = sc * pr_tanh(u[3] * e3[3]);
                const float zo2 = EVAL == 2 ? (z[m][2] - sh0) * pr_exp(-ls0) : fmaf(z[m][2], pr_exp(ls0), sh0);
                const float zo3 = EVAL == 2 ? (z[m][3] - sh1) * pr_exp(-ls1) : fmaf(z[m][3], pr_exp(ls1), sh1);
                if (EVAL == 2 && a.A) {   // sampling direction: the inverse of the Conv2d1x1 in front of the coupling follows it
                    const float q0 = z[m][0], q1 = z[m][1];
                    const float *mi = a.A;
                    reinterpret_cast<float4 *>(a.zout)[p] =
                        make_float4(q0 * mi[0] + q1 * mi[4] + zo2 * mi[8] + zo3 * mi[12], q0 * mi[1] + q1 * mi[5] + zo2 * mi[9] + zo3 * mi[13],
                                    q0 * mi[2] + q1 * mi[6] + zo2 * mi[10] + zo3 * mi[14], q0 * mi[3] + q1 * mi[7] + zo2 * mi[11] + zo3 * mi[15]);
                } else {
                    reinterpret_cast<float4 *>(a.zout)[p] = make_float4(z[m][0], z[m][1], zo2, zo3);
                }
                if (EVAL == 1) lpatch += (double)ls0 + (double)ls1;
                else lsum += ls0 + ls1;
                if (NEXT) {   // the coupling above: its input, and the pass-through half of it into the tile (every strip loop is over)
                    const float u0 = z[m][0], u1 = z[m][1];
                    const float v0 = u0 * mm[0] + u1 * mm[4] + zo2 * mm[8] + zo3 * mm[12], v1 = u0 * mm[1] + u1 * mm[5] + zo2 * mm[9] + zo3 * mm[13];
                    reinterpret_cast<float4 *>(a.next_zmixed)[p] =
                        make_float4(v0, v1, u0 * mm[2] + u1 * mm[6] + zo2 * mm[10] + zo3 * mm[14], u0 * mm[3] + u1 * mm[7] + zo2 * mm[11] + zo3 * mm[15]);
                    z0s[(r + 1) * PR_WP + n + 1] = v0;
                    z0s[PR_PL + (r + 1) * PR_WP + n + 1] = v1;
                }
            }
            if (EVAL == 1) {   // the patch's log-det share of this coupling: one fp64 sum over the workgroup, in a fixed order
                double *const dred = reinterpret_cast<double *>(red);
                const double ws = wsum(lpatch);
                __syncthreads();
                if (lane == 0) dred[w] = ws;
                __syncthreads();
                if (t == 0) {
                    double tot = 0.0;
#pragma unroll
                    for (int i = 0; i < NW; ++i) tot += dred[i];
                    a.ldp[b] += tot;
                }
            }
            if (NEXT) {
                __syncthreads();
                const float4 *const wn4 = reinterpret_cast<const float4 *>(img + PR_FWD);
                float sn[16], qn[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) sn[v] = qn[v] = 0.0f;
#pragma unroll
                for (int k = 0; k < TPW; ++k) {
                    const v16f d = pr_l1(wn4, z0s + gh * PR_PL + (row0 + k) * PR_WP + n, lane, gh);
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        sn[v] += d[v];
                        qn[v] = fmaf(d[v], d[v], qn[v]);
                    }
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) {   // (the entries a 16-lane row adds to are its own)
                    const float rs = row_sum16(sn[v]), rq = row_sum16(qn[v]);
                    if ((lane & 15) == 0) {
                        red[NW * 64 + (w * 4 + (lane >> 4)) * 16 + v] += rs;
                        red[NW * 128 + (w * 4 + (lane >> 4)) * 16 + v] += rq;
                    }
                }
            }
        }
    }
    if (STAGE < 2) {   // (red is a region of its own: no barrier in front)
        pr_chan_put<NW>(s1, red);
        pr_chan_put<NW>(q1, red + NW * 64);
        __syncthreads();
        pr_chan_get<NW>(red, a.stats, g.nslot, t);
        pr_chan_get<NW>(red + NW * 64, a.stats + 32, g.nslot, t - 64);
    } else if (!EVAL) {
        const float lv[1] = {lsum};
        pr_acc_put<1, NW>(lv, red);
        __syncthreads();
        pr_acc_get<1, NW>(red, a.ldacc, g.nslot, t);
        if (NEXT) {
            pr_chan_get<NW>(red + NW * 64, a.next_stats, g.nslot, t - 64);
            pr_chan_get<NW>(red + NW * 128, a.next_stats + 32, g.nslot, t - 128);
        }
    }
}

struct PrBwdArgs {
    const float *zin;          // the coupling input
    const float *img, *bn1, *bn2;
    float *bb2, *bb1;          // the two means of each BN backward: B finalises bb2, C bb1 (and reads bb2)
    const float *tail3;        // raw b3 (4), logs (4), rescale
    // stage A: the elementwise head
    const float *u, *zlat, *dz;   // zlat != null: d loss / d latent = latent / B;  else d loss / d (coupling output) in dz
    float *dz2;                // A writes d loss / d (z0, z1 as they leave, z2, z3 as they enter); C reads it
    float *gu;                 // A writes, B and C read: d loss / d (l_last output)
    float invB;
    Acc bstats_in;             // B: the sums A left (BN2's backward); C: the sums B left
    int nred;
    double n;
    Acc bstats;                // A: the two sums of BN2's backward; B: of BN1's
    Acc G;                     // parameter-gradient slots
    int off_w1, off_b1, off_w2, off_w3;
    // stage C
    float *dz_out;
    const float *zmix_in, *A;
    Acc dA;
};
constexpr int pr_bwd_nb(int stage) { return stage == 0 ? PR_A2T - PR_A3T : stage == 1 ? PR_A1Q - PR_A3T : PR_SIZE - PR_A3T; }
constexpr int pr_bwd_tiles(int stage) { return stage == 1 ? 2 : 1; }
constexpr size_t pr_bwd_lds(int stage, int nw)
{
    return (size_t)(2 * PR_PL + 4 * PR_PL + PR_A3 + pr_bwd_nb(stage) + 256 + nw * pr_bwd_tiles(stage) * 32 * PR_RP + (nw * 256 > 1024 ? nw * 256 : 1024)) *
           sizeof(float);
}

template <int STAGE, bool MIX, int NW, int GRAD = 2>
__global__ __launch_bounds__(64 * NW) void k_pr_bwd(Geo g, PrBwdArgs a)
{
#include "nf_train_pr_bwd.inc"
}

// Stage C of a coupling and stage A of the coupling below it in ONE launch: A of a patch needs nothing but C of the same patch (the
// d loss / d z the same lanes have just stored), so a workgroup walks its patches through C, then through A — one launch and one
// wait for the slowest workgroup less per coupling.
template <bool MIXC, int NW, int GRAD_A = 2, int GRAD_C = 2>   // GRAD_x: the stage with (2) or without (0) its filter gradient
__global__ __launch_bounds__(64 * NW) void k_pr_bwd_CA(Geo g, PrBwdArgs a, PrBwdArgs below)
{
    {
        constexpr int STAGE = 2, GRAD = GRAD_C;
        constexpr bool MIX = MIXC;
#include "nf_train_pr_bwd.inc"
    }
    __syncthreads();
    {
        constexpr int STAGE = 0, GRAD = GRAD_A;
        constexpr bool MIX = false;
        const PrBwdArgs &a = below;
#include "nf_train_pr_bwd.inc"
    }
}

// The two filter gradients a k_pr_bwd_CA launch leaves out (GRAD 0), on the side stream: d l_1/W of the coupling (stage C's product) and
// d l_last/W of the coupling below it (stage A's) — one launch, in the CUs a small minibatch leaves idle.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_pr_bwd_grads(Geo g, PrBwdArgs a, PrBwdArgs below)
{
    {
        constexpr int STAGE = 2, GRAD = 1;
        constexpr bool MIX = false;
#include "nf_train_pr_bwd.inc"
    }
    __syncthreads();
    {
        constexpr int STAGE = 0, GRAD = 1;
        constexpr bool MIX = false;
        const PrBwdArgs &a = below;
#include "nf_train_pr_bwd.inc"
    }
}
