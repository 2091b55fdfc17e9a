// nf_train_wide.h — part of nf_train.hip (included inside its anonymous namespace; not a standalone header).
// The trainer's stage kernels for coupling widths 16 and 32 on v_mfma_f32_32x32x2_f32.
// ---------------------------------------------------------------------------------------------
// widths 16 and 32 (32 = the paper-scale coupling CNN, job_noise_flow.sh:19): the filter gradients on the matrix cores
// (written for width 32 — the comments below count in its numbers; templates on the width, see RowTile)
// ---------------------------------------------------------------------------------------------
// A filter gradient is a GEMM whose K axis is the PIXELS of the minibatch: dW[i][j] = sum_p A[p][i] G[p][j].  One
// v_mfma_f32_32x32x2_f32 (exact fp32) takes two pixels: lane (i = lane & 31, k = lane >> 5) supplies A[p_k][i] and
// G[p_k][i'] — with the pixel-major [p][32] tensors that is one coalesced 128-byte row per lane half and operand — and the
// 32x32 result stays in 16 accumulator registers for the whole pixel loop.  The per-thread-per-pixel kernels above hold
// W (or 4 W) accumulators per thread and walk 128-byte rows with a 128-byte lane stride: 3.6 ms per coupling for
// d l_last/W at 1 024 patches, against ~0.1 ms here.
typedef float v16f __attribute__((ext_vector_type(16)));

// the 32x32 accumulators of the 4 wavefronts of a workgroup -> one slot per value;  value(row, col) = index of D[row][col]
// relative to `dst`, or -1 for an unused column.  red: [4][16][64] floats.
template <typename F>
__device__ __forceinline__ void mfma_tile_to_slots(const v16f &D, float *red, Acc dst, int nslot, F value)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) red[(wv * 16 + v) * 64 + ln] = D[v];
    __syncthreads();
    for (int e = t; e < 1024; e += 256) {
        const int v = e >> 6, l = e & 63;
        const float tot = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
        const int idx = value(8 * (v >> 2) + 4 * (l >> 5) + (v & 3), l & 31);
        if (idx >= 0) {
            float *d = dst.p + (size_t)idx * NSLOT;
            d[blockIdx.x] = tot;
            for (int q = blockIdx.x + gridDim.x; q < nslot; q += gridDim.x) d[q] = 0.0f;
        }
    }
}

// Operand staging: 32 consecutive rows of a [rows][32] tensor = 4 KB of consecutive memory.  The wavefront fetches them with
// four 16-byte loads per lane (1 KB of whole cache lines per instruction; one dword per lane and MFMA step left the kernels
// latency-bound at ~1 TB/s), parks them in its own LDS rows of 36 floats, and each MFMA step reads its two pixels back as
// one ds_read_b32 per operand.  LDS instructions of one wavefront execute in order, so no barrier — only a compiler fence.
// The same kernels serve width 16 ([rows][16] tensors, 2 KB per 32 rows, LDS rows of 20 floats): the channel GEMMs then use
// half of the instruction's M or K extent (A rows / K steps beyond the width are absent), the pixel-K GEMMs ignore the rows
// and columns beyond it.  W / 2 = the K steps of a channel GEMM = the values a lane holds of its pixel's row.
struct RowTile {
    float4 v[4];
};
template <int W>
__device__ __forceinline__ void rows_fetch(RowTile &r, const float *__restrict__ src, int64_t row0, int64_t row_end)
{
    static_assert(W == 16 || W == 32, "matrix-core trainer kernels: widths 16 and 32");
    constexpr int RQ = W / 4;        // 16-byte pieces per row
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int m = 0; m < W / 8; ++m) {
        const int64_t row = row0 + (ln + 64 * m) / RQ;
        r.v[m] = row < row_end ? reinterpret_cast<const float4 *>(src)[row * RQ + ln % RQ] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// plain copy / BN + ReLU of the lane's 4 channels (4 (ln % (W / 4)) ..) on the way into LDS; rows past the end stay zero
template <int W, bool BNRELU>
__device__ __forceinline__ void rows_park(const RowTile &r, float *lds, const float (&m)[4], const float (&rs)[4], int64_t row0,
                                          int64_t row_end)
{
    constexpr int RQ = W / 4, RP = W + 4;
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < W / 8; ++k) {
        const int lr = (ln + 64 * k) / RQ;
        float4 v = r.v[k];
        if (BNRELU) {
            const bool in = row0 + lr < row_end;
            v.x = in ? fmaxf((v.x - m[0]) * rs[0], 0.0f) : 0.0f;
            v.y = in ? fmaxf((v.y - m[1]) * rs[1], 0.0f) : 0.0f;
            v.z = in ? fmaxf((v.z - m[2]) * rs[2], 0.0f) : 0.0f;
            v.w = in ? fmaxf((v.w - m[3]) * rs[3], 0.0f) : 0.0f;
        }
        *reinterpret_cast<float4 *>(lds + lr * RP + 4 * (ln % RQ)) = v;
    }
}

// d l_2/W[i][j] = sum_p relu(bn1(h1))[p][i] * g_h2[p][j]
template <int W>
__global__ __launch_bounds__(256) void k_w2_grad_mfma(Geo g, const float *__restrict__ h1, const float *__restrict__ bn1,
                                                      const float *__restrict__ t1, int off_w2, Acc G)
{
    constexpr int RP = W + 4;
    __shared__ float stage[4][2][32 * 36];
    float *red = &stage[0][0][0];   // [4][16][64], used once the pixel loop is over (mfma_tile_to_slots starts with a barrier)
    static_assert(sizeof(stage) >= 4 * 16 * 64 * sizeof(float), "the reduction buffer must fit the staging area");
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    float m[4], rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = bn1[4 * (ln % (W / 4)) + k];
        rs[k] = bn1[W + 4 * (ln % (W / 4)) + k];
    }
    v16f D;
#pragma unroll
    for (int v = 0; v < 16; ++v) D[v] = 0.0f;
    float *sa = stage[wv][0], *sb = stage[wv][1];
    const int64_t ntiles = (g.npix + 31) >> 5, stride = (int64_t)gridDim.x * 4;
    int64_t T = (int64_t)blockIdx.x * 4 + wv;
    RowTile ra, rb;
    if (T < ntiles) {
        rows_fetch<W>(ra, h1, T * 32, g.npix);
        rows_fetch<W>(rb, t1, T * 32, g.npix);
    }
    for (; T < ntiles; T += stride) {
        wave_lds_fence();                                  // the previous tile's reads are issued
        rows_park<W, true>(ra, sa, m, rs, T * 32, g.npix);
        rows_park<W, false>(rb, sb, m, rs, T * 32, g.npix);
        if (T + stride < ntiles) {                         // next tile in flight during the MFMAs
            rows_fetch<W>(ra, h1, (T + stride) * 32, g.npix);
            rows_fetch<W>(rb, t1, (T + stride) * 32, g.npix);
        }
        wave_lds_fence();
        const int cw = col < W ? col : 0;   // rows / columns beyond the width are not used
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[(2 * s2 + half) * RP + cw], sb[(2 * s2 + half) * RP + cw], D, 0, 0, 0);
    }
    mfma_tile_to_slots(D, red, G + off_w2, g.nslot, [](int i, int j) { return i < W && j < W ? i * W + j : -1; });
}

// d l_last/W[tap][i][q] = sum_p' relu(bn2(h2))[p'][i] * gu[p' - tap][q]  (p' = the pixel the tap reads, inside the patch;
// gu from a zero-bordered tile of the patch), columns (tap, q): taps 0..7 in one 32-column tile, tap 8 in a second.
// The indicator channel's gradient, sum over the pixels whose tap falls on the padding ring, is the column sum of ALL of gu
// (the centre tap's column sum) minus the column sum over the taps that land inside — both fall out of the B operands.
template <int W>
__global__ __launch_bounds__(256) void k_w3_grad_mfma(Geo g, const float *__restrict__ h2, const float *__restrict__ bn2,
                                                      const float *__restrict__ gu, int off_w3, Acc G, int S)
{
    constexpr int RP = W + 4;
    extern __shared__ float smem[];   // gu tile [(H+2)(W+2)][4], then the tile index of every pixel of a patch (int)
    __shared__ float stage[4][32 * 36];
    float *red = &stage[0][0];        // [4][16][64], used once the pixel loop is over
    static_assert(sizeof(stage) >= 4 * 16 * 64 * sizeof(float), "the reduction buffer must fit the staging area");
    __shared__ float cs[2][4][64], cst[40];
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    const int Wp = g.W + 2, tile_px = (g.H + 2) * Wp;
    float m[4], rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = bn2[4 * (ln % (W / 4)) + k];
        rs[k] = bn2[W + 4 * (ln % (W / 4)) + k];
    }
    const int tap0 = col >> 2, q = col & 3, cw = col < W ? col : 0;
    const int d0 = (tap0 / 3 - 1) * Wp + (tap0 % 3 - 1), d1 = Wp + 1;   // tile offset of the pixel tap (di, dj) comes from
    v16f D0, D1;
#pragma unroll
    for (int v = 0; v < 16; ++v) D0[v] = D1[v] = 0.0f;
    float S0 = 0.0f, S1 = 0.0f;
    float *sa = stage[wv];
    int *lut = reinterpret_cast<int *>(smem + tile_px * 4);   // pixel -> its position in the bordered tile (no division per step)
    for (int i = t; i < tile_px * 4; i += 256) smem[i] = 0.0f;
    for (int px = t; px < g.HW; px += 256) {
        const int r = px / g.W;
        lut[px] = (r + 1) * Wp + (px - r * g.W) + 1;
    }
    const int npatch = (int)(g.npix / g.HW), ntiles = (g.HW + 31) >> 5;
    for (int unit = blockIdx.x; unit < npatch * S; unit += gridDim.x) {   // S workgroups share a patch's tiles
        const int b = unit / S, T0 = wv + 4 * (unit - b * S);
        const float *hb = h2 + (int64_t)b * g.HW * W;
        RowTile ra;
        if (T0 < ntiles) rows_fetch<W>(ra, hb, T0 * 32, g.HW);
        __syncthreads();   // the border is zero / the previous patch is done with
        for (int px = t; px < g.HW; px += 256)
            reinterpret_cast<float4 *>(smem)[lut[px]] = reinterpret_cast<const float4 *>(gu)[(int64_t)b * g.HW + px];
        __syncthreads();
        for (int T = T0; T < ntiles; T += 4 * S) {
            wave_lds_fence();
            rows_park<W, true>(ra, sa, m, rs, T * 32, g.HW);
            if (T + 4 * S < ntiles) rows_fetch<W>(ra, hb, (T + 4 * S) * 32, g.HW);
            wave_lds_fence();
#pragma unroll 4
            for (int s2 = 0; s2 < 16; ++s2) {
                const int pp = T * 32 + 2 * s2 + half;
                float b0 = 0.0f, b1 = 0.0f;
                if (pp < g.HW) {
                    const int tp = lut[pp];
                    b0 = smem[(tp - d0) * 4 + q];
                    if (col < 4) b1 = smem[(tp - d1) * 4 + q];
                }
                const float a = sa[(2 * s2 + half) * RP + cw];
                S0 += b0;
                S1 += b1;
                D0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, D0, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, D1, 0, 0, 0);
            }
        }
    }
    const Acc dst = G + off_w3;
    mfma_tile_to_slots(D0, red, dst, g.nslot, [](int i, int c) { return i < W ? (c >> 2) * (W + 1) * 4 + i * 4 + (c & 3) : -1; });
    mfma_tile_to_slots(D1, red, dst, g.nslot, [](int i, int c) { return i < W && c < 4 ? 8 * (W + 1) * 4 + i * 4 + c : -1; });
    cs[0][wv][ln] = S0;
    cs[1][wv][ln] = S1;
    __syncthreads();
    if (t < 36) {   // column sums over the 4 wavefronts and both lane halves: t = tap * 4 + q
        const int k = t < 32 ? 0 : 1, c = t < 32 ? t : t - 32;
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) tot += cs[k][w][c] + cs[k][w][c + 32];
        cst[t] = tot;
    }
    __syncthreads();
    if (t < 36) {
        float *d = dst.p + (size_t)((t >> 2) * (W + 1) * 4 + W * 4 + (t & 3)) * NSLOT;
        d[blockIdx.x] = cst[16 + (t & 3)] - cst[t];
        for (int k = blockIdx.x + gridDim.x; k < g.nslot; k += gridDim.x) d[k] = 0.0f;
    }
}

// d l_1/W[tap][c][j] = sum_p z[p + tap][c] * g_h1[p][j]  (c: the two pass-through channels; z from a zero-bordered tile of the
// patch): rows (tap, c) = 18 of the 32, columns j, K = the pixels.
template <int W>
__global__ __launch_bounds__(256) void k_w1_grad_mfma(Geo g, const float *__restrict__ zin, const float *__restrict__ t2, int off_w1,
                                                      Acc G, int S)
{
    constexpr int RP = W + 4;
    extern __shared__ float smem[];   // z tile [(H+2)(W+2)][2], then the tile index of every pixel of a patch (int)
    __shared__ float stage[4][32 * 36];
    float *red = &stage[0][0];        // [4][16][64], used once the pixel loop is over
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    const int Wp = g.W + 2, tile_px = (g.H + 2) * Wp;
    const int tap = col >> 1, ch = col & 1;               // row of the result = tap * 2 + ch, valid below 18
    const int da = (tap / 3 - 1) * Wp + (tap % 3 - 1);
    const float unused_m[4] = {0.f, 0.f, 0.f, 0.f};
    v16f D;
#pragma unroll
    for (int v = 0; v < 16; ++v) D[v] = 0.0f;
    float *sb = stage[wv];
    int *lut = reinterpret_cast<int *>(smem + tile_px * 2);
    for (int i = t; i < tile_px * 2; i += 256) smem[i] = 0.0f;
    for (int px = t; px < g.HW; px += 256) {
        const int r = px / g.W;
        lut[px] = (r + 1) * Wp + (px - r * g.W) + 1;
    }
    const int npatch = (int)(g.npix / g.HW), ntiles = (g.HW + 31) >> 5;
    for (int unit = blockIdx.x; unit < npatch * S; unit += gridDim.x) {   // S workgroups share a patch's tiles
        const int b = unit / S, T0 = wv + 4 * (unit - b * S);
        const float *tb = t2 + (int64_t)b * g.HW * W;
        RowTile rb;
        if (T0 < ntiles) rows_fetch<W>(rb, tb, T0 * 32, g.HW);
        __syncthreads();
        for (int px = t; px < g.HW; px += 256)
            reinterpret_cast<float2 *>(smem)[lut[px]] = *reinterpret_cast<const float2 *>(zin + ((int64_t)b * g.HW + px) * 4);
        __syncthreads();
        for (int T = T0; T < ntiles; T += 4 * S) {
            wave_lds_fence();
            rows_park<W, false>(rb, sb, unused_m, unused_m, T * 32, g.HW);
            if (T + 4 * S < ntiles) rows_fetch<W>(rb, tb, (T + 4 * S) * 32, g.HW);
            wave_lds_fence();
#pragma unroll 4
            for (int s2 = 0; s2 < 16; ++s2) {
                const int pp = T * 32 + 2 * s2 + half;
                float a = 0.0f;
                if (pp < g.HW && col < 18) a = smem[(lut[pp] + da) * 2 + ch];
                D = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sb[(2 * s2 + half) * RP + (col < W ? col : 0)], D, 0, 0, 0);
            }
        }
    }
    // k_w1_grad's layout: [tap][c][j]
    mfma_tile_to_slots(D, red, G + off_w1, g.nslot, [](int i, int j) { return i < 18 && j < W ? (i >> 1) * 2 * W + (i & 1) * W + j : -1; });
}

// ---- widths 16 / 32: l_last forward on the matrix cores ----------------------------------------------------------------------
// u[p][q] = b[q] + sum_tap sum_i relu(bn2(h2))[p + tap][i] W3[tap][i][q] is evaluated transposed, as in the evaluation kernel
// (nf_wide.hip): P[p][(tap, q)] = sum_i A2[p][i] W3[tap][i][q] is a GEMM over the 32 channels with 36 output rows (taps 0..7 in
// one 32-row tile, tap 8 in a second), pixels on N, and u is the shift-add u[p][q] = sum_tap P[p + tap][(tap, q)].  One
// workgroup owns a band of rows of one patch: it computes P for the band and a one-row halo on each side (10 rows for 8 at
// 32x32: 25 % recomputed) into LDS, then one thread per pixel gathers its 9 taps and finishes the affine transform.
// The layer kernel does the same 1 188 MAC per pixel on the vector unit, re-normalising each of the 9 neighbours it reads.
template <int W>
__global__ __launch_bounds__(256) void k_c3_fwd_mfma(Geo g, const float *__restrict__ zin, const float *__restrict__ h2,
                                                       const float *__restrict__ bn2, const float *__restrict__ Pw, int off_w3,
                                                       float *__restrict__ zout, Acc ldacc, float *__restrict__ u_out, int BR)
{
    constexpr int PS = 36, RP = W + 4, HK = W / 2;
    extern __shared__ float smem[];   // P [pixels of the band + halo][36]
    __shared__ float stage[4][32 * 36];
    const float *W3 = Pw + off_w3, *b3 = W3 + 36 * (W + 1), *logs = b3 + 4;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    float a0[HK], a1[HK];             // A[(tap, q) = col][i = 2 s + half]
#pragma unroll
    for (int k = 0; k < HK; ++k) {
        const int i = 2 * k + half;
        a0[k] = W3[(col >> 2) * (W + 1) * 4 + i * 4 + (col & 3)];
        a1[k] = col < 4 ? W3[8 * (W + 1) * 4 + i * 4 + col] : 0.0f;
    }
    float m[4], rs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = bn2[4 * (ln % (W / 4)) + k];
        rs[k] = bn2[W + 4 * (ln % (W / 4)) + k];
    }
    const float sc = logs[4];
    const float e30 = expf(kLogscale * logs[0]), e31 = expf(kLogscale * logs[1]), e32 = expf(kLogscale * logs[2]),
                e33 = expf(kLogscale * logs[3]);
    float *sa = stage[wv];
    const int npatch = (int)(g.npix / g.HW), nbands = (g.H + BR - 1) / BR, units = npatch * nbands;
    float l = 0.0f;
    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int b = unit / nbands, r0 = (unit - b * nbands) * BR, r1 = min(r0 + BR, g.H);
        const int rlo = max(r0 - 1, 0), rhi = min(r1 + 1, g.H), ntiles = ((rhi - rlo) * g.W + 31) >> 5;
        const int64_t base = (int64_t)b * g.HW + rlo * g.W, end = (int64_t)b * g.HW + rhi * g.W;
        RowTile ra;
        if (wv < ntiles) rows_fetch<W>(ra, h2, base + wv * 32, end);
        __syncthreads();              // the previous band's gather is over
        for (int T = wv; T < ntiles; T += 4) {
            wave_lds_fence();
            rows_park<W, true>(ra, sa, m, rs, base + T * 32, end);
            if (T + 4 < ntiles) rows_fetch<W>(ra, h2, base + (T + 4) * 32, end);
            wave_lds_fence();
            v16f D0, D1;
#pragma unroll
            for (int v = 0; v < 16; ++v) D0[v] = D1[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < HK; ++k) {
                const float bv = sa[col * RP + 2 * k + half];
                D0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[k], bv, D0, 0, 0, 0);
                D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[k], bv, D1, 0, 0, 0);
            }
            float *pp = smem + (T * 32 + col) * PS;
#pragma unroll
            for (int v = 0; v < 16; v += 4)
                *reinterpret_cast<float4 *>(pp + 2 * v + 4 * half) = make_float4(D0[v], D0[v + 1], D0[v + 2], D0[v + 3]);
            if (half == 0) *reinterpret_cast<float4 *>(pp + 32) = make_float4(D1[0], D1[1], D1[2], D1[3]);
        }
        __syncthreads();
        for (int px = t; px < (r1 - r0) * g.W; px += 256) {
            const int rr0 = px / g.W, r = r0 + rr0, c = px - rr0 * g.W;
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = b3[k];
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int rr = r + di - 1;
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int cc = c + dj - 1, tap = di * 3 + dj;
                    if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {   // on the padding ring: zeros + indicator 1
                        const float *w = W3 + tap * (W + 1) * 4 + W * 4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) u[k] += w[k];
                    } else {
                        const float4 pv = *reinterpret_cast<const float4 *>(smem + ((rr - rlo) * g.W + cc) * PS + tap * 4);
                        u[0] += pv.x; u[1] += pv.y; u[2] += pv.z; u[3] += pv.w;
                    }
                }
            }
            const int64_t p = (int64_t)b * g.HW + r * g.W + c;
            if (u_out) reinterpret_cast<float4 *>(u_out)[p] = make_float4(u[0], u[1], u[2], u[3]);
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

// ---- widths 16 / 32: the 1x1 layer l_2, forward and transposed, as pixel GEMMs -------------------------------------------------
// 32 consecutive pixels of the batch on the N axis of v_mfma_f32_32x32x2_f32 (lane & 31 = the pixel, both lane halves),
// the 32 output channels on M, the 32 input channels on K in the order k(step s, lane half h) = 16 h + s: lane (p, h) then
// feeds the 16 consecutive floats [16 h, 16 h + 16) of its pixel's row — four 16-byte loads per tensor — and owns, in the
// result, the 16 output channels c(v, h) = 8 (v >> 2) + 4 h + (v & 3), four 16-byte stores.  The layer kernels walk the same
// rows one pixel per thread with W accumulators each (0.6 ms per coupling for the backward stage at 1 024 patches).
// Per-channel sums: 16 registers per lane, added up over the 32 lanes of a half and the 4 wavefronts once, at the end.

// red: [4][64][16];  vals[k] of lane (col, half) belongs to channel chan(k, half);  -> dst + chan
template <int N, typename F>
__device__ __forceinline__ void lane_sums_to_slots(const float (&vals)[N], float *red, Acc dst, int nslot, F chan)
{
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) red[(wv * 64 + ln) * N + k] = vals[k];
    __syncthreads();
    if (t < 2 * N) {   // t = (half, k)
        const int half = t / N, k = t % N;
        float tot = 0.0f;
        for (int w = 0; w < 4; ++w)
            for (int c = 0; c < 32; ++c) tot += red[(w * 64 + half * 32 + c) * N + k];
        float *d = dst.p + (size_t)chan(k, half) * NSLOT;
        d[blockIdx.x] = tot;
        for (int q = blockIdx.x + gridDim.x; q < nslot; q += gridDim.x) d[q] = 0.0f;
    }
}

__device__ __forceinline__ int mfma_row(int v, int half) { return 8 * (v >> 2) + 4 * half + (v & 3); }

// the reverse of rows_fetch / rows_park: 32 rows parked in LDS (stride W + 4) -> consecutive memory
template <int W>
__device__ __forceinline__ void rows_flush(const float *lds, float *__restrict__ dst, int64_t row0, int64_t row_end)
{
    constexpr int RQ = W / 4, RP = W + 4;
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int m = 0; m < W / 8; ++m) {
        const int lr = (ln + 64 * m) / RQ;
        if (row0 + lr < row_end)
            reinterpret_cast<float4 *>(dst)[(row0 + lr) * RQ + ln % RQ] = *reinterpret_cast<const float4 *>(lds + lr * RP + 4 * (ln % RQ));
    }
}

// BN1 + ReLU + l_2 + bias; statistics of the result (k_c2_fwd at widths 16 / 32); tensor tiles staged through LDS
template <int W>
__global__ __launch_bounds__(256) void k_c2_fwd_mfma(Geo g, const float *__restrict__ h1, Acc stats1, double n, float *__restrict__ P,
                                                       int off_m1, float *__restrict__ bn1_out, int off_w2, float *__restrict__ h2,
                                                       Acc stats, const float *__restrict__ Pw, bool fin)
{
    constexpr int RP = W + 4, HK = W / 2;
    __shared__ float bn1[2 * W];
    __shared__ float stage[4][32 * 36];
    __shared__ float red[4 * 64 * 16];
    bn_from_slots<W>(stats1, g.nslot, n, bn1, P, off_m1, off_m1 + W, bn1_out, fin);
    const float *W2 = Pw + off_w2, *b2 = W2 + W * W;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    float a[HK];   // A[out = col][k(s, half)] = W2[HK half + s][col]; output rows beyond the width do not exist
#pragma unroll
    for (int k = 0; k < HK; ++k) a[k] = col < W ? W2[(HK * half + k) * W + col] : 0.0f;
    float m1[4], r1[4], bo[HK], s2[HK], q2[HK];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // BN1 + ReLU is applied on the way into LDS: the lane's 4 channels of the 16-byte pieces it moves
        m1[k] = bn1[4 * (ln % (W / 4)) + k];
        r1[k] = bn1[W + 4 * (ln % (W / 4)) + k];
    }
#pragma unroll
    for (int k = 0; k < HK; ++k) {
        bo[k] = b2[mfma_row(k, half)];
        s2[k] = q2[k] = 0.0f;
    }
    float *sx = stage[wv];
    const int64_t ntiles = (g.npix + 31) >> 5, stride = (int64_t)gridDim.x * 4;
    int64_t T = (int64_t)blockIdx.x * 4 + wv;
    RowTile rx;
    if (T < ntiles) rows_fetch<W>(rx, h1, T * 32, g.npix);
    for (; T < ntiles; T += stride) {
        const bool in = T * 32 + col < g.npix;
        wave_lds_fence();             // the previous tile's flush has been issued
        rows_park<W, true>(rx, sx, m1, r1, T * 32, g.npix);
        if (T + stride < ntiles) rows_fetch<W>(rx, h1, (T + stride) * 32, g.npix);
        wave_lds_fence();
        float x[HK];
#pragma unroll
        for (int k = 0; k < HK; k += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(sx + col * RP + HK * half + k);
            x[k] = v.x; x[k + 1] = v.y; x[k + 2] = v.z; x[k + 3] = v.w;
        }
        v16f D;
#pragma unroll
        for (int v = 0; v < 16; ++v) D[v] = 0.0f;
#pragma unroll
        for (int k = 0; k < HK; ++k) D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], x[k], D, 0, 0, 0);
        wave_lds_fence();             // every lane has read its inputs: the tile takes the result
#pragma unroll
        for (int v = 0; v < HK; v += 4) {
            const float4 o = make_float4(D[v] + bo[v], D[v + 1] + bo[v + 1], D[v + 2] + bo[v + 2], D[v + 3] + bo[v + 3]);
            *reinterpret_cast<float4 *>(sx + col * RP + mfma_row(v, half)) = o;
            if (in) {
                s2[v] += o.x; s2[v + 1] += o.y; s2[v + 2] += o.z; s2[v + 3] += o.w;
                q2[v] = fmaf(o.x, o.x, q2[v]); q2[v + 1] = fmaf(o.y, o.y, q2[v + 1]);
                q2[v + 2] = fmaf(o.z, o.z, q2[v + 2]); q2[v + 3] = fmaf(o.w, o.w, q2[v + 3]);
            }
        }
        wave_lds_fence();
        rows_flush<W>(sx, h2, T * 32, g.npix);
    }
    lane_sums_to_slots(s2, red, stats, g.nslot, [](int k, int h) { return mfma_row(k, h); });
    lane_sums_to_slots(q2, red, stats, g.nslot, [](int k, int h) { return W + mfma_row(k, h); });
}

// BN2 backward -> g_h2 (t1, in place), d l_2/b; transposed l_2 + ReLU mask -> d loss / d xhat1 (t2) and its two batch sums
// (k_c2_bwd at widths 16 / 32).  Five passes over [pixels][W] tensors: all of them through wavefront-private LDS tiles, so that
// every global access is a whole 4 KB tile in 16-byte pieces.
// WGRAD: d l_2/W = A1^T g_h2 is accumulated here as well — both operands are in the staged tiles at that point (K = the 32
// pixels of the tile) — instead of by k_w2_grad_mfma from a second pass over h1 and a stored g_h2; t1 is then read only.
template <int W, bool WGRAD>
__global__ __launch_bounds__(256) void k_c2_bwd_mfma(Geo g, const float *__restrict__ h1, const float *__restrict__ bn1,
                                                       const float *__restrict__ h2, const float *__restrict__ bn2, Acc bstats2,
                                                       double n, const float *__restrict__ P, int off_w2, float *__restrict__ t1,
                                                       float *__restrict__ t2, Acc bstats, Acc G, const float *__restrict__ pre)
{
    constexpr int RP = W + 4, HK = W / 2;
    __shared__ float bb2[2 * W], sbn2[2 * W], sbn1[2 * W];
    __shared__ float stage[4][3][32 * 36];
    float *red = &stage[0][0][0];     // [4][64][16], used once the pixel loop is over (lane_sums_to_slots starts with a barrier)
    static_assert(sizeof(stage) >= 4 * 64 * 16 * sizeof(float), "the reduction buffer must fit the staging area");
    if (threadIdx.x < 2 * W) {
        sbn2[threadIdx.x] = bn2[threadIdx.x];
        sbn1[threadIdx.x] = bn1[threadIdx.x];
    }
    bnb_from_slots<W>(bstats2, g.nslot, n, bb2, pre);
    const float *W2 = P + off_w2;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    float a[HK];   // A[in = col][k(s, half)] = W2[col][HK half + s]
#pragma unroll
    for (int k = 0; k < HK; ++k) a[k] = col < W ? W2[col * W + HK * half + k] : 0.0f;
    float gb[HK], s1[HK], q1[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) gb[k] = s1[k] = q1[k] = 0.0f;
    v16f DW;                          // WGRAD: d l_2/W[i][j], lane (j = col), rows i
#pragma unroll
    for (int v = 0; v < 16; ++v) DW[v] = 0.0f;
    float *sg = stage[wv][0], *sh = stage[wv][1], *sx = stage[wv][2];
    const float unused[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t ntiles = (g.npix + 31) >> 5, stride = (int64_t)gridDim.x * 4;
    int64_t T = (int64_t)blockIdx.x * 4 + wv;
    RowTile rg, rh, rx;
    if (T < ntiles) {
        rows_fetch<W>(rg, t1, T * 32, g.npix);
        rows_fetch<W>(rh, h2, T * 32, g.npix);
        rows_fetch<W>(rx, h1, T * 32, g.npix);
    }
    for (; T < ntiles; T += stride) {
        const int64_t p = T * 32 + col;
        const bool in = p < g.npix;
        wave_lds_fence();
        rows_park<W, false>(rg, sg, unused, unused, T * 32, g.npix);
        rows_park<W, false>(rh, sh, unused, unused, T * 32, g.npix);
        rows_park<W, false>(rx, sx, unused, unused, T * 32, g.npix);
        if (T + stride < ntiles) {
            rows_fetch<W>(rg, t1, (T + stride) * 32, g.npix);
            rows_fetch<W>(rh, h2, (T + stride) * 32, g.npix);
            rows_fetch<W>(rx, h1, (T + stride) * 32, g.npix);
        }
        wave_lds_fence();
        float gx[HK], hv[HK], x1[HK];
#pragma unroll
        for (int k = 0; k < HK; k += 4) {
            const float4 u = *reinterpret_cast<const float4 *>(sg + col * RP + HK * half + k);
            const float4 v = *reinterpret_cast<const float4 *>(sh + col * RP + HK * half + k);
            const float4 y = *reinterpret_cast<const float4 *>(sx + col * RP + mfma_row(k, half));
            gx[k] = u.x; gx[k + 1] = u.y; gx[k + 2] = u.z; gx[k + 3] = u.w;
            hv[k] = v.x; hv[k + 1] = v.y; hv[k + 2] = v.z; hv[k + 3] = v.w;
            x1[k] = y.x; x1[k + 1] = y.y; x1[k + 2] = y.z; x1[k + 3] = y.w;
        }
        v16f D;
#pragma unroll
        for (int v = 0; v < 16; ++v) D[v] = 0.0f;
#pragma unroll
        for (int k = 0; k < HK; ++k) {
            const int j = HK * half + k;
            const float rs = sbn2[W + j], xh = (hv[k] - sbn2[j]) * rs;
            const float gh2 = in ? rs * (gx[k] - bb2[j] - xh * bb2[W + j]) : 0.0f;
            gx[k] = gh2;
            gb[k] += gh2;
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], gh2, D, 0, 0, 0);
        }
        float o[HK];
#pragma unroll
        for (int v = 0; v < HK; ++v) {
            const int i = mfma_row(v, half);
            const float xh = (x1[v] - sbn1[i]) * sbn1[W + i];
            o[v] = (in && xh > 0.0f) ? D[v] : 0.0f;
            s1[v] += o[v];
            q1[v] = fmaf(o[v], xh, q1[v]);
        }
        wave_lds_fence();             // every lane has read its inputs: the tiles take the results
#pragma unroll
        for (int k = 0; k < HK; k += 4)
            *reinterpret_cast<float4 *>(sg + col * RP + HK * half + k) = make_float4(gx[k], gx[k + 1], gx[k + 2], gx[k + 3]);
        if (WGRAD) {
            wave_lds_fence();
            const int cw = col < W ? col : 0;   // rows / columns beyond the width are not used
            const float m1 = sbn1[cw], r1 = sbn1[W + cw];
#pragma unroll
            for (int k = 0; k < 16; ++k) {   // pixels 2 k + half of the tile: A1[p][i = col] (rows past the end hold g_h2 = 0)
                const float av = fmaxf((sx[(2 * k + half) * RP + cw] - m1) * r1, 0.0f);
                DW = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sg[(2 * k + half) * RP + cw], DW, 0, 0, 0);
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int k = 0; k < HK; k += 4)
            *reinterpret_cast<float4 *>(sx + col * RP + mfma_row(k, half)) = make_float4(o[k], o[k + 1], o[k + 2], o[k + 3]);
        wave_lds_fence();
        if (!WGRAD) rows_flush<W>(sg, t1, T * 32, g.npix);
        rows_flush<W>(sx, t2, T * 32, g.npix);
    }
    if (WGRAD) mfma_tile_to_slots(DW, red, G + off_w2, g.nslot, [](int i, int j) { return i < W && j < W ? i * W + j : -1; });
    lane_sums_to_slots(s1, red, bstats, g.nslot, [](int k, int h) { return mfma_row(k, h); });
    lane_sums_to_slots(q1, red, bstats, g.nslot, [](int k, int h) { return W + mfma_row(k, h); });
    lane_sums_to_slots(gb, red, G + off_w2 + W * W, g.nslot, [](int k, int h) { return HK * h + k; });
}

// transposed l_last + ReLU mask -> d loss / d xhat2 (t1) and the two batch sums of the BN2 backward (k_c3_dh at widths 16 / 32):
// g[p][i] = sum_(tap, q) W3[tap][i][q] gu[p - tap][q] — K = 36 = (tap, q), the 32 channels on M, pixels on N; the B operands
// come from a zero-bordered LDS tile of the patch's gu (K order: step s -> tap s >> 1, q = 2 half + (s & 1), one 8-byte
// read per tap), the mask from h2 through a staged tile that then takes the result.
// WGRAD: d l_last/W (k_w3_grad_mfma's sums) is accumulated here too — the h2 tile and the patch's gu tile are both in LDS.
// c3b.u != null: the elementwise stage in front of it (k_c3_bwd: affine transform / tanh / exp(3 logs) backward, from the kept
// l_last output u) runs while the gu tile is filled — one launch and one round trip of gu less; the first of the S workgroups
// of a patch also stores gu (for k_w3_grad_mfma), the updated dz and the 9 scalar sums.
struct C3Bwd {
    const float *u, *zin, *zlat;   // zlat != null: first stage of the backward pass (d loss / d latent = latent / B)
    const float *dz;               // d loss / d (coupling output), read by EVERY workgroup of the patch ...
    float *dz_out, *gu_out;        // ... so the updated one goes to a second buffer (the coupling's last stage reads it from there)
    float invB;
};
template <int W, bool WGRAD>
__global__ __launch_bounds__(256) void k_c3_dh_mfma(Geo g, const float *__restrict__ h2, const float *__restrict__ bn2,
                                                      const float *__restrict__ P, int off_w3, const float *__restrict__ gu,
                                                      float *__restrict__ t1, Acc bstats, Acc G, int S, C3Bwd c3b)
{
    constexpr int RP = W + 4, HK = W / 2;
    extern __shared__ float smem[];   // gu tile [(H+2)(W+2)][4], then the tile index of every pixel of a patch (int)
    __shared__ float stage[4][32 * 36];
    __shared__ float sbn2[2 * W];
    __shared__ float cs[2][4][64], cst[40];
    float *red = &stage[0][0];        // [4][64][16], used once the pixel loop is over
    const float *W3 = P + off_w3;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    const int Wp = g.W + 2, tile_px = (g.H + 2) * Wp;
    if (t < 2 * W) sbn2[t] = bn2[t];
    float a[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) a[k] = col < W ? W3[(k >> 1) * (W + 1) * 4 + col * 4 + 2 * half + (k & 1)] : 0.0f;
    float s2[HK], q2[HK];
#pragma unroll
    for (int k = 0; k < HK; ++k) s2[k] = q2[k] = 0.0f;
    // WGRAD (see k_w3_grad_mfma): columns (tap, q), taps 0..7 in D0, tap 8 in D1; S0 / S1 = column sums of the B operands
    const int cw = col < W ? col : 0, wq = col & 3, wd0 = ((col >> 2) / 3 - 1) * Wp + ((col >> 2) % 3 - 1), wd1 = Wp + 1;
    v16f D0, D1;
#pragma unroll
    for (int v = 0; v < 16; ++v) D0[v] = D1[v] = 0.0f;
    float S0 = 0.0f, S1 = 0.0f;
    float *sx = stage[wv];
    const float unused[4] = {0.f, 0.f, 0.f, 0.f};
    // the elementwise stage (c3b.u): d l_last/b (4), d logs (4), d rescale — adjacent in the raw layout
    const float *b3 = W3 + 36 * (W + 1), *logs = b3 + 4;
    const float sc = logs[4];
    float e3[4], tail[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) e3[k] = expf(kLogscale * logs[k]);
    int *lut = reinterpret_cast<int *>(smem + tile_px * 4);
    for (int i = t; i < tile_px * 4; i += 256) smem[i] = 0.0f;
    for (int px = t; px < g.HW; px += 256) {
        const int r = px / g.W;
        lut[px] = (r + 1) * Wp + (px - r * g.W) + 1;
    }
    const int npatch = (int)(g.npix / g.HW), ntiles = (g.HW + 31) >> 5;
    for (int unit = blockIdx.x; unit < npatch * S; unit += gridDim.x) {   // S workgroups share a patch's tiles
        const int b = unit / S, T0 = wv + 4 * (unit - b * S);
        const int64_t pb = (int64_t)b * g.HW;
        RowTile rx;
        if (T0 < ntiles) rows_fetch<W>(rx, h2, pb + T0 * 32, pb + g.HW);
        __syncthreads();              // the border is zero / the previous patch is done with
        if (c3b.u) {
            const bool first = unit == b * S;
            for (int px = t; px < g.HW; px += 256) {
                const int64_t p = pb + px;
                const float4 uv = reinterpret_cast<const float4 *>(c3b.u)[p], zi = reinterpret_cast<const float4 *>(c3b.zin)[p];
                float4 d;
                if (c3b.zlat) {
                    const float4 zl = reinterpret_cast<const float4 *>(c3b.zlat)[p];
                    d = make_float4(zl.x * c3b.invB, zl.y * c3b.invB, zl.z * c3b.invB, zl.w * c3b.invB);
                } else {
                    d = reinterpret_cast<const float4 *>(c3b.dz)[p];
                }
                const float uu[4] = {uv.x, uv.y, uv.z, uv.w}, z1[2] = {zi.z, zi.w}, gx1[2] = {d.z, d.w};
                float go[4], o[4], guv[4], gz1[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = uu[k] * e3[k];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float th = tanhf(o[2 + k]), E = expf(sc * th);
                    gz1[k] = gx1[k] * E;
                    const float gls = gx1[k] * z1[k] * E - c3b.invB;   // loss = mean(-(sum ls + ...))
                    if (first) tail[8] = fmaf(gls, th, tail[8]);
                    go[k] = gx1[k];                                  // shift
                    go[2 + k] = gls * sc * (1.0f - th * th);         // raw
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    guv[k] = go[k] * e3[k];
                    if (first) {
                        tail[4 + k] = fmaf(kLogscale * go[k], o[k], tail[4 + k]);
                        tail[k] += guv[k];
                    }
                }
                const float4 gv = make_float4(guv[0], guv[1], guv[2], guv[3]);
                reinterpret_cast<float4 *>(smem)[lut[px]] = gv;
                if (first) {
                    reinterpret_cast<float4 *>(c3b.gu_out)[p] = gv;
                    d.z = gz1[0];
                    d.w = gz1[1];
                    reinterpret_cast<float4 *>(c3b.dz_out)[p] = d;
                }
            }
        } else {
            for (int px = t; px < g.HW; px += 256) reinterpret_cast<float4 *>(smem)[lut[px]] = reinterpret_cast<const float4 *>(gu)[pb + px];
        }
        __syncthreads();
        for (int T = T0; T < ntiles; T += 4 * S) {
            const int pp = T * 32 + col;
            const bool in = pp < g.HW;
            wave_lds_fence();
            rows_park<W, false>(rx, sx, unused, unused, pb + T * 32, pb + g.HW);
            if (T + 4 * S < ntiles) rows_fetch<W>(rx, h2, pb + (T + 4 * S) * 32, pb + g.HW);
            const float *gt = smem + (in ? lut[pp] : 0) * 4 + 2 * half;
            v16f D;
#pragma unroll
            for (int v = 0; v < 16; ++v) D[v] = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float2 bv = in ? *reinterpret_cast<const float2 *>(gt - ((tap / 3 - 1) * Wp + (tap % 3 - 1)) * 4) : make_float2(0.f, 0.f);
                D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * tap], bv.x, D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * tap + 1], bv.y, D, 0, 0, 0);
            }
            wave_lds_fence();
            if (WGRAD) {
                const float m2 = sbn2[cw], r2 = sbn2[W + cw];
#pragma unroll 4
                for (int k = 0; k < 16; ++k) {
                    const int p2 = T * 32 + 2 * k + half;
                    float b0 = 0.0f, b1 = 0.0f;
                    if (p2 < g.HW) {
                        const int tp = lut[p2];
                        b0 = smem[(tp - wd0) * 4 + wq];
                        if (col < 4) b1 = smem[(tp - wd1) * 4 + wq];
                    }
                    const float av = p2 < g.HW ? fmaxf((sx[(2 * k + half) * RP + cw] - m2) * r2, 0.0f) : 0.0f;
                    S0 += b0;
                    S1 += b1;
                    D0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, D0, 0, 0, 0);
                    D1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, D1, 0, 0, 0);
                }
            }
            float o[HK];
#pragma unroll
            for (int v = 0; v < HK; v += 4) {
                const float4 y = *reinterpret_cast<const float4 *>(sx + col * RP + mfma_row(v, half));
                const float hv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = mfma_row(v + k, half);
                    const float xh = (hv[k] - sbn2[i]) * sbn2[W + i];
                    o[v + k] = (in && xh > 0.0f) ? D[v + k] : 0.0f;
                    s2[v + k] += o[v + k];
                    q2[v + k] = fmaf(o[v + k], xh, q2[v + k]);
                }
            }
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < HK; v += 4)
                *reinterpret_cast<float4 *>(sx + col * RP + mfma_row(v, half)) = make_float4(o[v], o[v + 1], o[v + 2], o[v + 3]);
            wave_lds_fence();
            rows_flush<W>(sx, t1, pb + T * 32, pb + g.HW);
        }
    }
    lane_sums_to_slots(s2, red, bstats, g.nslot, [](int k, int h) { return mfma_row(k, h); });
    lane_sums_to_slots(q2, red, bstats, g.nslot, [](int k, int h) { return W + mfma_row(k, h); });
    if (c3b.u) acc_add_n<9>(G + off_w3 + 36 * (W + 1), tail, g.nslot);
    if (WGRAD) {
        const Acc dst = G + off_w3;
        mfma_tile_to_slots(D0, red, dst, g.nslot, [](int i, int c) { return i < W ? (c >> 2) * (W + 1) * 4 + i * 4 + (c & 3) : -1; });
        mfma_tile_to_slots(D1, red, dst, g.nslot, [](int i, int c) { return i < W && c < 4 ? 8 * (W + 1) * 4 + i * 4 + c : -1; });
        cs[0][wv][ln] = S0;
        cs[1][wv][ln] = S1;
        __syncthreads();
        if (t < 36) {   // column sums over the 4 wavefronts and both lane halves: t = tap * 4 + q
            const int k = t < 32 ? 0 : 1, c = t < 32 ? t : t - 32;
            float tot = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) tot += cs[k][w][c] + cs[k][w][c + 32];
            cst[t] = tot;
        }
        __syncthreads();
        if (t < 36) {   // indicator channel: the pixels whose tap falls on the padding ring
            float *d = dst.p + (size_t)((t >> 2) * (W + 1) * 4 + W * 4 + (t & 3)) * NSLOT;
            d[blockIdx.x] = cst[16 + (t & 3)] - cst[t];
            for (int k = blockIdx.x + gridDim.x; k < g.nslot; k += gridDim.x) d[k] = 0.0f;
        }
    }
}

// transposed l_1 (k_c1_dz at widths 16 / 32): d z0[p][c] += sum_tap sum_j W1[tap][c][j] g_h1[p - tap][j], evaluated like the
// l_last forward: Q[p][(tap, c)] = sum_j W1[tap][c][j] g_h1[p][j] (18 rows of a 32-row tile, K = the 32 channels) for a band
// of rows + a one-row halo into LDS, then one thread per pixel adds its 9 taps up and runs the folded Conv2d1x1 backward.
template <int W, bool MIX>
__global__ __launch_bounds__(256) void k_c1_dz_mfma(Geo g, const float *__restrict__ t2, const float *__restrict__ P, int off_w1,
                                                      float *__restrict__ dz, const float *__restrict__ zmix_in,
                                                      const float *__restrict__ A, Acc dA, int BR, const float *__restrict__ dz_in)
{
    constexpr int QS = 20, RP = W + 4, HK = W / 2;
    extern __shared__ float smem[];   // Q [pixels of the band + halo][20]
    __shared__ float stage[4][32 * 36];
    const float *W1 = P + off_w1;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    float a[HK];                      // A[(tap, c) = col][j = 2 s + half]
#pragma unroll
    for (int k = 0; k < HK; ++k) a[k] = col < 18 ? W1[(col >> 1) * 2 * W + (col & 1) * W + 2 * k + half] : 0.0f;
    float mm[16], acc[16];
    if (MIX) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            mm[i] = A[i];
            acc[i] = 0.0f;
        }
    }
    const float unused[4] = {0.f, 0.f, 0.f, 0.f};
    float *sa = stage[wv];
    const int npatch = (int)(g.npix / g.HW), nbands = (g.H + BR - 1) / BR, units = npatch * nbands;
    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int b = unit / nbands, r0 = (unit - b * nbands) * BR, r1 = min(r0 + BR, g.H);
        const int rlo = max(r0 - 1, 0), rhi = min(r1 + 1, g.H), ntiles = ((rhi - rlo) * g.W + 31) >> 5;
        const int64_t base = (int64_t)b * g.HW + rlo * g.W, end = (int64_t)b * g.HW + rhi * g.W;
        RowTile ra;
        if (wv < ntiles) rows_fetch<W>(ra, t2, base + wv * 32, end);
        __syncthreads();              // the previous band's gather is over
        for (int T = wv; T < ntiles; T += 4) {
            wave_lds_fence();
            rows_park<W, false>(ra, sa, unused, unused, base + T * 32, end);
            if (T + 4 < ntiles) rows_fetch<W>(ra, t2, base + (T + 4) * 32, end);
            wave_lds_fence();
            v16f D;
#pragma unroll
            for (int v = 0; v < 16; ++v) D[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < HK; ++k) D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], sa[col * RP + 2 * k + half], D, 0, 0, 0);
            float *qp = smem + (T * 32 + col) * QS;   // rows 0..19 of the result (18, 19 are zero)
            *reinterpret_cast<float4 *>(qp + 4 * half) = make_float4(D[0], D[1], D[2], D[3]);
            *reinterpret_cast<float4 *>(qp + 8 + 4 * half) = make_float4(D[4], D[5], D[6], D[7]);
            if (half == 0) *reinterpret_cast<float4 *>(qp + 16) = make_float4(D[8], D[9], D[10], D[11]);
        }
        __syncthreads();
        for (int px = t; px < (r1 - r0) * g.W; px += 256) {
            const int rr0 = px / g.W, r = r0 + rr0, c = px - rr0 * g.W;
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int qr = r - (di - 1);
                if (qr < 0 || qr >= g.H) continue;
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int qc = c - (dj - 1);
                    if (qc < 0 || qc >= g.W) continue;
                    const float2 qv = *reinterpret_cast<const float2 *>(smem + ((qr - rlo) * g.W + qc) * QS + (di * 3 + dj) * 2);
                    a0 += qv.x;
                    a1 += qv.y;
                }
            }
            const int64_t p = (int64_t)b * g.HW + r * g.W + c;
            if (MIX) {
                const float4 dv = reinterpret_cast<const float4 *>(dz_in ? dz_in : dz)[p], zv = reinterpret_cast<const float4 *>(zmix_in)[p];
                const float d[4] = {dv.x + a0, dv.y + a1, dv.z, dv.w}, zi[4] = {zv.x, zv.y, zv.z, zv.w};
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = mm[i * 4] * d[0] + mm[i * 4 + 1] * d[1] + mm[i * 4 + 2] * d[2] + mm[i * 4 + 3] * d[3];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i * 4 + j] = fmaf(zi[i], d[j], acc[i * 4 + j]);
                }
                reinterpret_cast<float4 *>(dz)[p] = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                if (dz_in) {
                    const float4 v = reinterpret_cast<const float4 *>(dz_in)[p];
                    reinterpret_cast<float4 *>(dz)[p] = make_float4(v.x + a0, v.y + a1, v.z, v.w);
                } else {
                    float2 *d = reinterpret_cast<float2 *>(dz + p * 4);
                    const float2 v = *d;
                    *d = make_float2(v.x + a0, v.y + a1);
                }
            }
        }
    }
    if (MIX) acc_add_n<16>(dA, acc, g.nslot);
}

// l_1 (3x3 SAME conv of the pass-through half, folded Conv2d1x1) + bias + statistics at widths 16 / 32 (k_c1_fwd): K = (tap, c) = 18,
// step = tap, lane half = c; the B operands from a zero-bordered LDS tile of the patch's (mixed) pass-through channels.
template <int W, bool MIX>
__global__ __launch_bounds__(256) void k_c1_fwd_mfma(Geo g, const float *__restrict__ zin, const float *__restrict__ A,
                                                       float *__restrict__ zmixed, const float *__restrict__ P, int off,
                                                       float *__restrict__ h1, Acc stats, int S)
{
    constexpr int RP = W + 4, HK = W / 2;
    extern __shared__ float smem[];   // z tile [(H+2)(W+2)][2], then the tile index of every pixel of a patch (int)
    __shared__ float stage[4][32 * 36];
    float *red = &stage[0][0];        // [4][64][16], used once the pixel loop is over
    const float *W1 = P + off, *b1 = W1 + 18 * W;
    const int t = threadIdx.x, wv = t >> 6, ln = t & 63, col = ln & 31, half = ln >> 5;
    const int Wp = g.W + 2, tile_px = (g.H + 2) * Wp;
    float a[9], bo[HK], s1[HK], q1[HK];
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = col < W ? W1[k * 2 * W + half * W + col] : 0.0f;
#pragma unroll
    for (int k = 0; k < HK; ++k) {
        bo[k] = b1[mfma_row(k, half)];
        s1[k] = q1[k] = 0.0f;
    }
    float mm[16];
    if (MIX) {
#pragma unroll
        for (int i = 0; i < 16; ++i) mm[i] = A[i];
    }
    float *so = stage[wv];
    int *lut = reinterpret_cast<int *>(smem + tile_px * 2);
    for (int i = t; i < tile_px * 2; i += 256) smem[i] = 0.0f;
    for (int px = t; px < g.HW; px += 256) {
        const int r = px / g.W;
        lut[px] = (r + 1) * Wp + (px - r * g.W) + 1;
    }
    const int npatch = (int)(g.npix / g.HW), ntiles = (g.HW + 31) >> 5;
    for (int unit = blockIdx.x; unit < npatch * S; unit += gridDim.x) {   // S workgroups share a patch's tiles
        const int b = unit / S, part = unit - b * S;
        const int64_t pb = (int64_t)b * g.HW;
        __syncthreads();              // the border is zero / the previous patch is done with
        for (int px = t; px < g.HW; px += 256) {
            const float4 u = reinterpret_cast<const float4 *>(zin)[pb + px];
            float2 v = make_float2(u.x, u.y);
            if (MIX) {
                v.x = u.x * mm[0] + u.y * mm[4] + u.z * mm[8] + u.w * mm[12];
                v.y = u.x * mm[1] + u.y * mm[5] + u.z * mm[9] + u.w * mm[13];
                if (part == 0)
                    reinterpret_cast<float4 *>(zmixed)[pb + px] = make_float4(v.x, v.y, u.x * mm[2] + u.y * mm[6] + u.z * mm[10] + u.w * mm[14],
                                                                             u.x * mm[3] + u.y * mm[7] + u.z * mm[11] + u.w * mm[15]);
            }
            reinterpret_cast<float2 *>(smem)[lut[px]] = v;
        }
        __syncthreads();
        for (int T = wv + 4 * part; T < ntiles; T += 4 * S) {
            const int pp = T * 32 + col;
            const bool in = pp < g.HW;
            const float *zt = smem + (in ? lut[pp] : 0) * 2 + half;
            v16f D;
#pragma unroll
            for (int v = 0; v < 16; ++v) D[v] = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float bv = in ? zt[((tap / 3 - 1) * Wp + (tap % 3 - 1)) * 2] : 0.0f;
                D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tap], bv, D, 0, 0, 0);
            }
            wave_lds_fence();         // the previous tile's flush has been issued
#pragma unroll
            for (int v = 0; v < HK; v += 4) {
                const float4 o = make_float4(D[v] + bo[v], D[v + 1] + bo[v + 1], D[v + 2] + bo[v + 2], D[v + 3] + bo[v + 3]);
                *reinterpret_cast<float4 *>(so + col * RP + mfma_row(v, half)) = o;
                if (in) {
                    s1[v] += o.x; s1[v + 1] += o.y; s1[v + 2] += o.z; s1[v + 3] += o.w;
                    q1[v] = fmaf(o.x, o.x, q1[v]); q1[v + 1] = fmaf(o.y, o.y, q1[v + 1]);
                    q1[v + 2] = fmaf(o.z, o.z, q1[v + 2]); q1[v + 3] = fmaf(o.w, o.w, q1[v + 3]);
                }
            }
            wave_lds_fence();
            rows_flush<W>(so, h1, pb + T * 32, pb + g.HW);
        }
    }
    lane_sums_to_slots(s1, red, stats, g.nslot, [](int k, int h) { return mfma_row(k, h); });
    lane_sums_to_slots(q1, red, stats, g.nslot, [](int k, int h) { return W + mfma_row(k, h); });
}

