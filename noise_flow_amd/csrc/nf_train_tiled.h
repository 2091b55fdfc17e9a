// nf_train_tiled.h — part of nf_train.hip (included inside its anonymous namespace; not a standalone header).
// The per-patch tiled stages of the trainer: couplings of width 4 / 8 on patches of up to 1 024 pixels.
// ---------------------------------------------------------------------------------------------
// tiled backward stages: ONE workgroup per patch, the 3x3 neighbourhoods through zero-bordered LDS tiles
// ---------------------------------------------------------------------------------------------
// On the main stream the backward of a coupling is a chain of five layer kernels (k_c3_bwd -> k_c3_dh -> k_c2_bwd ->
// k_c1_bwd -> k_c1_dz); only the two batch reductions of the BN backward formula really separate them.  For couplings of
// width <= 8 on patches of up to 1024 pixels the chain is cut at exactly those two points:
//   A' = k_c3_bwd + k_c3_dh   |   k_c2_bwd   |   C' = k_c1_bwd + k_c1_dz (+ the folded Conv2d1x1)
// and C' of one coupling shares its launch with A' of the coupling below it: 2 launches per coupling on the critical path
// instead of 5.  The three filter-gradient kernels stay what they are, on the side stream, fed by the tensors these
// stages leave in HBM (gu, t1, t2) — fusing THEM in was measured to cost more than it saves (their ~290 value sums per
// coupling multiply with the wavefront count).  One thread per pixel, NT = 256 / 512 / 1024 threads by patch size.
// The sums of a launch are reduced at its END, all groups together: every group parks its 16-lane row sums (row_sum16: 4 issues per
// value instead of wsum's 11) in a region of its own, ONE barrier, then thread t0 + k adds value k up — wavefront by wavefront, its four
// rows joined as wsum joins them — and stores this workgroup's slot directly.  (Round 6; before, every group had a barrier pair of
// its own and went through a staging array: 9 barriers of 16 wavefronts per k_tiled_CA launch.)
template <int N, int NT>
__device__ __forceinline__ void stage_put_n(const float (&v)[N], float *part /* [NT/16][N] */)
{
    const int row = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float sv = row_sum16(v[k]);
        if ((threadIdx.x & 15) == 0) part[row * N + k] = sv;
    }
}
template <int N, int NT>
__device__ __forceinline__ void stage_sum_flush(Acc dst, const float *part, int t0, int nslot)
{
    const int k = (int)threadIdx.x - t0;
    if (k >= 0 && k < N) {
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) {
            const float *q = part + (4 * i) * N + k;
            tot += (q[0] + q[N]) + (q[2 * N] + q[3 * N]);
        }
        float *d = dst.p + (size_t)k * NSLOT;
        d[blockIdx.x] = tot;
        for (int q = blockIdx.x + gridDim.x; q < nslot; q += gridDim.x) d[q] = 0.0f;
    }
}

// One pixel's W values of a [pixel][W] LDS tile as 16-byte accesses (W = 4 / 8: the entry is 16-byte aligned).  Scalar accesses
// put consecutive lanes W words apart — 8 distinct banks for 32 lanes, a 4-way conflict on every read (counters: 56 - 68 % of
// the LDS-active cycles of these kernels); a 16-byte access per lane is conflict-free.
template <int W>
__device__ __forceinline__ void lds_get(const float *p, float (&v)[W])
{
#pragma unroll
    for (int j = 0; j < W; j += 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p + j);
        v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
    }
}
template <int W>
__device__ __forceinline__ void lds_put(float *p, const float (&v)[W])
{
#pragma unroll
    for (int j = 0; j < W; j += 4) *reinterpret_cast<float4 *>(p + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
}

template <int NT>
__device__ __forceinline__ void zero_floats(float *p, int n)
{
    for (int i = threadIdx.x; i < n; i += NT) p[i] = 0.0f;
}

template <int W, int NT>
__device__ __forceinline__ void bnb_from_slots_nt(Acc bstats, int nslot, double n, float *sh)
{
    for (int j = threadIdx.x >> 6; j < W; j += NT / 64) {
        double a, b;
        acc_total2(bstats + j, bstats + W + j, nslot, a, b);
        if ((threadIdx.x & 63) == 0) {
            sh[j] = (float)(a / n);
            sh[W + j] = (float)(b / n);
        }
    }
    __syncthreads();
}

// A'.  dz: d loss / d (coupling output) of the owned pixel in, with its second half replaced by d loss / d z1 out.
//   stg: [9] d l_last/b, d l_last/logs, d rescale (adjacent in the raw layout), then [2 W] BN2 sums
template <int W, int NT>
__device__ __forceinline__ void tiled_phase_A(const Geo &g, int b, int r, int c, bool ok, float (&dz)[4], const float4 zi,
                                              const float (&h2v)[W], const float *__restrict__ bn2,
                                              const float *__restrict__ P, int off_w3, float invB,
                                              float *__restrict__ gu, float *__restrict__ t1, float (&sq)[2 * W], float (&tail)[9], float *TH,
                                              float *TU)
{
    const int Wp = g.W + 2, tile_px = (g.H + 2) * Wp;
    const float *W3 = P + off_w3, *b3 = W3 + 36 * (W + 1), *logs = b3 + 4;
    const float sc = logs[4];
    const int64_t gp = (int64_t)b * g.HW + threadIdx.x;   // this thread's pixel in the batch tensors
    const int tp = (r + 1) * Wp + c + 1;                  // ... in the tiles
    zero_floats<NT>(TH, tile_px * W);
    zero_floats<NT>(TU, tile_px * 4);
    __syncthreads();
    if (ok) {
        float a2v[W];
#pragma unroll
        for (int i = 0; i < W; ++i) a2v[i] = fmaxf((h2v[i] - bn2[i]) * bn2[W + i], 0.0f);
        lds_put<W>(TH + tp * W, a2v);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) tail[k] = 0.0f;                      // d b (4), d logs (4), d rescale
    if (ok) {
        float u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = b3[q];
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int rr = r + di - 1;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int cc = c + dj - 1;
                const float *w = W3 + (di * 3 + dj) * (W + 1) * 4;
                if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {   // padding ring: zeros + indicator 1
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[q] += w[W * 4 + q];
                } else {
                    float hv[W];
                    lds_get<W>(TH + (tp + (di - 1) * Wp + (dj - 1)) * W, hv);
#pragma unroll
                    for (int i = 0; i < W; ++i) {
                        const float a = hv[i];
#pragma unroll
                        for (int q = 0; q < 4; ++q) u[q] = fmaf(a, w[i * 4 + q], u[q]);
                    }
                }
            }
        }
        const float z1[2] = {zi.z, zi.w}, gx1[2] = {dz[2], dz[3]};
        float go[4], o[4], e3[4], guv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e3[q] = expf(kLogscale * logs[q]);
            o[q] = u[q] * e3[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float t = tanhf(o[2 + q]), E = expf(sc * t);
            dz[2 + q] = gx1[q] * E;
            const float gls = gx1[q] * z1[q] * E - invB;   // loss = mean(-(sum ls + ...))
            tail[8] = fmaf(gls, t, tail[8]);
            go[q] = gx1[q];                                // shift
            go[2 + q] = gls * sc * (1.0f - t * t);         // raw
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tail[4 + q] = kLogscale * go[q] * o[q];
            guv[q] = go[q] * e3[q];
            tail[q] = guv[q];
        }
        const float4 gv = make_float4(guv[0], guv[1], guv[2], guv[3]);
        *reinterpret_cast<float4 *>(TU + tp * 4) = gv;
        reinterpret_cast<float4 *>(gu)[gp] = gv;           // k_w3_grad reads it on the side stream
    }
    __syncthreads();
    // transposed l_last + ReLU mask -> d loss / d xhat2 (t1) and the two batch sums of the BN backward formula
#pragma unroll
    for (int j = 0; j < 2 * W; ++j) sq[j] = 0.0f;
    if (ok) {
        float gh[W];
#pragma unroll
        for (int i = 0; i < W; ++i) gh[i] = 0.0f;
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const float4 gv = *reinterpret_cast<const float4 *>(TU + (tp - (di - 1) * Wp - (dj - 1)) * 4);
                const float *w = W3 + (di * 3 + dj) * (W + 1) * 4;
#pragma unroll
                for (int i = 0; i < W; ++i)
                    gh[i] += w[i * 4] * gv.x + w[i * 4 + 1] * gv.y + w[i * 4 + 2] * gv.z + w[i * 4 + 3] * gv.w;
            }
        float xhv[W];
        lds_get<W>(TH + tp * W, xhv);
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const float xh = xhv[i];                         // relu(xhat2): equals xhat2 wherever the mask lets gx through
            const float gx = xh > 0.0f ? gh[i] : 0.0f;
            t1[gp * W + i] = gx;
            sq[i] = gx;
            sq[W + i] = gx * xh;
        }
    }
}

// C'.  dz: the A' result of this coupling (read back from HBM) in, d loss / d (input of the layer below) out.
//   stg: [W] d l_1/b, then [16] d A
template <int W, int NT, bool MIX>
__device__ __forceinline__ void tiled_phase_C(const Geo &g, int b, int r, int c, bool ok, float (&dz)[4], const float4 zv,
                                              const float *__restrict__ A, const float (&h1v)[W], const float (&t2v)[W],
                                              const float *__restrict__ bn1, const float *bb1,
                                              const float *__restrict__ P, int off_w1, float *__restrict__ t2, float (&gh)[W],
                                              float (&accA)[16], float *TG)
{
    const int Wp = g.W + 2, tile_px = (g.H + 2) * Wp;
    const float *W1 = P + off_w1;
    const int64_t gp = (int64_t)b * g.HW + threadIdx.x;
    const int tp = (r + 1) * Wp + c + 1;
    zero_floats<NT>(TG, tile_px * W);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < W; ++j) gh[j] = 0.0f;
    if (ok) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float xh = (h1v[j] - bn1[j]) * bn1[W + j];
            gh[j] = bn1[W + j] * (t2v[j] - bb1[j] - xh * bb1[W + j]);
            t2[gp * W + j] = gh[j];                          // k_w1_grad reads it on the side stream
        }
        lds_put<W>(TG + tp * W, gh);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) accA[i] = 0.0f;
    if (ok) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                float gq[W];
                lds_get<W>(TG + (tp - (di - 1) * Wp - (dj - 1)) * W, gq);
                const float *w = W1 + (di * 3 + dj) * 2 * W;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    a0 = fmaf(w[j], gq[j], a0);
                    a1 = fmaf(w[W + j], gq[j], a1);
                }
            }
        const float d[4] = {dz[0] + a0, dz[1] + a1, dz[2], dz[3]};
        if (MIX) {
            const float zi[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dz[i] = A[i * 4] * d[0] + A[i * 4 + 1] * d[1] + A[i * 4 + 2] * d[2] + A[i * 4 + 3] * d[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) accA[i * 4 + j] = zi[i] * d[j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) dz[i] = d[i];
        }
    }
}

struct TiledA {   // operands of a stage A'
    const float *zin, *h2, *bn2;
    int off_w3;
    float *gu, *t1;
    Acc bstats2;
};

struct TiledC {   // operands of a stage C'
    const float *zpre, *A, *h1, *bn1;
    float *t2;
    int off_w1;
    Acc bstats1, dA;
};

// C' of one coupling (skipped for the first launch of the pass: HAS_C = false) and A' of the coupling below it (NEXT_A)
template <int W, int NT, bool HAS_C, bool MIX, bool NEXT_A>
__global__ __launch_bounds__(NT) void k_tiled_CA(Geo g, TiledC cc, TiledA a, double n, const float *__restrict__ P, float invB,
                                                 float *__restrict__ dz, const float *__restrict__ zlat, Acc G)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float bb1[2 * W];
    const int tile_px = (g.H + 2) * (g.W + 2);
    float *part = smem + tile_px * (W + 4);   // row sums of the four groups: [NT/16][W], [..][16], [..][2 W], [..][9]
    float *const partB = part + (NT / 16) * W, *const partS = partB + (NT / 16) * 16, *const partT = partS + (NT / 16) * 2 * W;
    const bool ok = (int)threadIdx.x < g.HW;
    const int r = ok ? (int)threadIdx.x / g.W : 0, c = ok ? (int)threadIdx.x - r * g.W : 0;
    const int b = blockIdx.x;
    const int64_t gp = (int64_t)b * g.HW + threadIdx.x;
    // every global operand of this pixel is requested up front: the stages below are a chain of short LDS phases, and one
    // exposed memory latency per stage (5 of them) was most of the kernel's time
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), zv = v, zi = v;
    float h1v[W], t2v[W], h2v[W];
#pragma unroll
    for (int j = 0; j < W; ++j) h1v[j] = t2v[j] = h2v[j] = 0.0f;
    if (ok) {
        if (zlat) {                                        // first stage of the pass: d loss / d latent = latent / B
            const float4 zl = reinterpret_cast<const float4 *>(zlat)[gp];
            v = make_float4(zl.x * invB, zl.y * invB, zl.z * invB, zl.w * invB);
        } else {
            v = reinterpret_cast<const float4 *>(dz)[gp];
        }
        if (HAS_C) {
#pragma unroll
            for (int j = 0; j < W; j += 4) {
                *reinterpret_cast<float4 *>(h1v + j) = *reinterpret_cast<const float4 *>(cc.h1 + gp * W + j);
                *reinterpret_cast<float4 *>(t2v + j) = *reinterpret_cast<const float4 *>(cc.t2 + gp * W + j);
            }
            if (MIX) zv = reinterpret_cast<const float4 *>(cc.zpre)[gp];
        }
        if (NEXT_A) {
#pragma unroll
            for (int j = 0; j < W; j += 4) *reinterpret_cast<float4 *>(h2v + j) = *reinterpret_cast<const float4 *>(a.h2 + gp * W + j);
            zi = reinterpret_cast<const float4 *>(a.zin)[gp];
        }
    }
    if (HAS_C) bnb_from_slots_nt<W, NT>(cc.bstats1, g.nslot, n, bb1);
    float d[4] = {v.x, v.y, v.z, v.w};
    [[maybe_unused]] float gb1[W], accA[16], sq[2 * W], tail[9];
    if (HAS_C) tiled_phase_C<W, NT, MIX>(g, b, r, c, ok, d, zv, cc.A, h1v, t2v, cc.bn1, bb1, P, cc.off_w1, cc.t2, gb1, accA, smem);
    if (HAS_C && NEXT_A) __syncthreads();   // phase A's tiles take the place of phase C's
    if (NEXT_A) tiled_phase_A<W, NT>(g, b, r, c, ok, d, zi, h2v, a.bn2, P, a.off_w3, invB, a.gu, a.t1, sq, tail, smem, smem + tile_px * W);
    if (ok) reinterpret_cast<float4 *>(dz)[gp] = make_float4(d[0], d[1], d[2], d[3]);
    // the launch's sums: parked, one barrier, added up and stored by their owner threads
    if (HAS_C) {
        stage_put_n<W, NT>(gb1, part);
        if (MIX) stage_put_n<16, NT>(accA, partB);
    }
    if (NEXT_A) {
        stage_put_n<2 * W, NT>(sq, partS);
        stage_put_n<9, NT>(tail, partT);
    }
    __syncthreads();
    if (HAS_C) {
        stage_sum_flush<W, NT>(G + cc.off_w1 + 18 * W, part, 0, g.nslot);            // d l_1/b
        if (MIX) stage_sum_flush<16, NT>(cc.dA, partB, 64, g.nslot);
    }
    if (NEXT_A) {
        stage_sum_flush<9, NT>(G + a.off_w3 + 36 * (W + 1), partT, 128, g.nslot);    // d l_last/b, d logs, d rescale
        stage_sum_flush<2 * W, NT>(a.bstats2, partS, 192, g.nslot);
    }
}

// ---- the forward pass, same idea: stage 3 of one coupling (BN2 + ReLU + l_last + affine) and stage 1 of the coupling above
// it (folded Conv2d1x1 + l_1 + its statistics) share a launch; k_c2_fwd stays between the two batch reductions.  The
// per-pixel arithmetic keeps the order of k_c1_fwd / l_last_u (the zero border of the tiles adds exact zeros), so h1, u and
// z are bit-identical to the layer kernels'; only the grouping of the fp32 partial sums of the statistics differs.
template <int W, int NT>
__device__ __forceinline__ void bn_from_slots_nt(Acc stats, int nslot, double n, float *sh, float *__restrict__ P, int off_mean,
                                                 int off_var, float *__restrict__ bn_out)
{
    for (int j = threadIdx.x >> 6; j < W; j += NT / 64) {
        double sm, sq;
        acc_total2(stats + j, stats + W + j, nslot, sm, sq);
        const double m = sm / n;
        double v = sq / n - m * m;
        if (v < 0.0) v = 0.0;
        if ((threadIdx.x & 63) == 0) {
            const float mf = (float)m, rf = (float)(1.0 / sqrt(v + (double)kBnEps));
            sh[j] = mf;
            sh[W + j] = rf;
            if (blockIdx.x == 0) {
                bn_out[j] = mf;
                bn_out[W + j] = rf;
                P[off_mean + j] -= kBnDecay * (P[off_mean + j] - mf);
                P[off_var + j] -= kBnDecay * (P[off_var + j] - (float)v);
            }
        }
    }
    __syncthreads();
}

struct TiledF3 {   // operands of stage 3 of a coupling
    const float *zin, *h2;
    Acc stats2;
    int off_m2, off_w3;
    float *bn2_out, *zout;
    Acc ldacc;
};

struct TiledF1 {   // operands of stage 1 of a coupling (A / zmixed: the folded Conv2d1x1)
    const float *A;
    float *zmixed;
    int off_w1;
    float *h1;
    Acc stats1;
};

template <int W, int NT, bool HAS_3, bool MIX, bool HAS_1>
__global__ __launch_bounds__(NT) void k_tiled_fwd(Geo g, TiledF3 f3, TiledF1 f1, const float *__restrict__ zsrc, double n,
                                                  float *__restrict__ P, const float *__restrict__ Pw)
{   // Pw = P.  The filters are read through their own read-only pointer: behind the pointer the running moments are written
    // through, the compiler cannot prove them unclobbered and fetches every (wavefront-uniform) weight with a vector load
    // instead of a scalar one — 76 extra vector loads per thread in this kernel.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float bn2[2 * W];
    const int Wp = g.W + 2, tile_px = (g.H + 2) * Wp;
    float *TH = smem, *TZ = smem + tile_px * W, *part = TZ + tile_px * 2;   // row sums: [NT/16][1] (log-det), [NT/16][2 W] (statistics)
    float *const partS = part + NT / 16;
    const bool ok = (int)threadIdx.x < g.HW;
    const int r = ok ? (int)threadIdx.x / g.W : 0, c = ok ? (int)threadIdx.x - r * g.W : 0;
    const int b = blockIdx.x, tp = (r + 1) * Wp + c + 1;
    const int64_t gp = (int64_t)b * g.HW + threadIdx.x;
    // the pixel's global operands are requested before the statistics are added up (see k_tiled_CA)
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float h2v[W];
#pragma unroll
    for (int j = 0; j < W; ++j) h2v[j] = 0.0f;
    if (ok) {
        if (HAS_3) {
#pragma unroll
            for (int j = 0; j < W; j += 4) *reinterpret_cast<float4 *>(h2v + j) = *reinterpret_cast<const float4 *>(f3.h2 + gp * W + j);
            z = reinterpret_cast<const float4 *>(f3.zin)[gp];
        } else {
            z = reinterpret_cast<const float4 *>(zsrc)[gp];
        }
    }
    zero_floats<NT>(smem, tile_px * (W + 2));
    if (HAS_3) bn_from_slots_nt<W, NT>(f3.stats2, g.nslot, n, bn2, P, f3.off_m2, f3.off_m2 + W, f3.bn2_out);
    __syncthreads();
    float lv[1] = {0.0f};
    if (HAS_3) {
        const float *W3 = Pw + f3.off_w3, *b3 = W3 + 36 * (W + 1), *logs = b3 + 4;
        if (ok) {
            float a2v[W];
#pragma unroll
            for (int i = 0; i < W; ++i) a2v[i] = fmaxf((h2v[i] - bn2[i]) * bn2[W + i], 0.0f);
            lds_put<W>(TH + tp * W, a2v);
        }
        __syncthreads();
        if (ok) {
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = b3[k];
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int rr = r + di - 1;
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int cc = c + dj - 1;
                    const float *w = W3 + (di * 3 + dj) * (W + 1) * 4;
                    if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) {   // on the padding ring: zeros + indicator 1
#pragma unroll
                        for (int k = 0; k < 4; ++k) u[k] += w[W * 4 + k];
                    } else {
                        float hv[W];
                        lds_get<W>(TH + (tp + (di - 1) * Wp + (dj - 1)) * W, hv);
#pragma unroll
                        for (int i = 0; i < W; ++i) {
                            const float a = hv[i];
#pragma unroll
                            for (int k = 0; k < 4; ++k) u[k] = fmaf(a, w[i * 4 + k], u[k]);
                        }
                    }
                }
            }
            const float sc = logs[4];
            const float4 zi = z;
            const float sh0 = u[0] * expf(kLogscale * logs[0]), sh1 = u[1] * expf(kLogscale * logs[1]);
            const float ls0 = sc * tanhf(u[2] * expf(kLogscale * logs[2])), ls1 = sc * tanhf(u[3] * expf(kLogscale * logs[3]));
            z = make_float4(zi.x, zi.y, fmaf(zi.z, expf(ls0), sh0), fmaf(zi.w, expf(ls1), sh1));
            reinterpret_cast<float4 *>(f3.zout)[gp] = z;
            lv[0] = ls0 + ls1;
        }
    }
    [[maybe_unused]] float sq[2 * W];
#pragma unroll
    for (int j = 0; j < 2 * W; ++j) sq[j] = 0.0f;
    if (HAS_1) {
        const float *W1 = Pw + f1.off_w1, *b1 = W1 + 18 * W;
        if (ok) {
            float2 v = make_float2(z.x, z.y);
            if (MIX) {
                const float *m = f1.A;
                v.x = z.x * m[0] + z.y * m[4] + z.z * m[8] + z.w * m[12];
                v.y = z.x * m[1] + z.y * m[5] + z.z * m[9] + z.w * m[13];
                reinterpret_cast<float4 *>(f1.zmixed)[gp] = make_float4(v.x, v.y, z.x * m[2] + z.y * m[6] + z.z * m[10] + z.w * m[14],
                                                                        z.x * m[3] + z.y * m[7] + z.z * m[11] + z.w * m[15]);
            }
            *reinterpret_cast<float2 *>(TZ + tp * 2) = v;
        }
        __syncthreads();
        if (ok) {
            float h[W];
#pragma unroll
            for (int j = 0; j < W; ++j) h[j] = b1[j];
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int rr = r + di - 1, cc = c + dj - 1;
                    if (rr < 0 || rr >= g.H || cc < 0 || cc >= g.W) continue;
                    const float2 v = *reinterpret_cast<const float2 *>(TZ + (tp + (di - 1) * Wp + (dj - 1)) * 2);
                    const float *w = W1 + (di * 3 + dj) * 2 * W;
#pragma unroll
                    for (int j = 0; j < W; ++j) h[j] = fmaf(v.x, w[j], fmaf(v.y, w[W + j], h[j]));
                }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                f1.h1[gp * W + j] = h[j];
                sq[j] = h[j];
                sq[W + j] = h[j] * h[j];
            }
        }
    }
    if (HAS_3) stage_put_n<1, NT>(lv, part);
    if (HAS_1) stage_put_n<2 * W, NT>(sq, partS);
    __syncthreads();
    if (HAS_3) stage_sum_flush<1, NT>(f3.ldacc, part, 0, g.nslot);
    if (HAS_1) stage_sum_flush<2 * W, NT>(f1.stats1, partS, 64, g.nslot);
}

