// What the four GEMM kernels (nf_gemm.hip, nf_gemm16.hip: variants A and B each) share: everything AROUND the coupling CNN —
// where a patch (or, NF_K_TILED, a tile of an image) sits, the input draw / load, the per-pixel layers (Conv2d1x1, the sdn and
// gain families), the affine half of a coupling behind its CNN, and the epilogue (prior, log-det, batch sums).  The CNNs
// themselves (three GEMMs per coupling in two operand precisions and two work splits) stay in their kernels.
//
// Pixel ownership is the kernels': thread t owns pixels p = t + GT m, m < OWN, at (pr[m], pc[m]) of the patch, act[m] = inside it.
//
// Replaces (reference, /root/reference): layers.py:108-124 (Conv2d1x1), :251-375 (AffineCoupling, the part behind the CNN),
// cond_utils.py:205-239 (sdn), noise_flow_model.py:394-428, :477-478, :537-539 (objective, sd_z, prior).
#pragma once
#include <atomic>
#include <type_traits>

namespace {

// Where this "patch" sits: on its own ([B,H,W,4] tensors), or — NF_K_TILED (nf_device.h, "overlapping tiles") — as tile
// b % tiles of image b / tiles: pixel (r, c) of the tile is pixel (oy + r, ox + c) of an IH x IW image, border masks follow the
// image border, and results are reported for the core window [cy0, cy1) x [cx0, cx1) only.
struct GemmTile {
    size_t patch_off;    // floats in front of the patch's / image's tensor
    int64_t patch_id;    // Philox key
    int oy, ox, IH, IW, cy0, cy1, cx0, cx1;
    bool tiled;
    __device__ __forceinline__ int gi(bool act, int r, int c) const { return act ? (oy + r) * IW + ox + c : 0; }   // index in the tensors
    __device__ __forceinline__ bool own(bool act, int r, int c) const                                               // reported by this launch
    {
        const int R = oy + r, C = ox + c;
        return act && R >= cy0 && R < cy1 && C >= cx0 && C < cx1;
    }
    __device__ __forceinline__ int border(int r, int c) const    // index into a coupling's 16-entry border table
    {
        return (oy + r == 0 ? 1 : 0) | (oy + r == IH - 1 ? 2 : 0) | (ox + c == 0 ? 4 : 0) | (ox + c == IW - 1 ? 8 : 0);
    }
};

__device__ __forceinline__ GemmTile gemm_tile(const NfLaunch &a, int64_t b, int H, int W)
{
    GemmTile T;
    T.patch_off = (size_t)b * (size_t)(H * W) * 4u;
    T.patch_id = b;
    T.oy = 0; T.ox = 0; T.IH = H; T.IW = W; T.cy0 = 0; T.cy1 = H; T.cx0 = 0; T.cx1 = W;
    T.tiled = (a.flags & NF_K_TILED) != 0;
    if (T.tiled) {
        const int nt = a.tile_ny * a.tile_nx;
        const int64_t img = b / nt;
        const int ti = (int)(b - img * nt);
        const int ty = ti / a.tile_nx, tx = ti - ty * a.tile_nx;
        T.IH = a.img_H;
        T.IW = a.img_W;
        T.oy = nf_tile_origin(ty, T.IH, H, a.tile_halo);
        T.ox = nf_tile_origin(tx, T.IW, W, a.tile_halo);
        T.cy0 = nf_tile_core0(ty, T.IH, H, a.tile_halo);
        T.cy1 = nf_tile_core1(ty, a.tile_ny, T.IH, H, a.tile_halo);
        T.cx0 = nf_tile_core0(tx, T.IW, W, a.tile_halo);
        T.cx1 = nf_tile_core1(tx, a.tile_nx, T.IW, W, a.tile_halo);
        T.patch_off = (size_t)img * (size_t)T.IH * (size_t)T.IW * 4u;
        T.patch_id = img;
    }
    return T;
}

// the 4 channels of each owned pixel -> registers: the in-kernel Philox / Box-Muller draw, or the input tensor
template <int OWN, bool PHILOX>
__device__ __forceinline__ void gemm_input(const NfLaunch &a, const GemmTile &T, const int (&pr)[OWN], const int (&pc)[OWN], const bool (&act)[OWN],
                                           float (&z)[OWN][4])
{
#pragma unroll
    for (int m = 0; m < OWN; ++m) {
        const int gi = T.gi(act[m], pr[m], pc[m]);
        if (PHILOX) {
            philox_normal4(a.seed, a.patch_base + T.patch_id, (uint32_t)gi, NF_STREAM_SAMP, z[m]);
#pragma unroll
            for (int q = 0; q < 4; ++q) z[m][q] *= a.in_scale;
        } else {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act[m]) v = reinterpret_cast<const float4 *>(a.in + T.patch_off)[gi];
            z[m][0] = v.x * a.in_scale;
            z[m][1] = v.y * a.in_scale;
            z[m][2] = v.z * a.in_scale;
            z[m][3] = v.w * a.in_scale;
        }
    }
}

// Conv2d1x1 (and whatever was folded into it): z <- z @ M, M = P[0..15] row-major (wave-uniform scalar loads)
template <int OWN>
__device__ __forceinline__ void gemm_mix(cfloat_p P, float (&z)[OWN][4])
{
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
}

// AffineCouplingSdnEx5 and its relatives: scale = sqrt(beta1*y/gain + beta2)  (cond_utils.py:238)
template <int OWN>
__device__ __forceinline__ void gemm_sdn(int type, int slot, const NfLaunch &a, const GemmTile &T, const int (&pr)[OWN], const int (&pc)[OWN],
                                         const bool (&act)[OWN], float (&z)[OWN][4], float &ld)
{
    const float4 *y4 = reinterpret_cast<const float4 *>(a.y + T.patch_off);
    const float ck1 = a.cond_a[slot & 3], cb2 = a.cond_b[slot & 3];
#pragma unroll
    for (int m = 0; m < OWN; ++m) {
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (act[m]) yv = y4[T.gi(act[m], pr[m], pc[m])];
        const bool own = T.own(act[m], pr[m], pc[m]);
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
}

template <int OWN>
__device__ __forceinline__ void gemm_scale(float s, float (&z)[OWN][4])
{
#pragma unroll
    for (int m = 0; m < OWN; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) z[m][q] *= s;
}

// The affine half of a coupling behind its CNN: o = the 4 raw outputs of l_last per owned pixel (without the border-table /
// bias entry), etab = the coupling's 16 x 4 border table, scl / m2scl = scale log2(e), -2 scale log2(e).
//   HALF   fp16-CNN layouts (the raw columns are pre-scaled by 2 log2(e) there too: inside the rounded weights, nf_host.hip::to_half_w3)
template <int OWN, bool HALF>
__device__ __forceinline__ void gemm_finish_coupling(int type, const float *__restrict__ etab, float scl, float m2scl, const GemmTile &T,
                                                     const int (&pr)[OWN], const int (&pc)[OWN], const bool (&act)[OWN], float (&o)[OWN][4],
                                                     float (&z)[OWN][4], float &ld2)
{
#pragma unroll
    for (int m = 0; m < OWN; ++m) {
        const float4 eb = *reinterpret_cast<const float4 *>(etab + 4 * (act[m] ? T.border(pr[m], pc[m]) : 0));
        o[m][0] += eb.x; o[m][1] += eb.y;
        o[m][2] += eb.z; o[m][3] += eb.w;
        // raw columns pre-scaled by 2 log2(e):  t = exp2(raw') = exp(2 raw);
        // ls*log2(e) = scl*tanh(raw) = scl - 2 scl/(t + 1); log-det accumulated in log2 units
        const float l0 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[m][2]) + 1.0f), m2scl, scl);
        const float l1 = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(o[m][3]) + 1.0f), m2scl, scl);
        if (type == NF_OP_COUPLING_FWD) {
            z[m][2] = fmaf(z[m][2], __builtin_amdgcn_exp2f(l0), o[m][0]);
            z[m][3] = fmaf(z[m][3], __builtin_amdgcn_exp2f(l1), o[m][1]);
            if (T.own(act[m], pr[m], pc[m])) ld2 += l0 + l1;
        } else {
            z[m][2] = (z[m][2] - o[m][0]) * __builtin_amdgcn_exp2f(-l0);
            z[m][3] = (z[m][3] - o[m][1]) * __builtin_amdgcn_exp2f(-l1);
        }
    }
}

// ---- epilogue (as nf_flow_kernel): outputs, per-patch nll / sd_z / log-det or, tiled, the tile's share of its image's sums ----
//   red   [3][GW] floats of LDS;  every thread of the workgroup calls this (two barriers)
template <int OWN, int GT>
__device__ __forceinline__ void gemm_epilogue(const NfLaunch &a, const GemmTile &T, int64_t b, int HW, const int (&pr)[OWN], const int (&pc)[OWN],
                                              const bool (&act)[OWN], const float (&z)[OWN][4], float ld, float ld2, float *red, double &acc_nll,
                                              double &acc_sd)
{
    constexpr int GW = GT / 64;
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
    if (a.out) {
        float4 *out4 = reinterpret_cast<float4 *>(a.out + T.patch_off);
#pragma unroll
        for (int m = 0; m < OWN; ++m)
            if (T.own(act[m], pr[m], pc[m])) out4[T.gi(act[m], pr[m], pc[m])] = make_float4(z[m][0], z[m][1], z[m][2], z[m][3]);
    }
    if (a.nll_out || a.sd_out || a.ld_out || a.sums || (T.tiled && a.tile_part)) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < OWN; ++m)
            if (T.own(act[m], pr[m], pc[m])) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s1 += z[m][q];
                    s2 = fmaf(z[m][q], z[m][q], s2);
                }
            }
        float r0 = wave_sum(fmaf(ld2, 0.6931471805599453f, ld)), r1 = wave_sum(s1), r2 = wave_sum(s2);
        if (lane == 0) {
            red[wv] = r0;
            red[GW + wv] = r1;
            red[2 * GW + wv] = r2;
        }
        __syncthreads();
        if (t == 0) {
            r0 = 0.f; r1 = 0.f; r2 = 0.f;
#pragma unroll
            for (int i = 0; i < GW; ++i) {
                r0 += red[i];
                r1 += red[GW + i];
                r2 += red[2 * GW + i];
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
            const double sd = sqrt(var);
            if (a.nll_out) a.nll_out[b] = (float)nll;
            if (a.sd_out) a.sd_out[b] = (float)sd;
            if (a.ld_out) a.ld_out[b] = (float)logdet;
            acc_nll += (double)(float)nll;
            acc_sd += (double)(float)sd;
        }
        __syncthreads();   // scratch is reused by the next patch
    }
}

// the workgroup's share of the call's batch sums (thread 0)
__device__ __forceinline__ void gemm_flush_sums(const NfLaunch &a, double acc_nll, double acc_sd)
{
    if (a.sums && threadIdx.x == 0 && !(a.flags & NF_K_TILED)) {
        double *sp = a.sums;
        if (a.flags & NF_K_SUMS_WIDE) sp += (size_t)(blockIdx.x & (NF_SUMS_SLOTS - 1)) * NF_SUMS_STRIDE;
        atomicAdd(&sp[0], acc_nll);
        atomicAdd(&sp[1], acc_sd);
        if (blockIdx.x == 0) atomicAdd(&sp[2], (double)a.B);
    }
}

// ---- launching (host) ------------------------------------------------------------------------------------------------------------
// What nf_gemm.hip and nf_gemm16.hip share around their kernels: one persistent workgroup of GT threads per CU (the band / slab
// images take most of a CU's LDS), dynamic LDS beyond 64 KiB opted into once per device and kernel instantiation (`lds_set`: the
// caller's static per-instantiation record of the largest size enabled so far; racy but idempotent), and the choice of pixels
// per thread by patch size.
template <int GT, typename K>
inline hipError_t gemm_launch_per_cu(K kern, size_t lds, std::atomic<size_t> (&lds_set)[16], const NfProgram &prog, const NfLaunch &a, int n_cu,
                                     int device, hipStream_t stream)
{
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    std::atomic<size_t> &cur = lds_set[device & 15];
    if (lds > cur.load(std::memory_order_relaxed) || device > 15) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        cur.store(lds, std::memory_order_relaxed);
    }
    int64_t groups = n_cu;
    if (a.B < groups) groups = a.B;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(GT), lds, stream, prog, a);
    return hipGetLastError();
}

// f(std::integral_constant<int, OWN>) with OWN = pixels per thread for a patch of hw pixels (2, 4 or 8: up to 64 x 64)
template <int GT, typename F>
inline hipError_t gemm_by_own(int hw, F &&f)
{
    if (hw <= 2 * GT) return f(std::integral_constant<int, 2>{});
    if (hw <= 4 * GT) return f(std::integral_constant<int, 4>{});
    return f(std::integral_constant<int, 8>{});
}

}  // namespace
