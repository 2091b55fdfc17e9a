// Device helpers shared by the fused flow kernels (nf_kernels.hip, nf_wide.hip): Philox4x32-10 /
// Box-Muller, the hardware-transcendental exp / tanh, one-instruction ReLU, wavefront reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nf_device.h"

namespace {

// Model parameters are immutable for the lifetime of a handle: address them through
// the constant address space so that every wave-uniform fetch becomes an s_load
// (scalar cache, SGPR operands) instead of a per-lane global_load into VGPRs.
typedef const float __attribute__((address_space(4))) *cfloat_p;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// --------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al. 2011) — keyed by
// (seed, patch index, pixel, stream) so that data are identical for any sharding.
// --------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

__device__ __forceinline__ uint4 philox_pixel(uint64_t seed, int64_t patch, uint32_t pixel, uint32_t stream)
{
    const uint64_t k = (uint64_t)patch;
    return philox4x32_10(make_uint4((uint32_t)k, (uint32_t)(k >> 32), pixel, stream),
                         make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

// u32 -> U[0,1) with 24 bits (exact in fp32)
__device__ __forceinline__ float u01_24(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }
// u32 -> U(0,1) with 23 bits, never 0 (exact in fp32)
__device__ __forceinline__ float u01_open(uint32_t r) { return (float)(r >> 9) * 1.1920928955078125e-7f + 5.9604644775390625e-8f; }

// Box-Muller: two uniforms -> two N(0,1), on the hardware transcendental unit:
// r = sqrt(-2 ln u1) via v_log_f32 (log2) + v_sqrt_f32; sin/cos(2 pi u2) via v_sin_f32 / v_cos_f32,
// whose argument is in revolutions.  ~1e-6 absolute agreement with the libm formulation
// (oracle/philox.py); the u32 stream itself is bit-exact.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &n0, float &n1)
{
    const float u1 = u01_open(a), u2 = u01_open(b);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ void philox_normal4(uint64_t seed, int64_t patch, uint32_t pixel, uint32_t stream, float v[4])
{
    const uint4 r = philox_pixel(seed, patch, pixel, stream);
    box_muller(r.x, r.y, v[0], v[1]);
    box_muller(r.z, r.w, v[2], v[3]);
}

// exp / tanh on the hardware transcendental unit (v_exp_f32, v_rcp_f32: 1 ulp each).
// |ls| <= rescaling_scale < 1, so exp(ls) carries ~2e-7 relative error; tanh has
// ~1e-7 ABSOLUTE error (it only ever enters as scale*tanh(raw)), both far inside
// the 1e-5 parity budget (tests/test_gpu_parity.py measures it).
__device__ __forceinline__ float nf_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float nf_tanh(float x)
{
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // exp(2x); inf/0 saturate correctly
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// ReLU as ONE instruction: signed-integer max of the bit pattern with 0 (negative floats,
// including -0.0, have the sign bit set).  fmaxf()/v_med3 cost two because the compiler
// canonicalises the MFMA result first; inline asm is not an option because hipcc pads no
// VALU->MFMA hazard wait states around an asm statement.
__device__ __forceinline__ float nf_relu(float x)
{
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

// Wavefront sum, the total in every lane: four DPP adds finish the 16-lane rows (xor 1 / xor 2 inside the quads, the
// half-row and row mirrors), four v_readlane + three adds join the rows — ~11 issues instead of the six dependent
// ds_bpermute round trips of a __shfl_xor butterfly.
__device__ __forceinline__ float wave_sum(float v)
{
#define NF_WS_DPP(x, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (CTRL), 0xf, 0xf, false))
    v += NF_WS_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
    v += NF_WS_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
    v += NF_WS_DPP(v, 0x141);   // row_half_mirror
    v += NF_WS_DPP(v, 0x140);   // row_mirror
#undef NF_WS_DPP
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// every lane of a 16-lane DPP row gets its row's sum (the first four issues of wave_sum)
__device__ __forceinline__ float row16_sum(float v)
{
#define NF_WS_DPP(x, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (CTRL), 0xf, 0xf, false))
    v += NF_WS_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
    v += NF_WS_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
    v += NF_WS_DPP(v, 0x141);   // row_half_mirror
    v += NF_WS_DPP(v, 0x140);   // row_mirror
#undef NF_WS_DPP
    return v;
}

// the same for a double (both halves travel through the DPP / readlane network)
template <int CTRL>
__device__ __forceinline__ double dpp_get_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_get_f64<0xB1>(v);
    v += dpp_get_f64<0x4E>(v);
    v += dpp_get_f64<0x141>(v);
    v += dpp_get_f64<0x140>(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k), __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

}  // namespace
