// Do fp32 MFMA (4x4x1 16-block) and fp32 VALU FMAs overlap on one gfx950 SIMD?
// A: VALU only, B: MFMA only, C: both (same instruction counts, independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v4f c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
    for (int i = 0; i < iters; ++i) {
        if (MODE & 1) {   // 32 independent-ish VALU FMAs (8 chains x 4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = fmaf(a0, s, 1.0f); a1 = fmaf(a1, s, 1.0f); a2 = fmaf(a2, s, 1.0f); a3 = fmaf(a3, s, 1.0f);
                a4 = fmaf(a4, s, 1.0f); a5 = fmaf(a5, s, 1.0f); a6 = fmaf(a6, s, 1.0f); a7 = fmaf(a7, s, 1.0f);
            }
        }
        if (MODE & 2) {   // 8 MFMAs (4 chains x 2) = 8*8 = 64 pipe cycles vs 32 FMAs * 2 = 64 cycles
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, a0 * 0 + 1.0f, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, 1.0f, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, 1.0f, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, 1.0f, c3, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3];
}
template <int MODE> float run(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    for (int wpc = 1; wpc <= 4; wpc *= 2) {   // workgroups per CU (x4 waves each => waves per SIMD)
        int blocks = 256 * wpc, iters = 20000;
        float a = run<1>(d, blocks, iters), b = run<2>(d, blocks, iters), c = run<3>(d, blocks, iters);
        // per SIMD per iteration cycles at ~2.2 GHz
        double cyc = 2.2e6 / (double)iters / wpc;
        printf("waves/SIMD=%d  VALU %.3f ms (%.1f cyc/iter/wave)  MFMA %.3f ms (%.1f)  both %.3f ms (%.1f)  sum=%.3f max=%.3f\n", wpc,
               a, a * cyc, b, b * cyc, c, c * cyc, a + b, a > b ? a : b);
    }
    return 0;
}
