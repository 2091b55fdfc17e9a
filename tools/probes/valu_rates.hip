// Issue-rate probe for the instructions the fused kernel is built from (gfx950).
// Every kernel runs `iters` x 32 independent instructions per wave, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s, float u)
{
    float a[16];
    v2f p[8];
    v4f c[4];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
    for (int i = 0; i < 8; ++i) p[i] = v2f{(float)threadIdx.x + i, (float)i};
    for (int i = 0; i < 4; ++i) c[i] = v4f{0, 0, 0, 0};
    const v2f sv = {s, u};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 32 x v_fma_f32, SGPR multiplier
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(u));
        } else if (MODE == 1) {   // 16 x v_pk_fma_f32 (= 32 FMAs), SGPR-pair multiplier
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "s"(sv));
        } else if (MODE == 2) {   // 32 x v_fmac_f32 VGPR operands
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(u), "v"(s));
        } else if (MODE == 3) {   // 32 x v_exp_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 4) {   // 32 x v_rcp_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 5) {   // 32 x MFMA 4x4x1, 4 independent chains (builtin: compiler adds hazards nops)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(s, u, c[i], 0, 0, 0);
        } else if (MODE == 6) {   // 32 x v_max_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(u));
        } else if (MODE == 7) {   // 32 x v_pk_mul_f32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sv));
        } else if (MODE == 8) {   // 16 v_fma + 16 v_exp interleaved
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(u));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 8) & 15]));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1];
    for (int i = 0; i < 4; ++i) r += c[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(float *d, const char *name, int n_inst)
{
    const int blocks = 256 * 4, iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0.5f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per SIMD: cycles per instruction per SIMD at 2.2 GHz (approx; DVFS moves it)
    printf("%-34s %.3f ms  -> %.2f cyc / wave-instruction (at 2.2 GHz)\n", name, ms, ms * 2.2e6 / (double)iters / 4.0 / n_inst);
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    run<0>(d, "v_fma_f32 (sgpr operand)", 32);
    run<1>(d, "v_pk_fma_f32 (2 FMA/lane)", 16);
    run<2>(d, "v_fmac_f32 (vgpr operands)", 32);
    run<3>(d, "v_exp_f32", 32);
    run<4>(d, "v_rcp_f32", 32);
    run<5>(d, "v_mfma_f32_4x4x1_16b (4 chains)", 32);
    run<6>(d, "v_max_f32", 32);
    run<7>(d, "v_pk_mul_f32", 32);
    run<8>(d, "v_fma + v_exp interleaved", 32);
    return 0;
}
