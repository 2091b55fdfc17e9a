// How much does interleaving fp32 VALU work with v_mfma_f32_4x4x1 bursts cost, as a function of
// the burst length and of s_setprio, with 4 waves per SIMD running unsynchronised?
// Per iteration every wave issues N MFMAs (4 chains) then 4N v_fma (same pipe time: 8N cycles each).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int N, int PRIO>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    v4f c[4];
    float a[8];
    for (int i = 0; i < 4; ++i) c[i] = v4f{(float)i, 0, 0, (float)threadIdx.x};
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    // de-synchronise the waves of the SIMD
    for (int i = 0; i < (int)((threadIdx.x >> 6) + blockIdx.x % 7) * 13; ++i) a[0] = fmaf(a[0], s, 1.0f);
    for (int it = 0; it < iters; ++it) {
        if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int r = 0; r < N / 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(s, 1.0f + i, c[i], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int r = 0; r < N / 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(s));
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i];
    for (int i = 0; i < 4; ++i) r += c[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int N, int PRIO> void run(float *d)
{
    const int blocks = 1024, iters = 200000 / N;   // same total work for every N
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<N, PRIO><<<blocks, 256>>>(d, iters, 0.999f);
    (void)hipEventRecord(e0);
    k<N, PRIO><<<blocks, 256>>>(d, iters, 0.999f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // ideal pipe time per SIMD: 4 waves * iters * (N*8 + 4N*2.0) cycles
    printf("burst N=%3d prio=%d: %.3f ms\n", N, PRIO, ms);
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * 4);
    run<4, 0>(d); run<16, 0>(d); run<64, 0>(d); run<256, 0>(d);
    run<4, 1>(d); run<16, 1>(d); run<64, 1>(d); run<256, 1>(d);
    return 0;
}
