// (1) What does __builtin_readcyclecounter() (s_memtime) tick at relative to the 100 MHz wall clock?
// (2) Issue interval of DEPENDENT v_mfma_f32_4x4x1 (one accumulator chain) vs 4 independent chains.
// Measured on MI355X: counter = shader clock (2.2-2.4 GHz); one chain: ~52-56 cycles per MFMA
// (dependent-issue latency); pipe occupancy of one MFMA is 8 cycles (2 passes).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void k(unsigned long long *out, int iters, float s)
{
    v4f c[4];
    for (int i = 0; i < 4; ++i) c[i] = v4f{(float)i, 0, 0, (float)threadIdx.x};   // distinct: no CSE of the chains
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(s, 1.0f + i, c[i], 0, 0, 0);
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (c[0][0] + c[1][0] + c[2][0] + c[3][0] == 12345.f) out[2] = 1;
}
template <int CHAINS> void run(unsigned long long *d, int blocks)
{
    unsigned long long h[3];
    const int iters = 100000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<CHAINS><<<blocks, 256>>>(d, iters, 0.5f);
    (void)hipEventRecord(e0);
    k<CHAINS><<<blocks, 256>>>(d, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("[host: kernel %.3f ms = %.1f cyc per MFMA per SIMD at the counter clock] ", ms,
           ms * 1e-3 * (100.0e6 * (double)h[0] / (double)h[1]) / ((double)iters * CHAINS * (blocks / 256)));
    printf("chains/wave=%d waves/SIMD=%d: counter %.0f MHz; %.1f cycles per MFMA per wave -> %.1f cycles per MFMA per SIMD\n", CHAINS,
           blocks / 256, 100.0 * (double)h[0] / (double)h[1], (double)h[0] / iters / CHAINS, (double)h[0] / iters / CHAINS / (blocks / 256));
}
int main()
{
    unsigned long long *d;
    (void)hipMalloc(&d, 64);
    for (int blocks : {256, 512, 1024, 2048}) { run<1>(d, blocks); run<2>(d, blocks); run<4>(d, blocks); }
    return 0;
}
