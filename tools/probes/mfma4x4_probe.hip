// Probe the operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950.
// a(lane) = 100 + lane, b(lane) = 1000*(lane+1): D_blk[i][j] = a_i * b_j tells us who is who.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(float *out)
{
    const int l = threadIdx.x;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 + l), (float)(1000 * (l + 1)), c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = d[v];
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_hyp = 1;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            // hypothesis: lane (blk, j), vgpr v=i -> A from lane 4*blk+i, B from lane 4*blk+j
            const int blk = l / 4, j = l % 4, i = v;
            const float want = (float)(100 + 4 * blk + i) * (float)(1000 * (4 * blk + j + 1));
            if (h[l * 4 + v] != want) ok_hyp = 0;
        }
    printf("hypothesis D[lane=(blk,j)][vgpr=i] = A[lane (blk,i)] * B[lane (blk,j)] : %s\n", ok_hyp ? "CONFIRMED" : "WRONG");
    for (int l = 0; l < 8; ++l) printf("lane %d: %.0f %.0f %.0f %.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
