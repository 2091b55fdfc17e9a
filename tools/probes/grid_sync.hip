// Probe: latency of a grid-wide barrier (cooperative groups) vs a kernel boundary, MI355X.
// hipcc -O3 --offload-arch=gfx950 grid_sync.hip -o grid_sync && ./grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
namespace cg = cooperative_groups;

__global__ void k_sync(int n, float *buf)
{
    cg::grid_group g = cg::this_grid();
    float v = buf[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < n; ++i) {
        buf[blockIdx.x * blockDim.x + threadIdx.x] = v + 1.0f;   // something to publish
        g.sync();
        v = buf[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x];   // read a neighbour's value
    }
    buf[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

__global__ void k_one(float *buf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    buf[i] = buf[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x] + 1.0f;
}

int main()
{
    for (int blocks : {256, 512, 1024}) {
        float *buf;
        hipMalloc(&buf, blocks * 256 * sizeof(float));
        hipMemset(buf, 0, blocks * 256 * sizeof(float));
        int n = 200;
        void *args[] = {&n, &buf};
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipError_t e = hipLaunchCooperativeKernel((void *)k_sync, dim3(blocks), dim3(256), args, 0, 0);
        hipDeviceSynchronize();
        if (e != hipSuccess) { printf("blocks %d: cooperative launch failed: %s\n", blocks, hipGetErrorString(e)); continue; }
        hipEventRecord(e0);
        hipLaunchCooperativeKernel((void *)k_sync, dim3(blocks), dim3(256), args, 0, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %4d: grid.sync round = %.2f us\n", blocks, ms * 1e3 / n);
        hipEventRecord(e0);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_one, dim3(blocks), dim3(256), 0, 0, buf);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %4d: kernel boundary  = %.2f us\n", blocks, ms * 1e3 / n);
        hipFree(buf);
    }
    return 0;
}
