/*
 * Standalone C client of the C ABI (no Python, no torch): builds a model from a raw
 * parameter file, evaluates the NLL of synthetic patches and synthesises noise.
 *
 *   gcc -std=c99 examples/c_abi_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lnoise_flow_amd/csrc -lnoiseflow_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/noise_flow_amd/csrc -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 *   ./c_abi_demo model.bin B        # model.bin written by tests/test_gpu_c_abi.py
 *
 * model.bin: int32 n_layers, then n_layers x {int32 type, int32 width, int64 offset},
 *            int64 n_params, then n_params floats  (the layout of include/noiseflow_hip.h).
 * Output (stdout): one line per patch "nll sd", then "sums <sum_nll> <sum_sd> <count>",
 *                  then "sample_checksum <sum of x>" for a Philox-seeded sampling call.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "noiseflow_hip.h"

#define CHECK_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)
#define CHECK_NF(e) do { int _r = (e); if (_r != NF_OK) { fprintf(stderr, "noiseflow error %d: %s (line %d)\n", _r, nf_last_error(), __LINE__); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s model.bin B\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("model"); return 1; }
    const long B = atol(argv[2]);
    int32_t n_layers;
    if (fread(&n_layers, 4, 1, f) != 1) return 1;
    nf_layer_desc *layers = (nf_layer_desc *)malloc(sizeof(nf_layer_desc) * n_layers);
    for (int i = 0; i < n_layers; ++i) {
        if (fread(&layers[i].type, 4, 1, f) != 1 || fread(&layers[i].width, 4, 1, f) != 1 ||
            fread(&layers[i].param_offset, 8, 1, f) != 1) return 1;
    }
    int64_t n_params;
    if (fread(&n_params, 8, 1, f) != 1) return 1;
    float *params = (float *)malloc(sizeof(float) * n_params);
    if (fread(params, 4, n_params, f) != (size_t)n_params) return 1;
    fclose(f);

    nf_config cfg = {32, 32, 4, n_layers, -1, 0};
    nf_handle *h = NULL;
    CHECK_NF(nf_create(&cfg, layers, params, (size_t)n_params, &h));

    const size_t n = (size_t)B * 32 * 32 * 4;
    float *x, *y, *xs, *nll, *sd;
    double *sums;
    CHECK_HIP(hipMalloc((void **)&x, n * 4));
    CHECK_HIP(hipMalloc((void **)&y, n * 4));
    CHECK_HIP(hipMalloc((void **)&xs, n * 4));
    CHECK_HIP(hipMalloc((void **)&nll, B * 4));
    CHECK_HIP(hipMalloc((void **)&sd, B * 4));
    CHECK_HIP(hipMalloc((void **)&sums, 3 * sizeof(double)));
    CHECK_NF(nf_synth_patches(7, 0, B, 32, 32, 0.000479f, 0.000002f, y, x, NULL));
    nf_cond cond = {100.0f, 2.0f, 0.0f, 0.0f};
    CHECK_NF(nf_nll(h, x, y, B, &cond, nll, sd, NULL, NULL, sums, 0, NULL));
    CHECK_NF(nf_sample(h, y, NULL, 99, 0, 0.6f, B, &cond, xs, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    float *hn = (float *)malloc(B * 4), *hs = (float *)malloc(B * 4), *hx = (float *)malloc(n * 4);
    double hsum[3];
    CHECK_HIP(hipMemcpy(hn, nll, B * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hs, sd, B * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hx, xs, n * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hsum, sums, sizeof(hsum), hipMemcpyDeviceToHost));
    for (long b = 0; b < B; ++b) printf("%.9g %.9g\n", hn[b], hs[b]);
    printf("sums %.17g %.17g %.17g\n", hsum[0], hsum[1], hsum[2]);
    double cs = 0.0;
    for (size_t i = 0; i < n; ++i) cs += hx[i];
    printf("sample_checksum %.17g\n", cs);

    /* error path: unknown camera id */
    nf_cond bad = {100.0f, 9.0f, 0.0f, 0.0f};
    const int rc = nf_nll(h, x, y, B, &bad, nll, NULL, NULL, NULL, NULL, 0, NULL);
    printf("bad_cam %d %s\n", rc, rc == NF_ECOND ? "NF_ECOND" : "unexpected");
    CHECK_NF(nf_destroy(h));
    return 0;
}
