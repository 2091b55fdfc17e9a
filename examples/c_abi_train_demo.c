/*
 * Standalone C client of the training entry points of the C ABI (no Python, no torch): builds a
 * trainer from a raw parameter file, runs K Adam steps on synthetic camera-NLF patches and writes the
 * trained raw parameters back.
 *
 *   gcc -std=c99 examples/c_abi_train_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lnoise_flow_amd/csrc -lnoiseflow_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/noise_flow_amd/csrc -Wl,-rpath,/opt/rocm/lib -o c_abi_train_demo
 *   ./c_abi_train_demo model.bin B K lr out.bin      # model.bin as in examples/c_abi_demo.c
 *
 * Output (stdout): one line per step "step <k> loss <loss> sd_z <sd_z>"; out.bin = n_params floats.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include "noiseflow_hip.h"

#define CHECK_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)
#define CHECK_NF(e) do { int _r = (e); if (_r != NF_OK) { fprintf(stderr, "noiseflow error %d: %s (line %d)\n", _r, nf_last_error(), __LINE__); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s model.bin B K lr out.bin\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("model"); return 1; }
    const long B = atol(argv[2]), K = atol(argv[3]);
    const float lr = (float)atof(argv[4]);
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
    nf_trainer *t = NULL;
    CHECK_NF(nf_trainer_create(&cfg, layers, params, (size_t)n_params, B, NF_OPT_ADAM, &t));

    const size_t n = (size_t)B * 32 * 32 * 4;
    float *x, *y, *loss_d, loss_h[2];
    CHECK_HIP(hipMalloc((void **)&x, n * 4));
    CHECK_HIP(hipMalloc((void **)&y, n * 4));
    CHECK_HIP(hipMalloc((void **)&loss_d, 2 * sizeof(float)));
    nf_cond cond = {800.0f, 2.0f, 0.0f, 0.0f};
    for (long k = 0; k < K; ++k) {
        /* a fresh minibatch per step: patches [k*B, (k+1)*B) of the S6 / ISO-800 camera NLF */
        CHECK_NF(nf_synth_patches(11, k * B, B, 32, 32, 0.003696f, 0.000002f, y, x, NULL));
        CHECK_NF(nf_trainer_step(t, x, y, B, &cond, lr, loss_d, NULL));
        CHECK_HIP(hipMemcpy(loss_h, loss_d, sizeof(loss_h), hipMemcpyDeviceToHost));
        printf("step %ld loss %.9g sd_z %.9g\n", k, (double)loss_h[0], (double)loss_h[1]);
    }
    if (nf_trainer_steps(t) != K) { fprintf(stderr, "step counter %lld != %ld\n", (long long)nf_trainer_steps(t), K); return 4; }
    CHECK_NF(nf_trainer_get_params(t, params, (size_t)n_params, NULL));
    f = fopen(argv[5], "wb");
    if (!f || fwrite(params, 4, n_params, f) != (size_t)n_params) { perror("out"); return 1; }
    fclose(f);
    /* error behaviour: batch larger than the workspace */
    int rc = nf_trainer_step(t, x, y, B + 1, &cond, lr, loss_d, NULL);
    printf("oversized_batch %d %s\n", rc, rc == NF_EINVAL ? "NF_EINVAL" : "?");
    CHECK_NF(nf_trainer_destroy(t));
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(loss_d);
    free(layers); free(params);
    return 0;
}
