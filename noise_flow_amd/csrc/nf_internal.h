// Host-side helpers shared by the translation units of libnoiseflow_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// record a thread-local error message (returned by nf_last_error) and hand back `code`
int nf_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int nf_fail_hip(hipError_t e, const char *what);

// geometry / device of a handle (nf_hostfed.hip sizes its staging from them)
struct nf_handle;
int nf_handle_geometry(const nf_handle *h, int32_t *H, int32_t *W, int32_t *device);
// frees the handle's host-fed pipeline, if one was created (nf_destroy)
void nf_hostpipe_release(nf_handle *h);
