// Stand-alone check + timing of the trainer's matrix-core GEMMs (noise_flow_amd/csrc/nf_train_mm.h) against fp64 CPU sums.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form tools/probes/mm_probe.hip -o tools/probes/mm_probe && tools/probes/mm_probe [time]
// Every (prologue, epilogue, vector width, tile) combination the trainer launches is run at a ragged shape; `time` adds the
// width-512 shapes of a 138-patch step (TFLOP/s of each product).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../noise_flow_amd/csrc/nf_train_mm.h"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

static uint32_t rs = 12345;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f; }
static std::vector<float> rvec(size_t n, float sc = 1.0f) { std::vector<float> v(n); for (auto &x : v) x = frand() * sc; return v; }
template <typename T> static T *dev(const std::vector<T> &v) { T *p; CK(hipMalloc(&p, std::max<size_t>(v.size(), 4) * sizeof(T))); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }
static std::vector<float> host(const float *p, size_t n) { std::vector<float> v(n); CK(hipMemcpy(v.data(), p, n * sizeof(float), hipMemcpyDeviceToHost)); return v; }

static int fails = 0;
static mm::Ctx cx{256, 0};

static float hxhat(float h, float b, float m, float r) { return fmaf(h, r, (b - m) * r); }

template <int APRO, int EPI, int AV>
static void check_pix(int64_t P, int N, int K, int nslot)
{
    const int lda = K, ldb = (K + 3) & ~3, ldc = N;
    auto A = rvec((size_t)P * lda), W = rvec((size_t)N * K, 0.3f), ab = rvec(K, 0.2f), am = rvec(K, 0.2f), ar = rvec(K), eb = rvec(N, 0.2f), em = rvec(N, 0.2f),
         er = rvec(N), H = rvec((size_t)P * N), A2 = rvec((size_t)P * lda), abb = rvec(2 * (size_t)K, 0.1f), ebb = rvec(2 * (size_t)N, 0.1f);
    for (auto &x : ar) x = 0.5f + fabsf(x);
    for (auto &x : er) x = 0.5f + fabsf(x);
    std::vector<float> abn(2 * K), ebn(2 * N), Bt((size_t)N * ldb, 0.0f);
    for (int k = 0; k < K; ++k) { abn[k] = am[k]; abn[K + k] = ar[k]; }
    for (int n = 0; n < N; ++n) { ebn[n] = em[n]; ebn[N + n] = er[n]; }
    float *dW = dev(W), *dBt = dev(Bt);
    const int ld = mm::mm_pack(0, 0, N, K, 0, dW, dBt);
    if (ld != ldb) { printf("pack ld\n"); ++fails; }
    float *dA = dev(A), *dC, *dab = dev(ab), *dabn = dev(abn), *deb = dev(eb), *debn = dev(ebn), *dH = dev(H), *dS, *dA2 = dev(A2), *dabb = dev(abb), *debb = dev(ebb);
    CK(hipMalloc(&dC, (size_t)P * ldc * sizeof(float)));
    CK(hipMemset(dC, 0, (size_t)P * ldc * sizeof(float)));
    CK(hipMalloc(&dS, (size_t)2 * N * mm::kSlotStride * sizeof(float)));
    CK(hipMemset(dS, 0x7f, (size_t)2 * N * mm::kSlotStride * sizeof(float)));   // stale slots must be overwritten
    mm::PixArgs a{};
    a.P = P; a.N = N; a.K = K; a.A = dA; a.lda = lda; a.Bt = dBt; a.ldb = ldb; a.C = dC; a.ldc = ldc;
    a.abias = dab; a.abn = dabn; a.ebias = deb; a.ebn = debn; a.eh = dH; a.ldh = N; a.stats = dS; a.nslot = nslot;
    a.A2 = dA2; a.abb = dabb; a.ebb = debb;
    if (!mm::mm_pix<APRO, EPI, AV>(cx, 0, a)) { printf("launch failed\n"); ++fails; return; }
    CK(hipDeviceSynchronize());
    auto C = host(dC, (size_t)P * ldc), S = host(dS, (size_t)2 * N * mm::kSlotStride);
    // reference
    double worst = 0.0, scale = 0.0, sworst = 0.0;
    std::vector<double> s1(N, 0.0), s2(N, 0.0), a1(N, 0.0), a2(N, 0.0);
    std::vector<float> Ap((size_t)K);
    for (int64_t p = 0; p < P; ++p) {
        for (int k = 0; k < K; ++k) {
            float v = A[p * lda + k];
            if (APRO == 1) v = fmaxf(hxhat(v, ab[k], am[k], ar[k]), 0.0f);
            if (APRO == 2) { const float xh = hxhat(A2[p * lda + k], ab[k], am[k], ar[k]); v = ar[k] * ((xh > 0.f ? v : 0.f) - abb[k] - xh * abb[K + k]); }
            Ap[k] = v;
        }
        for (int n = 0; n < N; ++n) {
            double acc = 0.0, aa = 0.0;
            for (int k = 0; k < K; ++k) { acc += (double)Ap[k] * W[(size_t)n * K + k]; aa += fabs((double)Ap[k] * W[(size_t)n * K + k]); }
            double got = C[p * ldc + n];
            if (EPI == 4) { if (got != 0.0) worst = 1.0; got = acc; }       // nothing may be stored; the sums are checked against the exact product
            if (EPI == 3) {
                const float xh = hxhat(H[p * N + n], eb[n], em[n], er[n]);
                const double want = er[n] * ((xh > 0.f ? acc : 0.0) - ebb[n] - (double)xh * ebb[N + n]);
                worst = std::max(worst, fabs(got - want) / (er[n] * (aa + fabs(ebb[n]) + fabs(xh * ebb[N + n])) + 1e-30));
                s1[n] += got; a1[n] += fabs(got);
                scale = std::max(scale, fabs(acc));
                continue;
            }
            worst = std::max(worst, fabs(got - acc) / (aa + 1e-30));
            scale = std::max(scale, fabs(acc));
            if (EPI == 1) { const double x = (float)got + eb[n]; s1[n] += x; s2[n] += x * x; a1[n] += fabs(x); a2[n] += x * x; }
            if (EPI == 2 || EPI == 4) { const float xh = hxhat(H[p * N + n], eb[n], em[n], er[n]); const double gx = xh > 0.f ? got : 0.0; s1[n] += gx; s2[n] += gx * xh; a1[n] += fabs(gx); a2[n] += fabs(gx * xh); }
        }
    }
    if (EPI != 0)
        for (int n = 0; n < N; ++n) {
            double t1 = 0.0, t2 = 0.0;
            for (int s = 0; s < nslot; ++s) { t1 += S[(size_t)n * mm::kSlotStride + s]; t2 += S[(size_t)(N + n) * mm::kSlotStride + s]; }
            sworst = std::max(sworst, fabs(t1 - s1[n]) / (a1[n] + 1e-30));
            if (EPI != 3) sworst = std::max(sworst, fabs(t2 - s2[n]) / (a2[n] + 1e-30));
        }
    const bool ok = worst < 1e-5 && sworst < 1e-5 && scale > 0.01;
    printf("%s pix<APRO %d EPI %d AV %d> P=%lld N=%d K=%d nslot=%d: max err / sum|terms| %.2e, stats %.2e\n", ok ? "ok  " : "FAIL", APRO, EPI, AV, (long long)P, N, K,
           nslot, worst, sworst);
    if (!ok) ++fails;
    (void)hipFree(dW); (void)hipFree(dBt); (void)hipFree(dA); (void)hipFree(dC); (void)hipFree(dab); (void)hipFree(dabn); (void)hipFree(deb); (void)hipFree(debn); (void)hipFree(dH); (void)hipFree(dS); (void)hipFree(dA2); (void)hipFree(dabb); (void)hipFree(debb);
}

template <int WMv, int TM, int TN, int APRO, int AV, int BV, int BPRO = 0>
static void check_kpix(int64_t npix, int M, int N, int nslot = 1024)
{
    auto A = rvec((size_t)npix * M), B = rvec((size_t)npix * N), ab = rvec(M, 0.2f), am = rvec(M, 0.2f), ar = rvec(M);
    auto B2 = rvec((size_t)npix * N), bb = rvec(N, 0.2f), bm = rvec(N, 0.2f), br = rvec(N), bbb = rvec(2 * (size_t)N, 0.1f);
    for (auto &x : ar) x = 0.5f + fabsf(x);
    for (auto &x : br) x = 0.5f + fabsf(x);
    std::vector<float> bbn(2 * N);
    for (int k = 0; k < N; ++k) { bbn[k] = bm[k]; bbn[N + k] = br[k]; }
    float *dB2 = dev(B2), *dbb = dev(bb), *dbbn = dev(bbn), *dbbb = dev(bbb), *dD;
    CK(hipMalloc(&dD, (size_t)N * mm::kSlotStride * sizeof(float)));
    CK(hipMemset(dD, 0x7f, (size_t)N * mm::kSlotStride * sizeof(float)));
    std::vector<float> abn(2 * M);
    for (int k = 0; k < M; ++k) { abn[k] = am[k]; abn[M + k] = ar[k]; }
    float *dA = dev(A), *dB = dev(B), *dab = dev(ab), *dabn = dev(abn), *dP;
    CK(hipMalloc(&dP, (size_t)(mm::kGradPartFloats + 4 * (size_t)M * N) * sizeof(float)));
    mm::KpixArgs k{};
    k.npix = npix; k.M = M; k.N = N; k.A = dA; k.lda = M; k.B = dB; k.ldb = N; k.part = dP; k.abias = dab; k.abn = dabn;
    k.B2 = dB2; k.bbias = dbb; k.bbn = dbbn; k.bbb = dbbb; k.dbias = dD; k.nslot = nslot;
    const int S = mm::mm_kpix_launch<WMv, TM, TN, APRO, AV, BV, BPRO>(cx, 0, k);
    CK(hipDeviceSynchronize());
    constexpr int NA = APRO == 3 ? 2 : 1;
    auto part = host(dP, (size_t)S * NA * M * N);
    std::vector<double> ref((size_t)M * N, 0.0), mag((size_t)M * N, 0.0), ref2((size_t)M * N, 0.0);
    for (int64_t p = 0; p < npix; ++p)
        for (int m = 0; m < M; ++m) {
            float v = A[p * M + m];
            if (APRO == 1 || APRO == 3) v = fmaxf(hxhat(v, ab[m], am[m], ar[m]), 0.0f);
            if (APRO == 3 && v > 0.f)
                for (int n = 0; n < N; ++n) ref2[(size_t)m * N + n] += (double)B[p * N + n];
            for (int n = 0; n < N; ++n) {
                float bv = B[p * N + n];
                if (BPRO == 2) { const float xh = hxhat(B2[p * N + n], bb[n], bm[n], br[n]); bv = br[n] * ((xh > 0.f ? bv : 0.f) - bbb[n] - xh * bbb[N + n]); }
                ref[(size_t)m * N + n] += (double)v * bv; mag[(size_t)m * N + n] += fabs((double)v * bv);
            }
        }
    double worst = 0.0, dworst = 0.0;
    if (BPRO == 2) {
        auto D = host(dD, (size_t)N * mm::kSlotStride);
        for (int n = 0; n < N; ++n) {
            double want = 0.0, mg = 0.0, got = 0.0;
            for (int64_t p = 0; p < npix; ++p) { const float xh = hxhat(B2[p * N + n], bb[n], bm[n], br[n]); const double bv = br[n] * ((xh > 0.f ? B[p * N + n] : 0.f) - bbb[n] - xh * bbb[N + n]); want += bv; mg += fabs(bv); }
            for (int sl = 0; sl < nslot; ++sl) got += D[(size_t)n * mm::kSlotStride + sl];
            dworst = std::max(dworst, fabs(got - want) / (mg + 1e-30));
        }
        if (S > nslot) dworst = 1.0;
    }
    for (size_t e = 0; e < (size_t)M * N; ++e) {
        double t = 0.0, t2 = 0.0;
        for (int s = 0; s < S; ++s) {
            t += part[(size_t)s * NA * M * N + e];
            if (NA == 2) t2 += part[((size_t)s * NA + 1) * M * N + e];
        }
        worst = std::max(worst, fabs(t - ref[e]) / (mag[e] + 1e-30));
        if (NA == 2) worst = std::max(worst, fabs(t2 - ref2[e]) / (fabs(ref2[e]) + (double)npix * 0.5 * 1e-3 + 1e-30) * 1e-3);
    }
    const bool ok = worst < 1e-5 && S > 0 && dworst < 1e-5;
    printf("%s kpix<%d %d %d APRO %d AV %d BV %d BPRO %d> npix=%lld M=%d N=%d S=%d: max err / sum|terms| %.2e, d bias %.2e\n", ok ? "ok  " : "FAIL", WMv, TM, TN, APRO, AV, BV, BPRO,
           (long long)npix, M, N, S, worst, dworst);
    if (!ok) ++fails;
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dab); (void)hipFree(dabn); (void)hipFree(dP); (void)hipFree(dB2); (void)hipFree(dbb); (void)hipFree(dbbn); (void)hipFree(dbbb); (void)hipFree(dD);
}

static int only = -1;   // time <k>: just the k-th product of the list (for counter runs)
static void timing(int w, int64_t P)
{
    int idx = 0;
    const int nslot = w <= 128 ? 512 : 276;   // as the trainer at 138 patches (nf_train.hip: blocks_for)
    std::vector<float> z((size_t)P * w, 0.5f), wz((size_t)w * w, 0.01f), c(2 * w, 1.0f);
    float *dA = dev(z), *dH = dev(z), *dC, *dBt = dev(wz), *dc = dev(c), *dS, *dP, *dG = nullptr;
    CK(hipMalloc(&dC, (size_t)P * w * sizeof(float)));
    CK(hipMalloc(&dS, (size_t)2 * w * mm::kSlotStride * sizeof(float)));
    CK(hipMalloc(&dP, (size_t)(mm::kGradPartFloats + 2 * (size_t)w * w) * sizeof(float)));
    CK(hipMalloc(&dG, (size_t)P * 36 * sizeof(float)));
    CK(hipMemset(dG, 0, (size_t)P * 36 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *what, double flop, auto &&fn) {
        if (only >= 0 && only != idx++) return;
        for (int i = 0; i < 3; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("time w=%d P=%lld %-34s %8.1f us  %6.1f TFLOP/s (%.3f of 157.3)\n", w, (long long)P, what, ms * 1e3, flop / (ms * 1e-3) * 1e-12, flop / (ms * 1e-3) * 1e-12 / 157.3);
    };
    mm::PixArgs a{};
    a.P = P; a.N = w; a.K = w; a.A = dA; a.lda = w; a.Bt = dBt; a.ldb = w; a.C = dC; a.ldc = w;
    a.abias = dc; a.abn = dc; a.ebias = dc; a.ebn = dc; a.eh = dH; a.ldh = w; a.stats = dS; a.nslot = nslot;
    const double f2 = 2.0 * P * w * w;
    run("l_2 fwd  <APRO 1, EPI 1>", f2, [&] { mm::mm_pix<1, 1, 4>(cx, 0, a); });
    run("l_2 bwd  <APRO 0, EPI 2>", f2, [&] { mm::mm_pix<0, 2, 4>(cx, 0, a); });
    run("plain    <APRO 0, EPI 0>", f2, [&] { mm::mm_pix<0, 0, 4>(cx, 0, a); });
    mm::PixArgs b = a;
    b.N = 36; b.C = dG; b.ldc = 36;
    run("l_last fwd N=36 <APRO 1>", 2.0 * P * w * 36, [&] { mm::mm_pix<1, 0, 4>(cx, 0, b); });
    b = a; b.K = 36; b.A = dG; b.lda = 36; b.ldb = 36; b.ebb = dc;
    run("l_last bwd K=36 <EPI 4>", 2.0 * P * w * 36, [&] { mm::mm_pix<0, 4, 4>(cx, 0, b); });
    run("l_last bwd K=36 <EPI 3>", 2.0 * P * w * 36, [&] { mm::mm_pix<0, 3, 4>(cx, 0, b); });
    b = a; b.N = 18; b.C = dG; b.ldc = 18; b.A2 = dH; b.abb = dc;
    run("l_1 bwd N=18 <APRO 2>", 2.0 * P * w * 18, [&] { mm::mm_pix<2, 0, 4>(cx, 0, b); });
    b = a; b.K = 20; b.A = dG; b.lda = 20; b.ldb = 20;
    run("l_1 fwd K=20 <EPI 1>", 2.0 * P * w * 18, [&] { mm::mm_pix<0, 1, 4>(cx, 0, b); });
    mm::KpixArgs k{};
    k.npix = P; k.M = w; k.N = w; k.A = dA; k.lda = w; k.B = dH; k.ldb = w; k.part = dP; k.abias = dc; k.abn = dc;
    run("d l_2/W  kpix<2,2,2,APRO 1>", f2, [&] { mm::mm_kpix_launch<2, 2, 2, 1, 4, 4>(cx, 0, k); });
    k.N = 36; k.B = dG; k.ldb = 36;
    run("d l_last/W + mask kpix<2,2,1,APRO 3>", 2.0 * P * w * 36, [&] { mm::mm_kpix_launch<2, 2, 1, 3, 4, 4>(cx, 0, k); });
    k.M = 18; k.N = w; k.A = dG; k.lda = 20; k.B = dH; k.ldb = w; k.B2 = dA; k.bbias = dc; k.bbn = dc; k.bbb = dc; k.dbias = dS; k.nslot = nslot;
    run("d l_1/W  kpix<.,1,.,BPRO 2>", 2.0 * P * w * 18, [&] {   // the channel tile the trainer picks for the width
        if (w > 128) mm::mm_kpix_launch<1, 1, 2, 0, 1, 4, 2>(cx, 0, k);
        else if (w > 64) mm::mm_kpix_launch<1, 1, 1, 0, 1, 4, 2>(cx, 0, k);
        else mm::mm_kpix_launch<2, 1, 1, 0, 1, 4, 2>(cx, 0, k);
    });
    (void)hipFree(dA); (void)hipFree(dH); (void)hipFree(dC); (void)hipFree(dBt); (void)hipFree(dc); (void)hipFree(dS); (void)hipFree(dP); (void)hipFree(dG);
}

int main(int argc, char **argv)
{
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (cus > 0) cx.n_cu = cus;
    printf("CUs: %d\n", cx.n_cu);
    if (argc > 2 && !strcmp(argv[1], "time")) {   // mm_probe time <k> [width]: no checks, one product
        only = atoi(argv[2]);
        timing(argc > 3 ? atoi(argv[3]) : 512, (argc > 4 ? atoi(argv[4]) : 138) * 1024);
        return 0;
    }
    // k_mm_pix: every template combination of nf_train_gemm.h at ragged shapes (tile widths 128 / 64 / 32 via N)
    check_pix<1, 1, 4>(1000, 512, 512, 7);
    check_pix<1, 1, 4>(777, 96, 96, 3);
    check_pix<1, 1, 4>(300, 64, 64, 1);
    check_pix<1, 1, 4>(130, 24, 24, 2);
    check_pix<1, 1, 1>(333, 50, 50, 2);
    check_pix<1, 1, 1>(100, 5, 5, 1);
    check_pix<0, 1, 1>(500, 50, 18, 3);      // l_1 at a width that is not a multiple of 4
    check_pix<0, 1, 1>(500, 130, 18, 3);
    check_pix<1, 0, 4>(900, 36, 128, 1);     // l_last forward
    check_pix<1, 0, 1>(200, 36, 50, 1);
    check_pix<0, 2, 4>(640, 512, 36, 5);     // l_last transposed
    check_pix<0, 2, 4>(257, 50, 36, 2);
    check_pix<0, 2, 4>(700, 128, 128, 4);    // l_2 transposed
    check_pix<0, 2, 1>(123, 50, 50, 1);
    check_pix<2, 0, 4>(515, 18, 96, 1);      // l_1 transposed, BN1 backward formed in the staging
    check_pix<2, 0, 4>(700, 18, 512, 1);
    check_pix<2, 0, 1>(129, 18, 50, 1);
    check_pix<0, 4, 4>(640, 512, 36, 5);     // l_last transposed, pass 1 (sums only) and pass 2 (BN2 backward stored)
    check_pix<0, 3, 4>(640, 512, 36, 5);
    check_pix<0, 4, 4>(257, 50, 36, 2);
    check_pix<0, 3, 4>(257, 50, 36, 2);
    check_pix<0, 3, 4>(300, 24, 36, 1);
    check_pix<0, 1, 4>(500, 50, 20, 3);      // l_1 forward on Z18 rows of 20
    check_pix<0, 1, 4>(900, 512, 20, 4);
    // k_mm_kpix
    check_kpix<2, 2, 2, 1, 4, 4>(5000, 512, 512);
    check_kpix<2, 2, 2, 1, 4, 4>(1030, 96, 96);
    check_kpix<2, 2, 2, 1, 1, 1>(700, 70, 70);
    check_kpix<2, 1, 1, 1, 4, 4>(999, 64, 64);
    check_kpix<2, 1, 1, 1, 4, 4>(999, 24, 24);
    check_kpix<2, 1, 1, 1, 1, 1>(400, 50, 50);
    check_kpix<2, 2, 1, 3, 4, 4>(3000, 512, 36);      // d l_last/W and the mask product beside it
    check_kpix<2, 2, 1, 3, 1, 4>(800, 70, 36);
    check_kpix<2, 1, 1, 3, 4, 4>(800, 48, 36);
    check_kpix<2, 1, 1, 3, 1, 4>(300, 5, 36);
    check_kpix<1, 1, 2, 0, 1, 4, 2>(2000, 18, 512, 9);
    check_kpix<1, 1, 2, 0, 1, 4, 2>(600, 18, 48, 2);
    check_kpix<1, 1, 2, 0, 1, 1, 2>(600, 18, 50, 1024);
    check_kpix<1, 1, 1, 0, 1, 4, 2>(900, 18, 96, 7);
    check_kpix<1, 1, 1, 0, 1, 1, 2>(900, 18, 70, 7);
    check_kpix<2, 1, 1, 0, 1, 4, 2>(900, 18, 64, 5);
    check_kpix<2, 1, 1, 0, 1, 4, 2>(700, 18, 24, 300);
    check_kpix<2, 1, 1, 0, 1, 1, 2>(500, 18, 5, 3);
    printf("%s (%d failures)\n", fails ? "PROBE FAILED" : "PROBE OK", fails);
    if (argc > 1 && !strcmp(argv[1], "time")) {
        timing(512, 138 * 1024);
        timing(128, 138 * 1024);
        timing(64, 138 * 1024);
    }
    return fails ? 1 : 0;
}
