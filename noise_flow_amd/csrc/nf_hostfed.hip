// Host-fed entry points of the C ABI: nf_nll_host / nf_sample_host.
//
// The reference's callers hand HOST numpy arrays to sess.run — float64 minibatches from MiniBatchSampler
// (sidd/MiniBatchSampler.py:54-55, cast to float32 at the feed), `sample_noise_nf(batch_x, ...)` in
// NoiseFlowWrapper.py:81-87 — and get host arrays back.  Over PCIe that call is 15-30 x slower than the kernel, so the
// boundary, not the kernel, decides what a drop-in caller sees.  These entries take host pointers and run the call as a
// chunked pipeline owned by the handle:
//
//   caller memory --(worker threads: float64 -> float32 narrowing / copy)--> pinned staging --(H2D, stream s)-->
//   device chunk --(the fused kernel, stream s)--> device results --(D2H, stream s)--> pinned staging
//   --(worker threads)--> caller memory
//
// with three chunks in flight on as many HIP streams: while chunk c is narrowed by the host threads, chunk c-1 crosses PCIe
// and chunk c-2 is in the kernel.  Tensor RESULTS (z, x) take no copy at all: the kernel stores them straight into
// page-locked host memory — the caller's buffer when that is page-locked, the slot's staging otherwise — so that the way
// back (posted PCIe writes) runs concurrently with the next chunk's H2D DMA (sampling is full duplex).  Patches are independent in evaluation mode, so chunking changes no per-patch result: outputs are bit-identical to
// the device-resident nf_nll / nf_sample on the same data (the batch sums differ only in fp64 summation order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <immintrin.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/noiseflow_hip.h"
#include "nf_internal.h"

namespace {

constexpr int kSlots = 3;

// ---- a small persistent worker pool: parallel_for over byte ranges -------------------------------------------------
class Pool {
public:
    explicit Pool(int n) : n_(n)
    {
        for (int i = 0; i < n_; ++i) th_.emplace_back([this, i] { loop(i); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return n_; }
    // fn(part, parts) on every worker; returns when all are done
    void run(const std::function<void(int, int)> &fn)
    {
        std::lock_guard<std::mutex> one(run_mu_);   // one job at a time (the pool is shared by every handle of the process)
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &fn;
        pending_ = n_;
        ++gen_;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)> *fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            if (fn) (*fn)(id, n_);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *fn_ = nullptr;
    uint64_t gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

int usable_threads()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1, CPU_COUNT(&set));
    // cgroup v2 CPU quota (a container granted 16 of 256 cores sees all of them in the affinity mask)
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0) n = std::min(n, (int)std::max(1L, (quota + period / 2) / period));
        }
        fclose(f);
    }
    if (const char *e = getenv("NF_HOSTFED_THREADS")) n = atoi(e);
    return std::max(1, std::min(n, 32));
}

// ONE pool per process (every handle's host-fed calls share it; a handle used to keep up to 32 threads of its own alive).  Created
// on first use; a fork()ed child has none of the parent's threads, so the child starts over with a fresh pool (pthread_atfork) —
// the parent's pool object is abandoned in the child, never joined.
std::mutex g_pool_mu;
Pool *g_pool = nullptr;
void pool_forget_in_child() { g_pool = nullptr; new (&g_pool_mu) std::mutex(); }
Pool &shared_pool()
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool) {
        static bool hooked = false;
        if (!hooked) {
            (void)pthread_atfork(nullptr, nullptr, pool_forget_in_child);
            hooked = true;
        }
        g_pool = new Pool(usable_threads());
    }
    return *g_pool;
}

// float64 -> float32 with NON-TEMPORAL stores: the destination is pinned staging that only the DMA engine reads next, so
// write-allocating its cache lines would cost a third of the loop's memory traffic (64 B read + 32 B RFO + 32 B write per
// 8 values) for nothing.  dst is 32-byte aligned (callers split on multiples of 16 floats of a page-aligned buffer).
__attribute__((target("avx2"))) void narrow_avx2(float *dst, const double *src, size_t n)
{
    size_t i = 0;
    if ((reinterpret_cast<uintptr_t>(dst) & 31u) == 0) {
        for (; i + 8 <= n; i += 8) {
            const __m128 lo = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i));
            const __m128 hi = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4));
            _mm256_stream_ps(dst + i, _mm256_set_m128(hi, lo));
        }
        _mm_sfence();
    }
    for (; i < n; ++i) dst[i] = (float)src[i];
}
__attribute__((target("avx2"))) void copy_nt_avx2(float *dst, const float *src, size_t n)
{
    size_t i = 0;
    if ((reinterpret_cast<uintptr_t>(dst) & 31u) == 0) {
        for (; i + 8 <= n; i += 8) _mm256_stream_ps(dst + i, _mm256_loadu_ps(src + i));
        _mm_sfence();
    }
    for (; i < n; ++i) dst[i] = src[i];
}
void narrow_base(float *dst, const double *src, size_t n)
{
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i];
}

// dst[0, n) <- src[0, n) as float32 (src float32 or float64), split over the pool
// `inline_only`: on the calling thread — when several callers feed one handle at once (the reference's 16 / 32 sess.run threads,
// job_noise_flow.sh:36) THEY are the parallelism, and queueing each of their small chunks on the shared pool would serialise them
void stage_in(float *dst, const void *src, int dtype, size_t n, bool inline_only)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    auto piece = [&](size_t a, size_t b) {
        if (a >= b) return;
        if (dtype == NF_HOST_F64) {
            if (avx2) narrow_avx2(dst + a, (const double *)src + a, b - a);
            else narrow_base(dst + a, (const double *)src + a, b - a);
        } else if (avx2) {
            copy_nt_avx2(dst + a, (const float *)src + a, b - a);
        } else {
            memcpy(dst + a, (const float *)src + a, (b - a) * sizeof(float));
        }
    };
    if (inline_only || n < (1u << 17)) {
        piece(0, n);
        return;
    }
    shared_pool().run([&](int part, int parts) {
        const size_t per = ((n + parts - 1) / parts + 15) & ~(size_t)15;
        const size_t a = std::min(n, per * part);
        piece(a, std::min(n, a + per));
    });
}

// can the kernel store its tensor result straight into [p, p + bytes)?  Page-locked memory the device addresses directly
// (hipHostMalloc / hipHostRegister) — at BOTH ends of the range (a buffer registered in part would fault under the kernel) — and
// 16-byte aligned, as the kernels' float4 stores need; anything else goes through the slot's staging
bool is_pinned(const void *p, size_t bytes)
{
    if (!p || bytes == 0 || (reinterpret_cast<uintptr_t>(p) & 15u) != 0) return false;
    for (const char *q : {(const char *)p, (const char *)p + bytes - 1}) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();   // ordinary pageable memory: not an error for us
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

void stage_out(float *dst, const float *src, size_t n, bool inline_only)
{
    if (inline_only || n < (1u << 17)) {
        memcpy(dst, src, n * sizeof(float));
        return;
    }
    shared_pool().run([&](int part, int parts) {
        const size_t per = ((n + parts - 1) / parts + 15) & ~(size_t)15;
        const size_t a = std::min(n, per * part), b = std::min(n, a + per);
        if (a < b) memcpy(dst + a, src + a, (b - a) * sizeof(float));
    });
}

struct Slot {
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    float *h_a = nullptr, *h_b = nullptr;   // pinned inputs  (x | eps, y)
    float *h_t = nullptr, *h_s = nullptr;   // pinned outputs (tensor z | x, 3 per-patch scalars)
    float *d_a = nullptr, *d_b = nullptr, *d_t = nullptr /* unused */, *d_s = nullptr;
    int64_t first = 0, count = 0;           // the chunk this slot carries (count = 0: free)
};

}  // namespace

struct nf_hostpipe {
    int device = 0;
    int64_t chunk = 0;      // patches per chunk (upper bound)
    int64_t cap = 0;        // patches the slot buffers hold now: grows with the calls, up to `chunk`
    size_t px = 0;          // floats per patch tensor (H*W*4)
    Slot s[kSlots];
    double *d_sums = nullptr;
    double *h_sums = nullptr;   // pinned
    bool busy = false;          // a call is on this pipe (guarded by g_pipes_mu)
};

namespace {

// handle -> its pipes (the handle type is opaque here).  A call takes a pipe that no other call is on, or makes one — so the
// host threads that share a handle (train_noise_flow.py:30-47: every queue worker calls sess.run on its own) overlap one
// caller's narrowing with another's DMA and kernel instead of queueing on one pipe.  At most max_pipes() calls are in flight on a
// handle, further callers wait their turn (asleep): measured with 138-patch float64 minibatches, 4 calls in flight give 1.8 x
// the single caller's rate, 16 give 0.8 x — 16 threads polling events and taking turns on the HIP runtime's locks cost more
// than the overlap they add (tools/host_fed.py).  NF_HOSTFED_PIPES overrides (1 .. 16).
int max_pipes()
{
    static const int n = [] {
        const char *e = getenv("NF_HOSTFED_PIPES");
        const int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : v > 16 ? 16 : v;
    }();
    return n;
}
std::mutex g_pipes_mu;
std::condition_variable g_pipes_cv;
std::vector<std::pair<nf_handle *, nf_hostpipe *>> g_pipes;

void pipe_free(nf_hostpipe *p)
{
    if (!p) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(p->device);
    for (Slot &s : p->s) {
        if (s.st) (void)hipStreamSynchronize(s.st);
        if (s.h_a) (void)hipHostFree(s.h_a);
        if (s.h_b) (void)hipHostFree(s.h_b);
        if (s.h_t) (void)hipHostFree(s.h_t);
        if (s.h_s) (void)hipHostFree(s.h_s);
        if (s.d_a) (void)hipFree(s.d_a);
        if (s.d_b) (void)hipFree(s.d_b);
        if (s.d_t) (void)hipFree(s.d_t);
        if (s.d_s) (void)hipFree(s.d_s);
        if (s.ev) (void)hipEventDestroy(s.ev);
        if (s.st) (void)hipStreamDestroy(s.st);
    }
    if (p->d_sums) (void)hipFree(p->d_sums);
    if (p->h_sums) (void)hipHostFree(p->h_sums);
    delete p;
    if (prev >= 0) (void)hipSetDevice(prev);
}

// a pipe of the handle that no call is on — created on demand (device buffers + pinned staging for kSlots chunks) —, marked busy;
// `others` = how many other calls are on the handle's pipes right now
int pipe_get(nf_handle *h, int H, int W, int device, nf_hostpipe **out, int *others)
{
    std::unique_lock<std::mutex> lk(g_pipes_mu);
    for (;;) {
        int mine = 0, busy = 0;
        nf_hostpipe *free_pipe = nullptr;
        for (auto &kv : g_pipes)
            if (kv.first == h) {
                ++mine;
                if (kv.second->busy) ++busy;
                else if (!free_pipe) free_pipe = kv.second;
            }
        *others = busy;
        if (free_pipe) {
            free_pipe->busy = true;
            *out = free_pipe;
            return NF_OK;
        }
        if (mine < max_pipes()) break;
        g_pipes_cv.wait(lk);
    }
    nf_hostpipe *p = new (std::nothrow) nf_hostpipe();
    if (!p) return nf_fail(NF_ENOMEM, "out of host memory");
    p->device = device;
    p->px = (size_t)H * W * 4;
    // ~8 MiB per tensor and chunk: long enough for PCIe and the kernel to run at rate, short enough that the first
    // chunk's narrowing (which nothing overlaps) stays a small share of a 1024-patch call
    int64_t ch = (int64_t)((8u << 20) / (p->px * sizeof(float)));
    if (const char *e = getenv("NF_HOSTFED_CHUNK")) ch = atoll(e);
    p->chunk = std::max<int64_t>(1, ch);
    hipError_t e = hipSuccess;
    for (Slot &s : p->s) {
        if ((e = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking)) != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming)) != hipSuccess) break;
    }
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_sums, 3 * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->h_sums, 3 * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) {
        pipe_free(p);
        return nf_fail_hip(e, "host-fed pipeline allocation");
    }
    p->busy = true;
    g_pipes.emplace_back(h, p);
    *out = p;
    return NF_OK;
}

struct PipeLease {   // hands the pipe back when the call returns
    nf_hostpipe *p = nullptr;
    ~PipeLease()
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(g_pipes_mu);
            p->busy = false;
        }
        // every waiter, not one: the condition variable is shared by the waiters of EVERY handle, and a waiter of another handle
        // woken alone would find no free pipe of its own, sleep again and leave this handle's waiter asleep beside an idle pipe
        g_pipes_cv.notify_all();
    }
};

// slot buffers for chunks of up to min(need, chunk) patches (a wrapper that samples one patch at a time never pins 72 MiB)
int pipe_reserve(nf_hostpipe *p, int64_t need)
{
    need = std::min(std::max<int64_t>(need, 1), p->chunk);
    if (need <= p->cap) return NF_OK;
    const int64_t cap = std::min(p->chunk, std::max(need, 2 * p->cap));
    const size_t tb = (size_t)cap * p->px * sizeof(float), sb = (size_t)cap * 3 * sizeof(float);
    hipError_t e = hipSuccess;
    for (Slot &s : p->s) {
        (void)hipStreamSynchronize(s.st);
        float **hp[4] = {&s.h_a, &s.h_b, &s.h_t, &s.h_s};
        float **dp[4] = {&s.d_a, &s.d_b, &s.d_t, &s.d_s};
        for (int k = 0; k < 4 && e == hipSuccess; ++k) {
            if (*hp[k]) (void)hipHostFree(*hp[k]);
            if (*dp[k]) (void)hipFree(*dp[k]);
            *hp[k] = *dp[k] = nullptr;
            if ((e = hipHostMalloc((void **)hp[k], k == 3 ? sb : tb, hipHostMallocDefault)) != hipSuccess) break;
            if (k != 2) e = hipMalloc((void **)dp[k], k == 3 ? sb : tb);   // tensor results never sit in device memory
        }
        if (e != hipSuccess) break;
    }
    if (e != hipSuccess) {
        p->cap = 0;
        return nf_fail_hip(e, "host-fed pipeline allocation");
    }
    p->cap = cap;
    return NF_OK;
}

// chunk length of a call: at least ~6 chunks per call, so that the narrowing of the first chunk (which nothing overlaps) and
// the return trip of the last stay a small share; never below 64 patches (launch + copy set-up ~20 us per chunk)
int64_t pipe_chunk(const nf_hostpipe *p, int64_t B)
{
    int64_t ch = std::max<int64_t>(64, (B + 5) / 6);
    return std::max<int64_t>(1, std::min(ch, p->cap));
}

struct DevGuard {
    int prev = -1;
    bool changed = false;
    int enter(int dev)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return nf_fail_hip(e, "hipGetDevice");
        if (prev != dev) {
            if ((e = hipSetDevice(dev)) != hipSuccess) return nf_fail_hip(e, "hipSetDevice");
            changed = true;
        }
        return NF_OK;
    }
    ~DevGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

}  // namespace

// called by nf_destroy (nf_host.hip)
void nf_hostpipe_release(nf_handle *h)
{
    std::vector<nf_hostpipe *> mine;
    {
        std::lock_guard<std::mutex> lk(g_pipes_mu);
        for (size_t i = 0; i < g_pipes.size();)
            if (g_pipes[i].first == h) {
                mine.push_back(g_pipes[i].second);
                g_pipes.erase(g_pipes.begin() + i);
            } else {
                ++i;
            }
    }
    for (nf_hostpipe *p : mine) pipe_free(p);
}

extern "C" {

int nf_nll_host(nf_handle *h, const void *x, const void *y, int32_t dtype, int64_t B, const nf_cond *cond, float *nll_out,
                float *sd_out, float *logdet_out, float *z_out, double *sums_out, uint32_t flags)
{
    if (!h || !x || !cond) return nf_fail(NF_EINVAL, "null argument");
    if (dtype != NF_HOST_F32 && dtype != NF_HOST_F64) return nf_fail(NF_EINVAL, "dtype must be NF_HOST_F32 or NF_HOST_F64");
    if (B < 0) return nf_fail(NF_EINVAL, "B must be >= 0");
    if (flags & ~(uint32_t)(NF_NO_PRIOR | NF_ACCUMULATE)) return nf_fail(NF_EINVAL, "nf_nll_host takes NF_NO_PRIOR / NF_ACCUMULATE only");
    int32_t H = 0, W = 0, device = 0;
    int rc = nf_handle_geometry(h, &H, &W, &device);
    if (rc != NF_OK) return rc;
    DevGuard guard;
    if ((rc = guard.enter(device)) != NF_OK) return rc;
    nf_hostpipe *p = nullptr;
    int others = 0;
    if ((rc = pipe_get(h, H, W, device, &p, &others)) != NF_OK) return rc;
    PipeLease lease;
    lease.p = p;
    const bool solo = others == 0;   // no other caller on this handle: the shared pool narrows; else this thread does
    if ((rc = pipe_reserve(p, B)) != NF_OK) return rc;
    const int64_t CH = pipe_chunk(p, B);
    const size_t px = p->px, esz = dtype == NF_HOST_F64 ? 8 : 4;
    const bool z_direct = is_pinned(z_out, (size_t)B * px * sizeof(float));   // page-locked caller memory: the kernel stores straight into it
    hipError_t e;
    if (sums_out && (e = hipMemsetAsync(p->d_sums, 0, 3 * sizeof(double), p->s[0].st)) != hipSuccess) return nf_fail_hip(e, "hipMemsetAsync");
    if (sums_out && (e = hipStreamSynchronize(p->s[0].st)) != hipSuccess) return nf_fail_hip(e, "hipStreamSynchronize");

    auto retire = [&](Slot &s) -> int {   // wait for the slot's chunk and hand its results to the caller
        if (s.count == 0) return NF_OK;
        hipError_t er = hipEventSynchronize(s.ev);
        if (er != hipSuccess) return nf_fail_hip(er, "host-fed chunk");
        const size_t n = (size_t)s.count;
        if (nll_out) memcpy(nll_out + s.first, s.h_s, n * sizeof(float));
        if (sd_out) memcpy(sd_out + s.first, s.h_s + CH, n * sizeof(float));
        if (logdet_out) memcpy(logdet_out + s.first, s.h_s + 2 * CH, n * sizeof(float));
        if (z_out && !z_direct) stage_out(z_out + (size_t)s.first * px, s.h_t, n * px, !solo);
        s.count = 0;
        return NF_OK;
    };

    int c = 0;
    for (int64_t first = 0; first < B; first += CH, ++c) {
        Slot &s = p->s[c % kSlots];
        if ((rc = retire(s)) != NF_OK) break;
        const int64_t n = std::min(CH, B - first);
        stage_in(s.h_a, (const char *)x + (size_t)first * px * esz, dtype, (size_t)n * px, !solo);
        if (y) stage_in(s.h_b, (const char *)y + (size_t)first * px * esz, dtype, (size_t)n * px, !solo);
        if ((e = hipMemcpyAsync(s.d_a, s.h_a, (size_t)n * px * 4, hipMemcpyHostToDevice, s.st)) != hipSuccess ||
            (y && (e = hipMemcpyAsync(s.d_b, s.h_b, (size_t)n * px * 4, hipMemcpyHostToDevice, s.st)) != hipSuccess)) {
            rc = nf_fail_hip(e, "hipMemcpyAsync(H2D)");
            break;
        }
        rc = nf_nll(h, s.d_a, y ? s.d_b : nullptr, n, cond, nll_out ? s.d_s : nullptr, sd_out ? s.d_s + CH : nullptr,
                    logdet_out ? s.d_s + 2 * CH : nullptr, z_out ? (z_direct ? z_out + (size_t)first * px : s.h_t) : nullptr,
                    sums_out ? p->d_sums : nullptr,
                    (flags & NF_NO_PRIOR) | NF_ACCUMULATE, s.st);
        if (rc != NF_OK) break;
        if (nll_out || sd_out || logdet_out)
            if ((e = hipMemcpyAsync(s.h_s, s.d_s, (size_t)CH * 3 * 4, hipMemcpyDeviceToHost, s.st)) != hipSuccess) rc = nf_fail_hip(e, "D2H");
        if (rc == NF_OK && (e = hipEventRecord(s.ev, s.st)) != hipSuccess) rc = nf_fail_hip(e, "hipEventRecord");
        if (rc != NF_OK) break;
        s.first = first;
        s.count = n;
    }
    for (int k = 0; k < kSlots; ++k) {   // drain in issue order (also on the error path: nothing may stay in flight)
        Slot &s = p->s[(c + k) % kSlots];
        if (rc == NF_OK) rc = retire(s);
        else if (s.count) { (void)hipEventSynchronize(s.ev); s.count = 0; }
    }
    if (rc != NF_OK) {
        for (Slot &s : p->s) (void)hipStreamSynchronize(s.st);
        return rc;
    }
    if (sums_out) {
        // every chunk's stream has been waited for: the atomics are complete
        if ((e = hipMemcpy(p->h_sums, p->d_sums, 3 * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) return nf_fail_hip(e, "hipMemcpy(sums)");
        for (int k = 0; k < 3; ++k) sums_out[k] = ((flags & NF_ACCUMULATE) ? sums_out[k] : 0.0) + p->h_sums[k];
    }
    return NF_OK;
}

int nf_sample_host(nf_handle *h, const void *y, int32_t y_dtype, const float *eps, uint64_t seed, int64_t patch_index_base,
                   float temp, int64_t B, const nf_cond *cond, float *x_out)
{
    if (!h || !cond || !x_out) return nf_fail(NF_EINVAL, "null argument");
    if (y_dtype != NF_HOST_F32 && y_dtype != NF_HOST_F64) return nf_fail(NF_EINVAL, "y_dtype must be NF_HOST_F32 or NF_HOST_F64");
    if (B < 0) return nf_fail(NF_EINVAL, "B must be >= 0");
    int32_t H = 0, W = 0, device = 0;
    int rc = nf_handle_geometry(h, &H, &W, &device);
    if (rc != NF_OK) return rc;
    DevGuard guard;
    if ((rc = guard.enter(device)) != NF_OK) return rc;
    nf_hostpipe *p = nullptr;
    int others = 0;
    if ((rc = pipe_get(h, H, W, device, &p, &others)) != NF_OK) return rc;
    PipeLease lease;
    lease.p = p;
    const bool solo = others == 0;
    if ((rc = pipe_reserve(p, B)) != NF_OK) return rc;
    const int64_t CH = pipe_chunk(p, B);
    const size_t px = p->px, esz = y_dtype == NF_HOST_F64 ? 8 : 4;
    const bool x_direct = is_pinned(x_out, (size_t)B * px * sizeof(float));   // page-locked caller memory: the kernel stores straight into it
    hipError_t e;

    auto retire = [&](Slot &s) -> int {
        if (s.count == 0) return NF_OK;
        hipError_t er = hipEventSynchronize(s.ev);
        if (er != hipSuccess) return nf_fail_hip(er, "host-fed chunk");
        if (!x_direct) stage_out(x_out + (size_t)s.first * px, s.h_t, (size_t)s.count * px, !solo);
        s.count = 0;
        return NF_OK;
    };

    int c = 0;
    for (int64_t first = 0; first < B; first += CH, ++c) {
        Slot &s = p->s[c % kSlots];
        if ((rc = retire(s)) != NF_OK) break;
        const int64_t n = std::min(CH, B - first);
        if (y) stage_in(s.h_b, (const char *)y + (size_t)first * px * esz, y_dtype, (size_t)n * px, !solo);
        if (eps) stage_in(s.h_a, eps + (size_t)first * px, NF_HOST_F32, (size_t)n * px, !solo);
        if ((y && (e = hipMemcpyAsync(s.d_b, s.h_b, (size_t)n * px * 4, hipMemcpyHostToDevice, s.st)) != hipSuccess) ||
            (eps && (e = hipMemcpyAsync(s.d_a, s.h_a, (size_t)n * px * 4, hipMemcpyHostToDevice, s.st)) != hipSuccess)) {
            rc = nf_fail_hip(e, "hipMemcpyAsync(H2D)");
            break;
        }
        // the kernel stores x straight into page-locked HOST memory (the caller's, or the slot's staging): posted PCIe writes
        // that run concurrently with the next chunk's H2D DMA — a D2H memcpy would queue behind it on the copy engine
        // (measured: 24 GB/s each way with a D2H copy, i.e. the two directions took turns)
        rc = nf_sample(h, y ? s.d_b : nullptr, eps ? s.d_a : nullptr, seed, patch_index_base + first, temp, n, cond,
                       x_direct ? x_out + (size_t)first * px : s.h_t, s.st);
        if (rc != NF_OK) break;
        if ((e = hipEventRecord(s.ev, s.st)) != hipSuccess) {
            rc = nf_fail_hip(e, "hipEventRecord");
            break;
        }
        s.first = first;
        s.count = n;
    }
    for (int k = 0; k < kSlots; ++k) {
        Slot &s = p->s[(c + k) % kSlots];
        if (rc == NF_OK) rc = retire(s);
        else if (s.count) { (void)hipEventSynchronize(s.ev); s.count = 0; }
    }
    if (rc != NF_OK)
        for (Slot &s : p->s) (void)hipStreamSynchronize(s.st);
    return rc;
}

}  // extern "C"
