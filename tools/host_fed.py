"""Host-fed (PCIe-inclusive) rates of the Python surface: numpy in -> numpy out (nf_nll_host / nf_sample_host), and the
bit-identity of their per-patch results with the device-resident path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
rng = np.random.RandomState(0)
for B in (64, 1024, 4096):
    y = rng.rand(B, 32, 32, 4).astype(np.float32)
    x = (rng.randn(B, 32, 32, 4) * 0.02).astype(np.float32)
    y64, x64 = y.astype(np.float64), x.astype(np.float64)      # the reference's minibatch dtype (quirk Q9)
    nll_h, _ = m._loss(x64, y64, [0], [0], [100], [2])
    nll_d, _ = m._loss(torch.tensor(x).cuda(), torch.tensor(y).cuda(), [0], [0], [100], [2])
    xs_h = m.sample(y64, 0.6, y64, [0], [0], [100], [2], seed=3)
    m._draws -= B
    xs_d = m.sample(torch.tensor(y).cuda(), 0.6, torch.tensor(y).cuda(), [0], [0], [100], [2], seed=3)
    print("B=%5d host-fed == device-resident: nll %s, sample %s" % (B, np.array_equal(nll_h, nll_d.cpu().numpy()),
                                                                  np.array_equal(xs_h, xs_d.cpu().numpy())), flush=True)
    for name, fn in (("loss(float32 numpy)", lambda: m.loss(x, y, [0], [0], [100], [2])),
                     ("loss(float64 numpy)", lambda: m.loss(x64, y64, [0], [0], [100], [2])),
                     ("_loss(float64 numpy)", lambda: m._loss(x64, y64, [0], [0], [100], [2])),
                     ("sample(float32 numpy)", lambda: m.sample(y, 0.6, y, [0], [0], [100], [2])),
                     ("sample(float64 numpy)", lambda: m.sample(y64, 0.6, y64, [0], [0], [100], [2]))):
        for _ in range(3):
            fn()
        n = 20
        t = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t) / n
        print("B=%5d %-24s %.3f ms/call  %.3e patches/s" % (B, name, dt * 1e3, B / dt), flush=True)

# ---- many callers on one handle (the reference's queue workers: job_noise_flow.sh:36 runs 16 threads, each calling sess.run
#      on its own minibatch of 138 float64 patches, train_noise_flow.py:30-47): aggregate rate, 1 thread vs 16
import threading
Bm = 138
ym = rng.rand(Bm, 32, 32, 4)
xm = rng.randn(Bm, 32, 32, 4) * 0.02
ref, _ = m._loss(xm, ym, [0], [0], [100], [2])
for nthreads in (1, 4, 16):
    calls = 200
    outs = [None] * nthreads

    def work(i):
        for _ in range(calls // nthreads):
            outs[i] = m._loss(xm, ym, [0], [0], [100], [2])[0]
    for i in range(nthreads):
        work(i)                                    # warm-up: every thread's pipe exists
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t = time.perf_counter()
    for k in th:
        k.start()
    for k in th:
        k.join()
    dt = time.perf_counter() - t
    done = (calls // nthreads) * nthreads
    same = all(np.array_equal(o, ref) for o in outs)
    print("host-fed _loss(float64), B=%d per call, %2d threads on ONE handle: %.3e patches/s aggregate, identical results: %s"
          % (Bm, nthreads, done * Bm / dt, same), flush=True)
