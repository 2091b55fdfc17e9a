"""Experiment: the training step replayed from a HIP graph (stream capture of nf_trainer_step) vs enqueued launch by launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.patches import synth_patches
from noise_flow_amd.train import Trainer
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
for B in (138, 1024):
    tr = Trainer([32, 32, 4], default_hps(), variables=v, max_batch=B)
    x, y = synth_patches(0, 0, B)
    for _ in range(5):
        tr.step(x, y, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        tr.step(x, y, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
    torch.cuda.synchronize()
    print("B=%d stream launches: %.3f ms/step" % (B, (time.perf_counter() - t) / 50 * 1e3))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            tr.step(x, y, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            tr.step(x, y, [0.0], [0.0], [100.0], [2.0], lr=1e-4, sync=False)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        print("B=%d graph replay   : %.3f ms/step" % (B, (time.perf_counter() - t) / 50 * 1e3))
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
