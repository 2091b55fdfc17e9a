"""BASELINE configs[3] on one GPU: 2^20 synthetic patches through the sharded-evaluation path
(synthesis + NLL + slotted sums, chunks alternating between two streams), wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from noise_flow_amd import NoiseFlow, default_hps
from noise_flow_amd.ckpt import load_checkpoint
from noise_flow_amd.dist import evaluate_sharded, flow_eval_chunk
v = load_checkpoint("models/NoiseFlow/ckpt/model.ckpt.best")
m = NoiseFlow([32, 32, 4], False, default_hps(), variables=v)
n = 1 << 20
for chunk in (1024, 4096, 32768):
    for ns in (1, 2):
        run = flow_eval_chunk(m, seed=0, n_streams=ns)
        evaluate_sharded(run, 1 << 16, chunk, 0, 1, torch.zeros(3, dtype=torch.float64, device="cuda"))   # warm-up
        torch.cuda.synchronize()
        t = time.perf_counter()
        mean, sd, cnt = evaluate_sharded(run, n, chunk, 0, 1, torch.zeros(3, dtype=torch.float64, device="cuda"))
        dt = time.perf_counter() - t
        print("chunk %6d streams %d: %.1f ms for %d patches = %.3e patches/s (incl. synthesis)  mean NLL/dim %.4f sd_z %.4f"
              % (chunk, ns, dt * 1e3, cnt, cnt / dt, mean / 4096, sd))
