#!/usr/bin/env python
"""Demo sampler — the data-free core of the reference's ``sample_noise_flow.py``.

The reference script (``sample_noise_flow.py:27-101``) loads three SIDD scenes (20 GB,
h5py), crops 10 patches each, calls ``NoiseFlowWrapper.sample_noise_nf``, renders
sRGB PNGs through the SIDD ISP and prints the mean marginal KL.  SIDD, h5py and
cv2 are not available here, so this script keeps exactly the hot-path part:

    clean patch [1,32,32,4] -> wrapper.sample_noise_nf(clean, 0, 0, iso, cam) -> crop 1 px
    -> clip(clean + noise) -> unpack_raw (Bayer) -> KL(real noise || synthesised noise)

on synthetic clean patches (or ``--clean file.npy`` with [N,32,32,4] values in [0,1]) with
"real" noise drawn from the camera NLF of ``cam_iso_nlf.txt`` (S6).  Output: ``--out`` .npz.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default=os.path.join(ROOT, "models", "NoiseFlow"))
    ap.add_argument("--clean", default=None, help=".npy of clean packed-raw patches [N,32,32,4] in [0,1]")
    ap.add_argument("--n", type=int, default=30)
    ap.add_argument("--iso", type=float, default=800.0)
    ap.add_argument("--cam", type=float, default=2.0, help="0..4 = IP, GP, S6, N6, G4")
    ap.add_argument("--temp", type=float, default=0.6)        # sample_noise_flow.py:36-40
    ap.add_argument("--out", default="samples_amd.npz")
    ap.add_argument("--seed", type=int, default=None, help="Philox key of the in-kernel N(0,1) draw (default: hps.txt's seed)")
    ap.add_argument("--compat", default=None, choices=["reference"],
                    help="'reference': what the upstream wrapper literally builds (sampling-graph-first template binding + "
                         "batch-statistics BN, quirks Q1/Q2) instead of the trained model's semantics")
    args = ap.parse_args()

    from noise_flow_amd import NoiseFlowWrapper
    from noise_flow_amd.harness import S6_NLF
    from noise_flow_amd.metrics import kl_div_3_data, noise_bin_edges
    from noise_flow_amd.patches import unpack_raw

    np.random.seed(12345)                                    # sample_noise_flow.py:58
    clean = np.load(args.clean).astype(np.float32) if args.clean else np.random.rand(args.n, 32, 32, 4).astype(np.float32)
    b1, b2 = S6_NLF.get(int(args.iso), S6_NLF[100])
    real_noise = np.random.randn(*clean.shape) * np.sqrt(b1 * clean + b2)
    nf = NoiseFlowWrapper(args.model, sampling_temperature=args.temp, seed=args.seed, compat=args.compat)
    klds, noisy_syn, noise_syn = [], [], []
    for p in range(clean.shape[0]):                          # batch_size = 1 like the reference demo
        c = clean[p:p + 1]
        n_full = np.squeeze(nf.sample_noise_nf(c, 0.0, 0.0, args.iso, args.cam))
        noise_syn.append(n_full)
        n_syn = n_full[1:-1, 1:-1, :]
        cc = np.squeeze(c)[1:-1, 1:-1, :]
        noisy_syn.append(unpack_raw(np.clip(cc + n_syn, 0.0, 1.0)))
        n_real = real_noise[p, 1:-1, 1:-1, :]
        klds.append(kl_div_3_data(unpack_raw(n_real).ravel(), unpack_raw(n_syn).ravel(), noise_bin_edges(200))[0])
    np.savez_compressed(args.out, clean=clean, noise_syn=np.stack(noise_syn), noisy_syn=np.stack(noisy_syn), kld=np.asarray(klds))
    print("Mean KL divergence = {}".format(np.mean(klds)))


if __name__ == "__main__":
    main()
