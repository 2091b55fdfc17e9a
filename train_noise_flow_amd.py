#!/usr/bin/env python
"""Training driver — the data-free core of the reference's ``train_noise_flow.py``.

The reference script (``train_noise_flow.py:240-530``) loads SIDD_Medium_Raw (h5py / .mat, 20 GB),
samples 32x32 patches into minibatch queues and runs, per epoch, test / sampling / training
through one tf.Session.  SIDD is not available here, so this driver keeps the loop and replaces
the data with counter-based synthetic patches drawn from the S6 camera NLF
(``nf_synth_patches``): ``y ~ U[0,1)``, ``x = eps * sqrt(beta1*y + beta2)`` — a target the flow can
actually learn (its optimum is the signal-dependent Gaussian itself).

    python train_noise_flow_amd.py --logdir /tmp/nf_run --epochs 20 --n_train 4140 --n_test 1380

writes ``train.txt / test.txt / sample.txt`` (TSV, reference columns), ``hps.txt`` and
``ckpt/model.ckpt-<epoch>``, ``ckpt/model.ckpt.best`` (TF-bundle format) under ``--logdir``.
Multi-GPU: launch under ``python -m torch.distributed.run --nproc-per-node N`` — every rank draws
its own shard of the training patches and the gradient is averaged with one RCCL all-reduce per
step; ``--sync_bn`` also all-reduces the batch-normalisation sums (the ranks then take the step a
single process would take on the union of their minibatches).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logdir", required=True)
    ap.add_argument("--arch", default="sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc")
    ap.add_argument("--width", type=int, default=4)
    ap.add_argument("--flow_permutation", type=int, default=1)      # sidd/ArgParser.py: 1 = Conv2d1x1, 0 = tfb.Permute, else none
    ap.add_argument("--decomp", default="LU", choices=["LU", "LU2", "NONE"])
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--lr", type=float, default=1e-4)                # job_noise_flow.sh:37
    ap.add_argument("--optim", default="adam", choices=["adam", "sgd"])
    ap.add_argument("--n_batch_train", type=int, default=138)
    ap.add_argument("--n_batch_test", type=int, default=138)
    ap.add_argument("--epochs_full_valid", type=int, default=10)
    ap.add_argument("--n_train", type=int, default=4140)
    ap.add_argument("--n_test", type=int, default=1380)
    ap.add_argument("--iso", type=float, default=800.0)
    ap.add_argument("--cam", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--sync_bn", action="store_true",
                    help="multi-GPU: batch-norm moments over the GLOBAL minibatch (all ranks take the single-process step)")
    ap.add_argument("--init", default=None, help="checkpoint prefix to start from (default: fresh initialisation)")
    ap.add_argument("--pipeline", default="resident", choices=["resident", "queues"],
                    help="resident: training minibatches generated once on the GPU; queues: the reference's host pipeline — image "
                         "tuples -> PatchSampler -> MiniBatchSampler queues (sidd/PatchSampler.py, sidd/MiniBatchSampler.py), numpy "
                         "float64 minibatch dicts fed per step")
    args = ap.parse_args()

    import torch
    from noise_flow_amd import NoiseFlow, default_hps, patches
    from noise_flow_amd.harness import S6_NLF, fit
    from noise_flow_amd.hps import hps_logger
    from noise_flow_amd.metrics import nll_gauss, nll_sdn
    from noise_flow_amd.train import Trainer
    from noise_flow_amd.ckpt import load_checkpoint

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    group = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
        group = True

    hps = default_hps(arch=args.arch, width=args.width, flow_permutation=args.flow_permutation, decomp=args.decomp, seed=args.seed,
                      optim=args.optim, lr=args.lr,
                      n_batch_train=args.n_batch_train, n_batch_test=args.n_batch_test, epochs=args.epochs)
    variables = load_checkpoint(args.init) if args.init else None
    trainer = Trainer([32, 32, 4], hps, variables=variables, max_batch=max(args.n_batch_train, 1))
    nf_eval = NoiseFlow([32, 32, 4], False, hps, variables=trainer.variables)
    nlf = S6_NLF.get(int(args.iso), S6_NLF[800])

    def minibatches(first, count, bs):
        out = []
        for k in range(first, first + count - bs + 1, bs):
            x, y = patches.synth_patches(args.seed, k, bs, nlf=nlf)
            out.append({"_x": x, "_y": y, "nlf0": [nlf[0]], "nlf1": [nlf[1]], "iso": [args.iso], "cam": [args.cam]})
        return out

    # every rank must take the SAME number of optimizer steps (one gradient all-reduce per step): equal blocks of
    # n_train // world patches, the remainder is dropped
    per_rank = args.n_train // world
    stages = []
    if args.pipeline == "queues":
        # the reference's host pipeline on in-memory image tuples: every "image" is a 4 x 4 mosaic of this rank's synthetic
        # 32x32 patches ('in' = the noise layer, sidd_utils.py:264-265); patches are drawn on the grid in shuffled order and
        # collated into float64 minibatch dicts by the sampler threads
        from noise_flow_amd.samplers import ImageTupleFeeder, MiniBatchSampler, PatchSampler, QueueEpoch
        np.random.seed(args.seed + rank)
        n_img = max(1, per_rank // 16)
        tuples = []
        for k in range(n_img):
            x, y = patches.synth_patches(args.seed, rank * per_rank + 16 * k, 16, nlf=nlf)
            mosaic = lambda t: t.cpu().numpy().astype(np.float64).reshape(4, 4, 32, 32, 4).transpose(0, 2, 1, 3, 4).reshape(1, 128, 128, 4)   # noqa: E731
            tuples.append({"in": mosaic(x), "gt": mosaic(y), "nlf0": nlf[0], "nlf1": nlf[1], "iso": args.iso, "cam": args.cam,
                           "fn": "synth_%04d" % k, "metadata": None})
        feeder = ImageTupleFeeder(tuples)
        ps = PatchSampler(feeder.get_queue(), patch_height=32, sampling="uniform", n_threads=1, n_pat_per_im=16, shuffle=True)
        ms = MiniBatchSampler(ps.get_queue(), minibatch_size=args.n_batch_train, n_threads=1)
        stages = [ms, ps, feeder]
        train_mbs = QueueEpoch(ms.get_queue(), max(1, (n_img * 16) // args.n_batch_train))
    else:
        train_mbs = minibatches(rank * per_rank, per_rank, args.n_batch_train)
    test_mbs = minibatches(args.n_train, args.n_test, args.n_batch_test)     # every rank evaluates the same test set
    # closed-form baselines of the test noise (sidd/PatchStatsCalculator.py:92-123)
    xt = np.concatenate([mb["_x"].cpu().numpy() for mb in test_mbs])
    yt = np.concatenate([mb["_y"].cpu().numpy() for mb in test_mbs])
    base_g = float(np.mean(nll_gauss(xt, xt.std())))
    base_sdn = float(np.mean(nll_sdn(xt, yt, nlf[0], nlf[1])))

    logdir = args.logdir if world == 1 or rank == 0 else os.path.join(args.logdir, "rank%d" % rank)
    os.makedirs(logdir, exist_ok=True)
    hps_logger(os.path.join(logdir, "hps.txt"), hps, nf_eval.get_layer_names(), nf_eval.num_params())
    log = (lambda s: print(s, flush=True)) if rank == 0 else None
    if log:
        log("train minibatches/epoch %d x %d patches (rank 0 of %d), test %d x %d; NLL_G %.2f NLL_SDN %.2f" % (
            len(train_mbs), args.n_batch_train, world, len(test_mbs), args.n_batch_test, base_g, base_sdn))
    res = fit(trainer, nf_eval, train_mbs, test_mbs, logdir, args.epochs, args.lr, args.epochs_full_valid,
              nll_gauss=base_g, nll_sdn=base_sdn, group=group, log=log, sync_bn=args.sync_bn,
              sc_sd=float(xt.std()))        # pat_stats['sc_in_sd'] of the reference: the KLD_G model of sample.txt
    for st in stages:
        st.close()
    if log:
        log("final: train %.4f  test %.4f (first %.4f)" % (res["train"][-1], res["test"][-1], res["test"][0]))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
