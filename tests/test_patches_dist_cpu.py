"""CPU tier: patch indexing / minibatch contract and the N>1 path
(world_size-2 gloo, run here without a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_patch_origins_match_reference_rule():
    from noise_flow_amd.patches import patch_origins, patch_index_to_origin
    ii, jj, n = patch_origins(100, 100, 32, 32)
    assert n == 9 and ii == [0, 0, 0, 32, 32, 32, 64, 64, 64] and jj == [0, 32, 64] * 3
    ii, jj, n = patch_origins(100, 70, 32, 32, n_pat_per_im=5)
    assert n == 5 and list(zip(ii, jj)) == [(0, 0), (0, 32), (32, 0), (32, 32), (64, 0)]
    assert patch_origins(31, 100, 32, 32)[2] == 0
    for h, w, ph, pw in ((100, 100, 32, 32), (64, 200, 32, 64), (33, 33, 32, 32)):
        ii, jj, n = patch_origins(h, w, ph, pw)
        assert [(i, j) for i, j in zip(ii, jj)] == [patch_index_to_origin(k, h, w, ph, pw) for k in range(n)]
    with pytest.raises(IndexError):
        patch_index_to_origin(9, 100, 100, 32, 32)
    a = patch_origins(100, 100, 32, 32, shuffle_seed=3)
    assert sorted(zip(a[0], a[1])) == sorted(zip(*patch_origins(100, 100, 32, 32)[:2]))


def test_bayer_packing_order_and_round_trip():
    from noise_flow_amd.patches import pack_raw, unpack_raw, extract_patches, make_minibatch
    raw = np.arange(8 * 12, dtype=np.float32).reshape(8, 12)
    p = pack_raw(raw)
    assert p.shape == (4, 6, 4)
    assert (p[0, 0] == [raw[0, 0], raw[0, 1], raw[1, 1], raw[1, 0]]).all()      # sidd_utils.py:741-744
    assert np.array_equal(unpack_raw(p), raw)
    img = np.random.RandomState(0).rand(70, 100, 4)
    pt = extract_patches(img, 32, 32)
    assert pt.shape == (6, 32, 32, 4) and np.array_equal(pt[4], img[32:64, 32:64])
    mb = make_minibatch(pt + 0.1, pt, np.arange(6), 1e-3, 1e-6, 100, 2)
    assert mb["_x"].dtype == np.float64 and np.allclose(mb["_x"], 0.1)
    assert mb["iso"] == [100] and mb["cam"] == [2] and set(mb) >= {"_x", "_y", "pid", "nlf0", "nlf1", "iso", "cam", "fn", "metadata"}


def test_shard_range_partitions_exactly():
    from noise_flow_amd.patches import shard_range
    for n in (0, 1, 7, 1024, 1048576, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, chunk, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from noise_flow_amd.dist import evaluate_sharded
    from oracle import philox
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    sums = torch.zeros(3, dtype=torch.float64)

    def eval_chunk(first, count, acc):
        # stand-in evaluator with the real data contract: statistics are a pure function of the patch index
        x, y = philox.synth_patches(7, first, count, 8, 8)
        acc += torch.tensor([float((x.astype(np.float64) ** 2).sum()), float(y.astype(np.float64).sum()), count],
                            dtype=torch.float64)
    out = evaluate_sharded(eval_chunk, n_total, chunk, rank, world, sums)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_evaluation_gloo(world):
    import torch.multiprocessing as mp
    from oracle import philox
    n_total, chunk = 101, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, chunk, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    x, y = philox.synth_patches(7, 0, n_total, 8, 8)
    want = ((x.astype(np.float64) ** 2).sum() / n_total, y.astype(np.float64).sum() / n_total, n_total)
    for _, out in res:
        assert out[2] == n_total
        assert abs(out[0] - want[0]) <= 1e-12 * abs(want[0]) and abs(out[1] - want[1]) <= 1e-12 * abs(want[1])
    assert all(p.exitcode == 0 for p in procs)


def test_evaluate_sharded_single_process_detects_miscount():
    import torch
    from noise_flow_amd.dist import evaluate_sharded
    sums = torch.zeros(3, dtype=torch.float64)
    with pytest.raises(RuntimeError):
        evaluate_sharded(lambda a, n, s: s.add_(torch.tensor([0.0, 0.0, n - 1.0], dtype=torch.float64)), 10, 4, 0, 1, sums)


def _timed_worker(rank, world, port, n_total, chunk, steps, warmup, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from noise_flow_amd.dist import timed_sharded_evaluations
    from oracle import philox
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    calls = {"chunks": 0, "finish": 0, "events": []}

    def eval_chunk(first, count, acc):
        calls["chunks"] += 1
        x, y = philox.synth_patches(7, first, count, 8, 8)
        acc += torch.tensor([float((x.astype(np.float64) ** 2).sum()), float(y.astype(np.float64).sum()), count],
                            dtype=torch.float64)

    def finish(acc):
        calls["finish"] += 1
    eval_chunk.finish = finish
    res = timed_sharded_evaluations(eval_chunk, n_total, chunk, rank, world, steps, warmup,
                                    lambda: torch.zeros(3, dtype=torch.float64),
                                    on_step=lambda i, what: calls["events"].append((i, what)))
    q.put((rank, res, calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_timed_sharded_evaluations_gloo(world):
    """The code path of `bench.py --gpus N` (BASELINE configs[3]): K complete evaluations of the sharded patch range,
    ONE all-reduce each, results identical on every rank and equal to the unsharded statistics."""
    import torch.multiprocessing as mp
    from oracle import philox
    from noise_flow_amd.patches import shard_range
    n_total, chunk, steps, warmup = 77, 16, 3, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timed_worker, args=(r, world, port, n_total, chunk, steps, warmup, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    x, y = philox.synth_patches(7, 0, n_total, 8, 8)
    want = ((x.astype(np.float64) ** 2).sum() / n_total, y.astype(np.float64).sum() / n_total)
    for rank, out, calls in res:
        start, stop = shard_range(n_total, rank, world)
        assert out["shard"] == (start, stop) and out["elapsed"] > 0 and len(out["results"]) == steps
        for mean, sd, n in out["results"]:
            assert n == n_total and abs(mean - want[0]) <= 1e-12 * abs(want[0]) and abs(sd - want[1]) <= 1e-12 * abs(want[1])
        n_chunks = -(-(stop - start) // chunk)
        assert calls["chunks"] == n_chunks * (steps + warmup) and calls["finish"] == steps + warmup
        assert calls["events"] == [(i, w) for i in range(steps) for w in ("begin", "end")]   # timed steps only


def test_timed_sharded_evaluations_without_process_group():
    import torch
    from noise_flow_amd.dist import timed_sharded_evaluations
    seen = []

    def eval_chunk(first, count, acc):
        seen.append((first, count))
        acc += torch.tensor([2.0 * count, 1.0 * count, float(count)], dtype=torch.float64)
    out = timed_sharded_evaluations(eval_chunk, 10, 4, 0, 1, 2, 0, lambda: torch.zeros(3, dtype=torch.float64))
    assert seen == [(0, 4), (4, 4), (8, 2)] * 2 and out["results"] == [(2.0, 1.0, 10)] * 2
    with pytest.raises(RuntimeError):
        timed_sharded_evaluations(lambda a, n, s: s.add_(torch.tensor([0.0, 0.0, n - 1.0], dtype=torch.float64)), 10, 4, 0, 1, 1, 0,
                                  lambda: torch.zeros(3, dtype=torch.float64))


# ---------------------------------------------------------------------------------------------------------
# bench.py's multi-GPU legs: launch convention and the configs[4] (64x64, fp16 CNN) sharded workload
# ---------------------------------------------------------------------------------------------------------
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("nf_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launches_its_own_ranks_when_not_under_torchrun():
    """`python bench.py --gpus N` without WORLD_SIZE must become the torch.distributed.run launcher (it used to exit)."""
    bench = _bench_module()
    cmd = bench.rank_launch_command(8, 12345, ["--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    # end to end on this GPU-less container: the launcher starts 2 ranks, every rank refuses to run without a GPU
    # (no CPU fallback), the launcher hands the failure on and prints no JSON line
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0
        assert "needs an MI355X" in out.stderr.decode()
        assert not [l for l in out.stdout.decode().split("\n") if l.strip().startswith("{")]


def test_sharded_workload_specs():
    bench = _bench_module()
    c4, c5 = bench.sharded_workload("c4"), bench.sharded_workload("c5")
    assert (c4["height"], c4["width"], c4["cnn_dtype"], c4["total"]) == (32, 32, "fp32", 1 << 20)
    assert (c5["height"], c5["width"], c5["cnn_dtype"], c5["total"]) == (64, 64, "fp16", 1 << 18)
    assert c4["bytes_per_patch"] == 32768 and c5["bytes_per_patch"] == 131072
    assert c4["bytes_per_patch"] * c4["total"] == c5["bytes_per_patch"] * c5["total"]       # the same 34 GB of input
    assert bench.sharded_workload("c5", 4096)["total"] == 4096


def _c5_worker(rank, world, port, n_total, q):
    """One rank of the configs[4] leg with the fp64 oracle (fp16-CNN emulation) standing in for the HIP kernel: the same
    `timed_sharded_evaluations` loop, `shard_range` blocks and (seed, global patch index) data contract bench.py uses."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from noise_flow_amd.ckpt import load_checkpoint
    from noise_flow_amd.dist import timed_sharded_evaluations
    from oracle import philox
    from oracle.nf_oracle import NoiseFlowOracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc = NoiseFlowOracle("sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc",
                          load_checkpoint(os.path.join(ROOT, "models", "NoiseFlow", "ckpt", "model.ckpt.best")), cnn_dtype="fp16")
    seen = []

    def eval_chunk(first, count, acc):
        seen.append((first, count))
        x, y = philox.synth_patches(0, first, count, 64, 64)
        nll, sd, _ = orc.nll(x, y, 100.0, 2.0)
        acc += torch.tensor([float(np.sum(nll)), float(sd) * count, float(count)], dtype=torch.float64)
    res = timed_sharded_evaluations(eval_chunk, n_total, n_total, rank, world, 1, 0, lambda: torch.zeros(3, dtype=torch.float64))
    q.put((rank, res["results"][0], seen))
    dist.destroy_process_group()


def test_c5_leg_sharding_gloo():
    """BASELINE configs[4] sharded over 2 ranks = the unsharded evaluation of the same 64x64 patch range."""
    import torch.multiprocessing as mp
    from noise_flow_amd.ckpt import load_checkpoint
    from noise_flow_amd.patches import shard_range
    from oracle import philox
    from oracle.nf_oracle import NoiseFlowOracle
    n_total, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c5_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    orc = NoiseFlowOracle("sdn5|unc|unc|unc|unc|gain4|unc|unc|unc|unc",
                          load_checkpoint(os.path.join(ROOT, "models", "NoiseFlow", "ckpt", "model.ckpt.best")), cnn_dtype="fp16")
    x, y = philox.synth_patches(0, 0, n_total, 64, 64)
    nll, sd, _ = orc.nll(x, y, 100.0, 2.0)
    for rank, (mean, msd, n), seen in res:
        a, b = shard_range(n_total, rank, world)
        assert seen == [(a, b - a)] and n == n_total
        assert abs(mean - float(np.mean(nll))) <= 1e-9 * abs(float(np.mean(nll)))


def test_image_loader_stage_and_tuple_post_processing():
    """sidd/ImageLoader.py + sidd_utils.load_one_tuple_images on in-memory 'files' (the h5py reader is replaceable): Bayer
    packing, NaN / clip clean-up, the NOISE layer as 'in', the NLF floor, ISO / camera from the scene directory name, the
    fn key, and the requeue of filename tuples for further epochs — feeding PatchSampler / MiniBatchSampler downstream."""
    import queue
    from noise_flow_amd.patches import pack_raw
    from noise_flow_amd.samplers import ImageLoader, MiniBatchSampler, PatchSampler, load_one_tuple_images
    rng = np.random.RandomState(3)
    files = {}
    tuples = []
    for k, (cam, iso) in enumerate((("S6", 100), ("GP", 3200))):
        sdir = "%04d_%03d_%s_%05d_00060_3200_L" % (k + 1, k + 1, cam, iso)
        assert len(sdir) == 30
        base = "/data/SIDD_Medium_Raw/Data/%s/%s" % (sdir, "raw/" if k == 0 else "")     # with and without the sub-directory level
        gt = rng.rand(64, 64)
        noisy = gt + 0.01 * rng.randn(64, 64)
        noisy[0, 0] = np.nan
        noisy[1, 1] = 1.7
        files[base + "NOISY_RAW_010.MAT"] = noisy
        files[base + "GT_RAW_010.MAT"] = gt
        tuples.append((base + "NOISY_RAW_010.MAT", base + "GT_RAW_010.MAT", "", base + "METADATA_RAW_010.MAT"))

    def read_meta(path):
        tags = np.empty((8, 1), object)
        tags[7, 0] = (None, None, [np.asarray([0.0012, -3.0, 9.9])])      # get_nlf: UnknownTags[7, 0][2][0][0:2]
        return {"UnknownTags": tags}
    loader = lambda ft: load_one_tuple_images(ft, read_raw=files.__getitem__, read_metadata=read_meta)   # noqa: E731
    noise, gt, var, nlf0, nlf1, iso, cam, meta = loader(tuples[1])
    assert noise.shape == (1, 32, 32, 4) and gt.shape == (1, 32, 32, 4) and var == []
    assert (iso, cam) == (3200.0, 1.0) and nlf0 == 0.0012 and nlf1 == 1e-6              # non-positive NLF -> 1e-6
    want_gt = np.clip(pack_raw(files[tuples[1][1]]), 0, 1)
    want_in = np.clip(np.nan_to_num(pack_raw(files[tuples[1][0]])), 0, 1)
    assert np.array_equal(gt[0], want_gt) and np.array_equal(noise[0], want_in - want_gt)
    n0 = loader(tuples[0])[0]
    assert n0[0, 0, 0, 0] == -loader(tuples[0])[1][0, 0, 0, 0]                          # NaN -> 0 before the subtraction
    fq = queue.Queue()
    for ft in tuples:
        fq.put(ft)
    il = ImageLoader(fq, max_queue_size=4, n_threads=1, requeue=True, loader=loader)
    ims = [il.get_queue().get(timeout=30) for _ in range(5)]                            # 2.5 epochs
    assert [im["fn"] for im in ims] == ["%s|NOISY_RAW_010.MAT" % t[0].split("/")[-3] for t in (tuples * 3)[:5]]   # parts[-3] | parts[-1]
    assert ims[0]["fn"].startswith("0001_001_S6_00100") and ims[1]["fn"].startswith("Data|") and ims[0]["cam"] == 2.0
    assert set(ims[0]) == {"in", "gt", "vr", "nlf0", "nlf1", "iso", "cam", "fn", "metadata"} and ims[1]["iso"] == 3200.0
    ps = PatchSampler(il.get_queue(), patch_height=16, sampling="uniform", n_threads=1, n_pat_per_im=4, shuffle=False)
    ms = MiniBatchSampler(ps.get_queue(), minibatch_size=4, n_threads=1)
    mb = ms.get_queue().get(timeout=30)
    assert mb["_x"].shape == (4, 16, 16, 4) and mb["_x"].dtype == np.float64 and len(mb["iso"]) == 1
    for st in (ms, ps, il):
        st.close()
