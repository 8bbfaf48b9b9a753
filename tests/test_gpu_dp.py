"""Data parallelism on the real kernels: 2 ranks (gloo over CUDA tensors - RCCL cannot place two ranks on the one GPU of
the test box) each run Trainer.train_step on their own ray shard; the all-reduced flat gradient must equal the gradient a
single process computes on the concatenated batch (eikonal term as a ratio of GLOBAL sums)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B, SEED = 16, 23


def _rays(rank):
    from vdn_train import synth
    cams = synth.make_cameras(SEED)
    px = np.floor(synth.uniform(SEED, "dpg/x%d" % rank, (B,)) * 400) + 200
    py = np.floor(synth.uniform(SEED, "dpg/y%d" % rank, (B,)) * 400) + 200
    o, d = synth.pixel_rays(cams[0], px, py)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(SEED, 0, B, rank=rank)
    return [o, d, near, far, synth.target_colors(o, d), t1, t2]


def _gt_feats(rank):
    from vdn_train import synth
    return synth.uniform(SEED, "dpg/feats%d" % rank, (B, 96)).astype(np.float32)


WDEPTH_CONF = dict(extract_depth=True, depth_start_iter=-1)       # the depth-feature loss is live (dpt_runner.py:239)


def _worker(rank, port, q, wdepth=False, backend="gloo"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vdn-nerf_amd"))
    import torch.distributed as dist
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # gloo: both ranks on the one GPU of the test box; nccl (= RCCL): one GPU per rank
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=2)
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=synth.make_all_states(SEED, wdepth=wdepth))
    tr = Trainer(rend, B, dev, conf=WDEPTH_CONF if wdepth else None, world_size=2, rank=rank)
    g = lambda x: torch.tensor(x).to(dev)
    o, d, near, far, rgb, t1, t2 = _rays(rank)
    sc = tr.train_step(g(o), g(d), g(near), g(far), g(rgb), gt_feats=g(_gt_feats(rank)) if wdepth else None, t_rand=g(t1), t_rand_out=g(t2))
    torch.cuda.synchronize()
    if rank == 0:
        q.put((tr.engine.grad_flat.cpu().numpy(), float(sc[3].item())))
    dist.barrier()
    dist.destroy_process_group()


def _two_rank_vs_single(wdepth, backend):
    import torch.multiprocessing as mp
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q, wdepth, backend)) for r in range(2)]
    for p in procs:
        p.start()
    flat, eik = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=synth.make_all_states(SEED, wdepth=wdepth))
    tr = Trainer(rend, 2 * B, dev, conf=WDEPTH_CONF if wdepth else None)
    g = lambda x: torch.tensor(x).to(dev)
    parts = [_rays(0), _rays(1)]
    cat = [np.concatenate([parts[0][i], parts[1][i]], 0) for i in range(7)]
    feats = g(np.concatenate([_gt_feats(0), _gt_feats(1)], 0)) if wdepth else None
    sc = tr.train_step(g(cat[0]), g(cat[1]), g(cat[2]), g(cat[3]), g(cat[4]), gt_feats=feats, t_rand=g(cat[5]), t_rand_out=g(cat[6]))
    ref = tr.engine.grad_flat.cpu().numpy()
    # the all-reduce payloads of SURVEY.md 8e: 5 636 348 B, 6 875 516 B with the VDN head
    assert flat.shape == ref.shape == ((1718879,) if wdepth else (1409087,))
    if wdepth:
        assert np.abs(ref[-297408:]).max() > 0          # the VDN head's gradient is live
    assert abs(eik - float(sc[3].item())) < 1e-5 * abs(eik)                 # global eikonal term
    assert np.abs(flat - ref).max() < 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("wdepth", [False, True])
def test_two_rank_gradient_equals_single_process_on_concatenated_batch(wdepth):
    _two_rank_vs_single(wdepth, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; the test box has one")
def test_two_rank_gradient_over_rccl():
    """The same equivalence with the gradient all-reduce, the eikonal reduction and the parameter broadcast over RCCL
    (backend "nccl" on ROCm), one GPU per rank."""
    _two_rank_vs_single(True, "nccl")


def _one_rank_worker(port, q, backend):
    """Fresh process, process group first (before anything touches the GPU), then two Trainers in lockstep: one without
    collectives, one with the collective path forced on in the one-rank group."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vdn-nerf_amd"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
    torch.cuda.set_device(dev)
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    conf = dict(WDEPTH_CONF, warm_up_end=10)
    trs = [Trainer(factory.build_renderer(wdepth=True, device=dev, states=synth.make_all_states(SEED, wdepth=True)), B, dev, conf=conf,
                   collectives=c) for c in (None, True)]
    assert not trs[0].coll.enabled and trs[1].coll.enabled
    g = lambda x: torch.tensor(x).to(dev)
    cams = synth.make_cameras(SEED)
    same = True
    for it in range(4):
        o, d = synth.random_pixel_batch(SEED, it, it, B, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(SEED, it, B)
        args = [g(o), g(d), g(near), g(far), g(synth.target_colors(o, d))]
        sc = [tr.train_step(*args, gt_feats=g(_gt_feats(0)), t_rand=g(t1), t_rand_out=g(t2)).clone() for tr in trs]
        same = same and torch.equal(sc[0], sc[1]) and torch.equal(trs[0].engine.grad_flat, trs[1].engine.grad_flat)
        same = same and torch.equal(trs[0].param_flat, trs[1].param_flat)
    torch.cuda.synchronize()
    q.put((bool(same), float(sc[1][0].item()), float(trs[1].engine.grad_flat.abs().max().item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_one_rank_group_runs_the_collective_path_bit_for_bit(backend):
    """RCCL on the one GPU of the test box: a ONE-rank "nccl" process group, the Trainer's collective path forced on
    (Trainer(collectives=True): the early eikonal-sum all-reduce beside the colour head / compositor, the SDF gradient slice on
    the main stream, the other slices on the side stream beside the next sampler, the parameter broadcast). A one-rank
    all-reduce is the identity, so every step must equal the step without collectives bit for bit - losses, gradients,
    parameters - which also exercises the ordering of both streams against RCCL's own stream."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(port, q, backend))
    p.start()
    same, loss, gmax = q.get(timeout=600)
    p.join(timeout=600)
    assert p.exitcode == 0
    assert same and np.isfinite(loss) and gmax > 0


def _steps_batch(rank, it):
    from vdn_train import synth
    cams = synth.make_cameras(SEED)
    o, d = synth.random_pixel_batch(SEED, it, it % len(cams), B, rank=rank, cams=cams)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(SEED, it, B, rank=rank)
    return [o, d, near, far, synth.target_colors(o, d), t1, t2]


MULTI_CONF = dict(extract_depth=True, depth_start_iter=2, warm_up_end=10)     # the depth-feature loss switches on at the 4th step
N_STEPS = 6


def _multi_worker(rank, port, q, fused):
    import hashlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "vdn-nerf_amd"))
    os.environ["VDN_DP_FUSED"] = "1" if fused else "0"
    import torch.distributed as dist
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    rend = factory.build_renderer(wdepth=True, device=dev, states=synth.make_all_states(SEED, wdepth=True), precision="bf16")
    tr = Trainer(rend, B, dev, conf=MULTI_CONF, world_size=2, rank=rank)          # the default schedule: two streams, overlapped
    g = lambda x: torch.tensor(x).to(dev)
    feats = g(_gt_feats(rank))
    digests, scal = [], []
    for it in range(N_STEPS):
        o, d, near, far, rgb, t1, t2 = _steps_batch(rank, it)
        sc = tr.train_step(g(o), g(d), g(near), g(far), g(rgb), gt_feats=feats, t_rand=g(t1), t_rand_out=g(t2))
        scal.append(sc.clone())
        flat = tr.param_flat                        # (joins the side stream)
        torch.cuda.synchronize()
        digests.append(hashlib.md5(flat.cpu().numpy().tobytes()).hexdigest())
    q.put((rank, digests, [x.cpu().numpy() for x in scal], tr.param_flat.cpu().numpy() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fused", [True, False])
def test_two_ranks_keep_identical_parameters_over_overlapped_steps(fused):
    """Six steps of the DEFAULT (two-stream, overlapped) data-parallel schedule on two ranks, VDN head on, the depth-feature loss
    switching on mid-run (the late Adam group starts stepping then): after EVERY step both ranks hold bit-identical parameters
    (replicas never drift: every gradient slice is all-reduced before its Adam step, on whichever stream), and the run stays with
    one process on the concatenated batches (losses to 1e-3, parameters to 1e-4 of their norm: Adam turns last-bit differences of
    near-zero gradients into sign flips of the update, so an element-wise bound does not exist). Both forms of the step: the
    one-launch compositor with the all-reduced foreground count (fused) and the early eikonal all-reduce (VDN_DP_FUSED=0)."""
    import torch.multiprocessing as mp
    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_multi_worker, args=(r, port, q, fused)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, dig, scal, flat = q.get(timeout=900)
        got[r] = (dig, scal, flat)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0], "replicas drifted apart: %s" % [a == b for a, b in zip(got[0][0], got[1][0])]
    assert len(set(got[0][0])) == N_STEPS                           # ... and the parameters did move every step
    # one process, the concatenated batches
    dev = torch.device("cuda:0")
    rend = factory.build_renderer(wdepth=True, device=dev, states=synth.make_all_states(SEED, wdepth=True), precision="bf16")
    tr = Trainer(rend, 2 * B, dev, conf=MULTI_CONF)
    g = lambda x: torch.tensor(x).to(dev)
    feats = g(np.concatenate([_gt_feats(0), _gt_feats(1)], 0))
    for it in range(N_STEPS):
        parts = [_steps_batch(0, it), _steps_batch(1, it)]
        cat = [np.concatenate([parts[0][i], parts[1][i]], 0) for i in range(7)]
        sc = tr.train_step(g(cat[0]), g(cat[1]), g(cat[2]), g(cat[3]), g(cat[4]), gt_feats=feats, t_rand=g(cat[5]), t_rand_out=g(cat[6])).cpu().numpy()
        # [loss, colour, psnr, eikonal, depth, mask]: the eikonal term is global on every rank; colour / depth are per-rank means
        r0, r1 = got[0][1][it], got[1][1][it]
        print("step %d  eikonal dp %.6f / %.6f single %.6f   colour dp %.6f single %.6f" % (it, r0[3], r1[3], sc[3], 0.5 * (r0[1] + r1[1]), sc[1]))
        assert r0[3] == r1[3]                                              # the reported eikonal term is the GLOBAL one on every rank
        # step 0 starts from identical parameters: summation order only; later steps follow parameters that differ in their last bits
        tol = 1e-4 if it == 0 else 2e-2
        assert abs(r0[3] - sc[3]) < tol * abs(sc[3]), (it, r0[3], sc[3])
        assert abs(0.5 * (r0[1] + r1[1]) - sc[1]) < tol * abs(sc[1]), (it, r0[1], r1[1], sc[1])
        assert (sc[4] > 0) == (it > 2) and (r0[4] > 0) == (it > 2)         # the depth term enters at the 4th step, on every rank alike
    ref = tr.param_flat.cpu().numpy()
    dp = got[0][2]
    assert np.linalg.norm(dp - ref) < 1e-4 * np.linalg.norm(ref), np.linalg.norm(dp - ref) / np.linalg.norm(ref)


def _bench(cmd, env=None, with_stderr=False):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return (json.loads(lines[0]), out.stderr) if with_stderr else json.loads(lines[0])


def test_bench_two_rank_control_flow():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per process), with gloo
    and both ranks on the one GPU of this box: the barriers, the gradient all-reduce, the max-over-ranks timing and the
    single JSON line of rank 0. And the plain `python bench.py --gpus 2` form, which must spawn the ranks itself."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VDN_DIST_BACKEND="gloo")
    # the driver's own flags: at N > 1 bench.py runs the headline leg, the exposed all-reduce time and the in-step roofline - nothing else
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    for cmd in ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", "29541"] + tail, [sys.executable] + tail):
        env.pop("WORLD_SIZE", None)
        d, err = _bench(cmd, env, with_stderr=True)
        assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
        assert d["config"]["parallelism"] == "dp2" and d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None
        assert d["trials"]["regions"] >= 5 and d["trials"]["steps_per_region"] == 3
        # one leg: no fp32 / wdepth / object-centric / runner legs at N > 1, and the collectives' exposed time is in the line
        assert not {"parity_path", "wdepth", "object_centric", "runner_flow", "all_samples_evaluated", "real_cameras"} & set(d)
        assert set(d["allreduce_exposed_ms"]) >= {"grad_sdf", "grad_nerf", "grad_heads"}
        # every rank made ONE communicator beyond the world group (vdn_train/dp.py: the shared side group; max over ranks, in the line)
        assert d["dp_process_groups_created_max_over_ranks"] == 1
    # --gpus that does not match the launcher's world size must not print a line for the wrong GPU count
    import subprocess
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, WORLD_SIZE="1"), cwd=root)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_bench_line_carries_every_leg():
    """One default-shaped run (shortened): the driver's contract keys, the executed-FLOP accounting, the trial spread and the
    fp32 parity-path / wdepth / all-samples legs in the same line."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = _bench([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "trials", "parity_path", "wdepth", "all_samples_evaluated", "object_centric"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "bf16" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["trials"]["regions"] >= 5 and d["trials"]["timed_seconds"] >= 1.0
    assert d["parity_path"]["dtype"] == "f32" and d["parity_path"]["value"] > 0 and d["parity_path"]["roofline"]["frac"] > 0
    assert d["wdepth"]["value"] > 0 and d["wdepth"]["value"] < d["value"]
    # model FLOP/s counts executed points only: never above the all-samples convention
    assert d["model_flops_per_s"] <= d["value"] * d["config"]["flop_per_ray"] * (1 + 1e-9)
    assert d["config"]["executed_flop_per_ray"] <= d["config"]["flop_per_ray"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "inference_launch", "training_launch_full_rows"}
    # the object-centric leg evaluates (nearly) every foreground sample and fewer background samples than the full-frame headline
    oc = d["object_centric"]
    assert oc["value"] > 0 and oc["foreground_points_evaluated_last_step"] > d["config"]["foreground_points_evaluated_last_step"]
    assert oc["background_points_evaluated_last_step"] < d["config"]["background_points_evaluated_last_step"]
    # round 6: bf16 against fp32 on the same rays, and the short summary that closes the line
    g = d["bf16_vs_fp32"]
    # (the MAX over 2 048 rays is one outlier ray and depends on the state the two timed steps left: 8e-4 ... 4e-2 seen; the mean is the bound)
    assert 0 < g["color_fine_max_rel_err"] < 0.2 and g["color_fine_mean_rel_err"] < 2e-3
    assert g["weight_sum_max_rel_err"] < 1e-3 and g["gradient_error_rel_err_max"] < 0.2
    assert list(d)[-1] == "summary" and d["summary"]["value_rays_per_s"] > 0 and "runner_flow_bf16_rays_per_s" in d["summary"]
    assert d["roofline"]["inference_launch_frac"] == d["roofline"]["inference_launch"]["frac"]
