#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the NeuS render hot path, 512 rays x 128 samples (+32 outside)
per GPU per step (BASELINE.json metric), synthetic 800x800 scene, random-init weights of the
shipped architecture.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One JSON line on rank 0 with the driver's contract plus `roofline` (dominant kernel, timed live with
HIP events) and `cpu_baseline` (the oracle = CPU restatement of the reference, timed on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "vdn-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

# algorithmic FLOPs (2 per MAC, GEMM work only) - SURVEY.md 8d
F_SDF, F_SDF1, F_GRAD, F_COL, F_VDN, F_NERF, F_NERF_DPT = 1049088, 918016, 918016, 542720, 590336, 1208320, 1232896
FLOP_PER_RAY_FWD = 112 * F_SDF1 + 128 * (F_SDF + F_GRAD) + 128 * F_COL + 160 * F_NERF          # 617 406 464
FLOP_PER_RAY_TRAIN = 112 * F_SDF1 + 3 * (128 * (F_SDF + F_GRAD) + 128 * F_COL + 160 * F_NERF)  # 1 646 583 808
PEAK = {"f32": 157.3e12, "bf16": 2.5e15}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def time_kernel(fn, iters=10):
    """Average device time of fn() (which launches on torch's current stream) via HIP events."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        e0[i].record()
        fn()
        e1[i].record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in zip(e0, e1)])) * 1e-3


def cpu_baseline(B, seed, wdepth=False, max_seconds=30.0):
    """The oracle's training step (render forward + loss + autograd backward) on the host cores: a bounded sample of the same
    workload - half a batch (256 rays x (64+64+32) samples) per iteration, as many iterations as fit in ~30 s after a small
    warm-up - reported in rays/s. Adam is excluded (negligible against a multi-second step)."""
    import oracle.neus_oracle as orc
    from vdn_train import synth
    st = synth.make_all_states(seed, wdepth=wdepth)
    cams = synth.make_cameras(seed)
    tt = torch.tensor
    nets = orc.nets_from_numpy(st, requires_grad=True)
    params = [p for _, p in orc.all_params(nets)]
    # 16 threads: the fastest setting for this oracle on the GPU box's 2 x EPYC host (tests/probes/cpu_threads.py: 8 -> 106,
    # 16 -> 120, 32 -> 102, 64 -> 62, 128 -> 27 rays/s); more threads only add synchronisation on these tensor sizes
    prev_threads = torch.get_num_threads()
    cores = min(int(os.environ.get("VDN_CPU_THREADS", "16")), os.cpu_count() or 1)
    torch.set_num_threads(cores)

    def one(n, step):
        o, d = synth.random_pixel_batch(seed, step, 0, n, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, step, n)
        lkw = dict(gt_feats=tt(synth.uniform(seed, "bench/feats", (n, 96)).astype(np.float32)), depth_ramp=0.5) if wdepth else {}
        out = orc.render(nets, tt(o), tt(d), tt(near), tt(far), background_rgb=torch.ones(1, 3), cos_anneal_ratio=0.5,
                         t_rand=tt(t1), t_rand_out=tt(t2))
        lo = orc.loss_from_render(out, tt(synth.target_colors(o, d)), **lkw)
        torch.autograd.grad(lo["loss"], params, allow_unused=True)

    one(32, 0)                                  # warm-up: thread pool, allocator
    n = max(1, B // 2)
    times, t_begin = [], time.time()
    while not times or (time.time() - t_begin + float(np.mean(times)) < max_seconds and len(times) < 5):
        t = time.time()
        one(n, 1 + len(times))
        times.append(time.time() - t)
    med = float(np.median(times))
    torch.set_num_threads(prev_threads)
    return {"value": n / med, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "%d x (render forward + loss + backward) of %d rays (64+64+32 samples each), oracle fp32 on %d threads, "
                      "median; %.0f s of CPU work" % (len(times), n, cores, sum(times))}


def spawn_ranks(n):
    """Run this script as n ranks under torch.distributed.run (one per GPU) as a CHILD process; returns its exit code."""
    import socket
    import subprocess
    if torch.cuda.device_count() < n and os.environ.get("VDN_DIST_BACKEND", "nccl") == "nccl":
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-all-samples", action="store_true", help="skip the second timed leg (work lists off); keeps a rocprof trace of "
                    "this command to one population of launches per kernel")
    ap.add_argument("--config", choices=["womsk_white", "womsk_white_wdepth"], default="womsk_white",
                    help="womsk_white = BASELINE.json configs[1] (the headline); womsk_white_wdepth = configs[2] (VDN head + depth-feature loss)")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="bf16 = BASELINE.json's headline config (bf16 MFMA, fp32 accumulate); fp32 = the parity path")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start N fresh ranks BEFORE anything touches the GPU (a process that
        # has initialised HIP must never exec) and relay rank 0's line. The driver's own torch.distributed.run launch sets
        # WORLD_SIZE and skips this.
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with `python -m torch.distributed.run --nproc-per-node %d bench.py "
              "--gpus %d ...` (or plain `python bench.py --gpus %d`)" % (args.gpus, world, args.gpus, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VDN_DIST_BACKEND", "nccl")       # "gloo": exercise the N > 1 control flow on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vdn_train import synth, factory
    from vdn_train.trainer import Trainer
    seed, B = 0, args.batch
    wdepth = args.config == "womsk_white_wdepth"
    st = synth.make_all_states(seed, wdepth=wdepth)
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=st, precision=args.precision)
    # wdepth: the depth-feature loss is live from the first timed step (dpt_runner.py:236 with depth_start_iter behind us)
    trainer = Trainer(rend, B, dev, conf=dict(extract_depth=True, depth_start_iter=-1) if wdepth else None, world_size=world, rank=rank)
    gt_feats = None
    cams = synth.make_cameras(seed)
    perm = np.argsort(synth.uniform(seed, "perm", (len(cams),)))
    bg = torch.ones(1, 3, device=dev)
    g = lambda x: torch.tensor(x).to(dev)
    if wdepth:
        gt_feats = g(synth.uniform(seed, "bench/feats/%d" % rank, (B, 96)).astype(np.float32))

    def batch(step):
        o, d = synth.random_pixel_batch(seed, step, int(perm[step % len(perm)]), B, rank=rank, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        return g(o), g(d), g(near), g(far), g(synth.target_colors(o, d))

    batches = [batch(s) for s in range(args.warmup + args.steps)]     # resident in HBM before the timed region

    def step(i):
        # one iteration of dpt_runner.py:197-259: sample -> render -> loss -> backward -> (all-reduce) -> Adam
        return trainer.train_step(*batches[i], gt_feats=gt_feats)

    for i in range(args.warmup):
        step(i)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.time()
    for i in range(args.steps):
        out = step(args.warmup + i)
    fence()
    dt = time.time() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    loss_final = float(out[0].item())
    # The same K steps with every sample evaluated, as the reference does: the default path skips samples that enter the loss
    # only through exact zeros ("Work lists") - identical results, reported side by side for transparency.
    dt_all = None
    if not args.no_all_samples:
        os.environ["VDN_FG_COMPACT"] = os.environ["VDN_BG_COMPACT"] = "0"
        for i in range(min(2, args.warmup + args.steps)):
            step(i)
        fence()
        t0 = time.time()
        for i in range(args.steps):
            step(args.warmup + i)
        fence()
        dt_all = time.time() - t0
        if world > 1:
            import torch.distributed as dist
            tmax = torch.tensor([dt_all], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_all = float(tmax.item())
        del os.environ["VDN_FG_COMPACT"], os.environ["VDN_BG_COMPACT"]
        step(0)                                   # restore the work lists of a default step for the kernel timings below
    # forward-only render() throughput on the same rays (inference path), reported beside the headline
    with torch.no_grad():
        for i in range(2):
            rend.render(*batches[i][:4], background_rgb=bg, cos_anneal_ratio=0.5)
        fence()
        t1 = time.time()
        nf = max(5, args.steps // 2)
        for i in range(nf):
            rend.render(*batches[i % len(batches)][:4], background_rgb=bg, cos_anneal_ratio=0.5)
        fence()
        fwd_rays_per_s = world * B * nf / (time.time() - t1)

    if rank == 0:
        rays = world * B * args.steps
        value = rays / dt
        flop_per_ray = FLOP_PER_RAY_TRAIN + (3 * (128 * F_VDN + 160 * (F_NERF_DPT - F_NERF)) if wdepth else 0)
        # The fused SDF-MLP kernel (north-star kernel): PE -> 9 layers -> sdf/feature + analytic gradient sweep on the
        # 65 536 render_core points, timed with HIP events exactly as it is launched inside the timed training step
        # (training-mode activation saves included), so it agrees with the rocprofv3 average of the same command.
        eng = trainer.engine
        o, d = batches[0][0], batches[0][1]
        tk = time_kernel(lambda: eng._sdf_forward(o, d))
        # the training step's launch covers the foreground work list of the last step (inside samples within the relaxed
        # sphere; the others enter the loss through exact zeros), the inference launch all 65 536 points
        fg_rows = int(eng.w["fg_active"][1].item())
        flops_train = (F_SDF + F_GRAD) * fg_rows
        flops = (F_SDF + F_GRAD) * eng.P
        dtype = "f32" if args.precision == "fp32" else "bf16"
        # the same kernel without the training saves (what render() launches under torch.no_grad())
        with torch.no_grad():
            tk_inf = time_kernel(lambda: rend.sdf_network._run(1, rays=(o, d, eng.w["mid_z"])))
        # the longest kernel of the step: the batched weight-gradient GEMM, HBM-bound (every saved plane read once)
        tdw = time_kernel(lambda: eng._launch_dw())
        dw_bytes, dw_flops = eng.dw_bytes(), eng.dw_flops()
        # HBM bytes per launch of the fused kernel from the L2 memory-side PMC counters (FETCH_SIZE / WRITE_SIZE in separate
        # rocprofv3 --pmc passes, gfx950 x2 correction on FETCH_SIZE): collected by tools/collect_traffic.sh, not live
        def traffic_of(tag):
            tf = os.path.join(ROOT, "profiles", "r01_traffic_sdf_fwd_%s%s.json" % (args.precision, tag))
            return json.load(open(tf)).get("hbm_bytes_per_launch") if os.path.exists(tf) else None
        traffic, traffic_inf = traffic_of("_train"), traffic_of("")
        if traffic is not None:
            traffic *= fg_rows / float(eng.P)        # PMC figure is for a 65 536-point launch; bytes scale with the rows
        tdwf = os.path.join(ROOT, "profiles", "r01_traffic_dw_gemm_%s.json" % dtype)
        traffic_dw = json.load(open(tdwf)).get("hbm_bytes_per_launch") if (os.path.exists(tdwf) and not wdepth) else None
        line = {
            "metric": "rays/sec (512-ray batch, 128 samples/ray)", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "training step of %s (hierarchical sampling + render forward + backward + gradient "
                                   "all-reduce + Adam): SDF 8x256 + colour 4x256 %s+ NeRF 8x256, 512 rays x (64 coarse + 64 importance "
                                   "+ 32 outside) per GPU per step" % (args.config, "+ VDN head 4x256->96 " if wdepth else ""),
                       "rays_per_gpu": B, "samples_per_ray": 128, "outside_samples": 32, "parallelism": "dp%d" % world,
                       "flop_per_ray": flop_per_ray, "allreduce_bytes": trainer.param_flat.numel() * 4,
                       # flop_per_ray is SURVEY.md 8d's algorithmic count (all 160 background samples per ray); background
                       # samples that render_core multiplies by zero (inside the unit sphere) are not evaluated
                       "background_points_evaluated_last_step": int(eng.w["bg_active"][1].item()),
                       "background_points_total": eng.Q,
                       "foreground_points_evaluated_last_step": fg_rows, "foreground_points_total": eng.P},
            "model_flops_per_s": value * flop_per_ray,
            "forward_only_rays_per_s": fwd_rays_per_s, "final_loss": loss_final,
            "all_samples_evaluated": None if dt_all is None else {
                "value": world * B * args.steps / dt_all, "ms_per_step": dt_all / args.steps * 1e3,
                "note": "same steps with VDN_FG_COMPACT=0 VDN_BG_COMPACT=0 (no zero-weight samples skipped)"},
            "roofline": {"bound": "mfma", "kernel": "sdf_fwd_kernel<%s> (fused PE + SDF MLP + gradient sweep, "
                                                   "training-mode launch of the timed step over its foreground work list)" % ("F32,1,4,false" if dtype == "f32" else "BF16,1,4,true"),
                         "achieved": flops_train / tk / 1e12, "peak": PEAK[dtype] / 1e12, "unit": "TFLOP/s",
                         "frac": flops_train / tk / PEAK[dtype], "traffic": traffic, "kernel_ms": tk * 1e3, "points": fg_rows,
                         "inference_launch": {"kernel_ms": tk_inf * 1e3, "achieved": flops / tk_inf / 1e12, "frac": flops / tk_inf / PEAK[dtype],
                                              "traffic": traffic_inf}},
            "roofline_dw_gemm": {"bound": "hbm", "kernel": "dw_gemm_%s_kernel (batched split-K weight-gradient GEMM, longest kernel of the step)" % dtype,
                                 "achieved": dw_bytes / tdw / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": dw_bytes / tdw / 8e12,
                                 "traffic": traffic_dw, "kernel_ms": tdw * 1e3, "tflops": dw_flops / tdw / 1e12},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(B, seed, wdepth)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                 # rank 0 is still timing single kernels / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
