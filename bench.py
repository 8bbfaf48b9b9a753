#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the NeuS render hot path, 512 rays x 128 samples (+32 outside)
per GPU per step (BASELINE.json metric), synthetic 800x800 scene, random-init weights of the
shipped architecture.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, or spawned here)

One JSON line on rank 0 with the driver's contract plus
  roofline      the fused SDF kernel as the timed step launches it (HIP events on the launch stream, in situ; `schedule` says on which
                schedule - the one-stream one of the committed kernel trace - with that schedule's own step time), beside its inference
                launch; traffic = PMC bytes from the committed profile of the same launch;
  roofline_in_step_two_streams   the same bracket on the DEFAULT schedule, the one `ms_per_step` is measured on;
  runner_flow   what an unchanged dpt_runner.py executes: render() + loss.backward() + torch.optim.Adam through the drop-in classes,
                and its image loops' render() with autograd on;
  real_cameras  the step on a camera rig the reference ships (tests/golden/pnf_rays.npz);
  parity_path   the SAME step on the fp32 kernels (the path that holds the 1e-4 tolerance), driver-timed in this run;
  wdepth        the womsk_white_wdepth step (VDN head + depth-feature loss);
  all_samples_evaluated  the step with the zero-weight work lists off;
  trials        every timed region is K steps bracketed by barrier + synchronize; `value` is the median region of at
                least 5 regions / 1 s of timed GPU work, the spread is reported;
  cpu_baseline  the oracle (CPU restatement of the reference) timed on this host.
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
PEAK = {"f32": 157.3e12, "bf16": 2.5e15}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_ROUND = "r06"      # profiles/<round>_traffic_*.json: the PMC traffic figures quoted in `roofline.traffic`


def _profile_file(suffix):
    """profiles/<round>_<suffix> of the newest round that has it (this round's counters are collected after its first bench)."""
    n = int(PROFILE_ROUND[1:])
    for r in range(n, 0, -1):
        f = os.path.join(ROOT, "profiles", "r%02d_%s" % (r, suffix))
        if os.path.exists(f):
            return f
    return None


def flop_per_ray(wdepth, fg_frac=1.0, bg_frac=1.0):
    """SURVEY.md 8d's training-step convention (no-grad sampler x1, differentiable part x3) with the fraction of the
    foreground (SDF + heads) / background (NeRF++) points the step actually evaluates."""
    fg = 128 * (F_SDF + F_GRAD + F_COL + (F_VDN if wdepth else 0))
    bg = 160 * (F_NERF_DPT if wdepth else F_NERF)
    return 112 * F_SDF1 + 3 * (fg * fg_frac + bg * bg_frac)


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


def physical_cores():
    """Physical cores of this host: unique (physical id, core id) pairs of /proc/cpuinfo; os.cpu_count() where that fails."""
    try:
        pairs, phys = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                pairs.add((phys, ln.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def time_kernel_stats(fn, iters=20):
    """Device time of fn() over `iters` launches (HIP events on the launch stream): median / mean / min / max in seconds."""
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
    t = np.array([a.elapsed_time(b) for a, b in zip(e0, e1)]) * 1e-3
    return {"median": float(np.median(t)), "mean": float(t.mean()), "min": float(t.min()), "max": float(t.max()), "launches": iters}


def cpu_baseline(B, seed, wdepth=False, full=False, budget_s=75.0):
    """The oracle (CPU restatement of the reference, verified equal to it: tests/test_oracle_golden.py) on the host cores,
    SURVEY.md 8d's protocol: batches of B = 512 rays x (64 + 64 + 32) samples, fp32, warm-up iterations first, the MEDIAN of the
    timed ones; forward only (render) and forward + backward (render + loss + autograd; Adam excluded: negligible against a
    multi-second step); at the thread count that is fastest on this host AND on all physical cores; plus the C1 reading of
    BASELINE.json configs[0] (512 rays x 64 samples: n_importance = 0) and the womsk_white_wdepth shapes.
    `full`: 3 warm-up + 5 timed iterations per figure (minutes of CPU time; --cpu-baseline-full, committed under profiles/).
    Default: a bounded sample of the same workload (1 warm-up + 3 timed for the headline figure, 1 + 2 / 1 + 1 for the
    others, within about `budget_s` seconds), so that the default bench run stays within a few minutes."""
    import oracle.neus_oracle as orc
    from vdn_train import synth
    cams = synth.make_cameras(seed)
    tt = torch.tensor
    prev_threads = torch.get_num_threads()
    phys = physical_cores()
    # the fastest setting for this oracle on the GPU box's 2 x EPYC host (tests/probes/cpu_threads.py: 8 -> 106, 16 -> 120,
    # 32 -> 102, 64 -> 62, 128 -> 27 rays/s): more threads only add synchronisation on these tensor sizes
    best = min(int(os.environ.get("VDN_CPU_THREADS", "16")), os.cpu_count() or 1)
    t_begin = time.time()
    nets_cache = {}

    def nets_of(wd):
        if wd not in nets_cache:
            nets_cache[wd] = orc.nets_from_numpy(synth.make_all_states(seed, wdepth=wd), requires_grad=True)
        return nets_cache[wd]

    def one(n, step, wd, backward, conf):
        nets = nets_of(wd)
        o, d = synth.random_pixel_batch(seed, step, 0, n, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        t1, t2 = synth.jitter(seed, step, n)
        kw = dict(background_rgb=torch.ones(1, 3), cos_anneal_ratio=0.5, t_rand=tt(t1), t_rand_out=tt(t2), conf=conf)
        if not backward:
            with torch.no_grad():        # (the oracle's SDF gradient is analytic: no autograd needed for the forward)
                orc.render(nets, tt(o), tt(d), tt(near), tt(far), **kw)
            return
        out = orc.render(nets, tt(o), tt(d), tt(near), tt(far), **kw)
        lkw = dict(gt_feats=tt(synth.uniform(seed, "bench/feats", (n, 96)).astype(np.float32)), depth_ramp=0.5) if wd else {}
        lo = orc.loss_from_render(out, tt(synth.target_colors(o, d)), **lkw)
        torch.autograd.grad(lo["loss"], [p for _, p in orc.all_params(nets)], allow_unused=True)

    def figure(threads, wd, backward, conf, warm, timed):
        torch.set_num_threads(threads)
        for i in range(warm):
            one(B, i, wd, backward, conf)
        ts = []
        for i in range(timed):
            t = time.time()
            one(B, warm + i, wd, backward, conf)
            ts.append(time.time() - t)
        return {"rays_per_s": B / float(np.median(ts)), "s_per_batch_median": float(np.median(ts)), "threads": threads,
                "warmup": warm, "timed": timed}

    C3 = orc.RendererConf()                                  # 64 coarse + 64 importance + 32 outside, as shipped
    C1 = orc.RendererConf(n_importance=0)                    # BASELINE.json configs[0] reading: 512 rays x 64 samples
    torch.set_num_threads(best)
    one(32, 0, wdepth, True, C3)                             # thread pool, allocator (cold first call)
    w5 = (3, 5) if full else None
    out = {}
    out["fwd_bwd"] = figure(best, wdepth, True, C3, *(w5 or (1, 3)))
    out["fwd_only"] = figure(best, wdepth, False, C3, *(w5 or (1, 2)))
    left = lambda: budget_s - (time.time() - t_begin)
    if full or left() > 12:
        out["c1_64_samples_fwd_bwd"] = figure(best, wdepth, True, C1, *(w5 or (0, 1)))
    if (full or left() > 12) and not wdepth:
        out["wdepth_fwd_bwd"] = figure(best, True, True, C3, *(w5 or (0, 1)))
    if phys != best and (full or left() > 0.6 * B / 25.0):    # all physical cores: ~25 rays/s on the GPU host => ~20 s per batch
        out["all_physical_cores_fwd_bwd"] = figure(phys, wdepth, True, C3, *(w5 or (0, 1)))
    torch.set_num_threads(prev_threads)
    spent = time.time() - t_begin
    return {"value": out["fwd_bwd"]["rays_per_s"], "unit": "rays/s", "cores": best, "kind": "port",
            "physical_cores": phys, "logical_cpus": os.cpu_count(),
            "sample": "oracle fp32, batches of %d rays x (64+64+32) samples: render + loss + autograd backward, median of %d timed "
                      "batches after %d warm-up on %d threads (the fastest thread count on this host; `all_physical_cores_fwd_bwd` "
                      "= the same on all %d physical cores); %.0f s of CPU work in all%s"
                      % (B, out["fwd_bwd"]["timed"], out["fwd_bwd"]["warmup"], best, phys, spent,
                         "" if full else " (bounded sample of SURVEY.md 8d's 3 + 5 protocol: --cpu-baseline-full runs all of it)"),
            "figures": out}


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


class Leg:
    """One configuration of the training step (precision x config) with its own renderer, trainer and resident batches."""

    def __init__(self, args, dev, world, rank, precision, wdepth, n_batches, crop=None, cams=None, focal=None):
        """cams / focal: another camera rig than the synthetic one (c2w [n,4,4], focal length in pixels of the 800 x 800 frame)."""
        from vdn_train import synth, factory
        from vdn_train.trainer import Trainer
        self.world, self.rank, self.dev, self.B, self.wdepth, self.precision = world, rank, dev, args.batch, wdepth, precision
        seed = 0
        st = synth.make_all_states(seed, wdepth=wdepth)
        self.rend = factory.build_renderer(wdepth=wdepth, device=dev, states=st, precision=precision)
        # wdepth: the depth-feature loss is live from the first timed step (dpt_runner.py:236 with depth_start_iter behind us)
        self.trainer = Trainer(self.rend, self.B, dev, conf=dict(extract_depth=True, depth_start_iter=-1) if wdepth else None,
                               world_size=world, rank=rank)
        cams = synth.make_cameras(seed) if cams is None else cams
        perm = np.argsort(synth.uniform(seed, "perm", (len(cams),)))
        g = lambda x: torch.tensor(x).to(dev)
        self.gt_feats = g(synth.uniform(seed, "bench/feats/%d" % rank, (self.B, 96)).astype(np.float32)) if wdepth else None
        focal = synth.FOCAL if focal is None else focal

        def batch(step):
            o, d = synth.random_pixel_batch(seed, step, int(perm[step % len(perm)]), self.B, rank=rank, cams=cams, crop=crop, focal=focal)
            near, far = synth.near_far_from_sphere(o, d)
            return g(o), g(d), g(near), g(far), g(synth.target_colors(o, d))
        self.batches = [batch(s) for s in range(n_batches)]     # resident in HBM before any timed region

    def step(self, i):
        # one iteration of dpt_runner.py:197-259: sample -> render -> loss -> backward -> (all-reduce) -> Adam
        self.steps_done = getattr(self, "steps_done", 0) + 1
        return self.trainer.train_step(*self.batches[i % len(self.batches)], gt_feats=self.gt_feats)

    def sdf_in_situ(self, n_steps=60):
        """The fused SDF kernel timed INSIDE `n_steps` consecutive training steps: HIP events on the launch stream right around
        each step's own launch (TrainEngine.sdf_probe), with that step's row count. -> mean / median / min / max and mean rows."""
        eng = self.trainer.engine
        self.fence()
        eng.sdf_probe = []
        for i in range(n_steps):
            self.step(self.steps_done)
        self.fence()
        probe, eng.sdf_probe = eng.sdf_probe, None
        t = np.array([a.elapsed_time(b) for a, b, _ in probe]) * 1e-3
        rows = np.array([eng.P if r is None else int(r.item()) for _, _, r in probe], dtype=np.float64)
        return {"kernel_ms": float(t.mean() * 1e3), "kernel_ms_median": float(np.median(t) * 1e3), "kernel_ms_min": float(t.min() * 1e3),
                "kernel_ms_max": float(t.max() * 1e3), "points": float(rows.mean()), "points_min": float(rows.min()),
                "points_max": float(rows.max()), "steps": n_steps, "workgroups_mean": float(np.ceil(rows / 128.0).mean()),
                # (bf16 default: the colour head rides in the same launch - csrc/k_sdf_fwd2.h MODE 3 - and is counted with it)
                "color_head_in_the_launch": bool(getattr(eng, "_color_fused", False)),
                "flops": float((F_SDF + F_GRAD + (F_COL if getattr(eng, "_color_fused", False) else 0)) * rows.mean())}

    def fence(self):
        torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def region(self, first, k):
        """Exactly k steps bracketed by barrier + synchronize; MAX over ranks (seconds)."""
        self.fence()
        t0 = time.time()
        for i in range(k):
            out = self.step(first + i)
        self.fence()
        dt = time.time() - t0
        if self.world > 1:
            import torch.distributed as dist
            tmax = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        self.last = out
        return dt

    def measure(self, warmup, steps, min_trials=5, min_seconds=1.0, max_trials=2000):
        for i in range(warmup):
            self.step(i)
        regions = []
        # (the loop condition only uses max-reduced times, so every rank runs the same number of regions)
        while len(regions) < min_trials or (sum(regions) < min_seconds and len(regions) < max_trials):
            regions.append(self.region(warmup + len(regions) * steps, steps))
        med = float(np.median(regions))
        eng = self.trainer.engine
        # mean work-list rows of the steady state: one more (untimed) pass over ALL resident batches - the lists differ by +-10 %
        # between images, and every mean that is compared with another (the kernel statistics under profiles/ quote this one:
        # roofline fractions are recomputed from rows / AverageUs) has to cover whole cycles of them - with the device-side row
        # counters copied out per step
        counts = []
        n_cycle = len(self.batches)
        for i in range(n_cycle):
            self.step(warmup + len(regions) * steps + i)
            counts.append((eng.w["fg_active"][1].clone(), eng.w["bg_active"][1].clone() if "bg_active" in eng.w else None))
        self.fence()
        rows_fg = float(np.mean([int(a.item()) for a, _ in counts]))
        rows_bg = float(np.mean([int(b.item()) for _, b in counts])) if counts[0][1] is not None else 0.0
        fg = int(eng.w["fg_active"][1].item())
        bg = int(eng.w["bg_active"][1].item()) if "bg_active" in eng.w else 0
        rays = self.world * self.B * steps
        fpr = flop_per_ray(self.wdepth, fg / float(eng.P), bg / float(eng.Q) if eng.Q else 1.0)
        return {"value": rays / med, "ms_per_step": med / steps * 1e3,
                "trials": {"regions": len(regions), "steps_per_region": steps, "timed_seconds": float(sum(regions)),
                           "ms_per_step_min": min(regions) / steps * 1e3, "ms_per_step_max": max(regions) / steps * 1e3},
                "work_list_rows_mean": {"foreground": rows_fg, "background": rows_bg, "over_steps": n_cycle},
                "foreground_points_evaluated_last_step": fg, "foreground_points_total": eng.P,
                "background_points_evaluated_last_step": bg, "background_points_total": eng.Q,
                "executed_flop_per_ray": fpr, "executed_model_flops_per_s": rays / med * fpr,
                "final_loss": float(self.last[0].item())}

    def sdf_kernel_roofline(self, n_lists=6, one_stream=None, in_situ_steps=60):
        """The fused SDF-MLP kernel (north-star kernel): PE -> 9 layers -> sdf/feature + analytic gradient sweep.
        `kernel_ms` / `points` / `frac` are IN SITU: HIP events around the launch inside `in_situ_steps` consecutive timed
        training steps - of `one_stream` (a Leg on the one-stream schedule VDN_SIDE_STREAM=0 VDN_OVERLAP=0, the schedule the
        committed rocprofv3 kernel trace profiles/<round>_train_bf16_kernel_stats.csv is taken on: its AverageUs is this number)
        when given, else of this leg as scheduled. Beside it: the same launch inside the steps of the default two-stream schedule
        (`in_step_two_streams`: the background network's kernels share the chip with it), the old isolated timing (`isolated`:
        the step's launch repeated back to back on an otherwise idle chip, over the lists of `n_lists` steps), the inference
        launch and the training launch over all rows (medians of 20 isolated launches). Every rank runs the steps (they hold
        collectives); only rank 0 reports."""
        eng, rend = self.trainer.engine, self.rend
        dtype = "f32" if self.precision == "fp32" else "bf16"
        in_situ_steps = max(1, int(round(in_situ_steps / float(len(self.batches))))) * len(self.batches)     # whole cycles of the batches
        situ_two = self.sdf_in_situ(in_situ_steps)
        situ = situ_two
        if one_stream is not None:
            situ = one_stream.sdf_in_situ(in_situ_steps)
        t_sum, rows_sum, per_list = 0.0, 0, []
        for j in range(n_lists):
            self.step(j)
            self.fence()
            if self.rank == 0:
                o, d = self.batches[j % len(self.batches)][0], self.batches[j % len(self.batches)][1]
                rows = int(eng.w["fg_active"][1].item())
                t = time_kernel(lambda: eng._sdf_forward(o, d), iters=4)
                t_sum, rows_sum = t_sum + t, rows_sum + rows
                per_list.append({"points": rows, "kernel_ms": t * 1e3})
        self.step(0)
        self.fence()
        if self.rank != 0:
            return None
        o, d = self.batches[0][0], self.batches[0][1]
        with torch.no_grad():
            tk_inf = time_kernel_stats(lambda: rend.sdf_network._run(1, rays=(o, d, eng.w["mid_z"])))
            # SURVEY.md 8d's configuration C2 on the same 65 536 points: K1 + K3 + K5 + K6 = PE + SDF MLP + gradient sweep, colour
            # head, NeuS alpha + compositing (NeuSRenderer._shade, renderer.py:239-315), timed as render() issues it, end to end
            bgc = torch.ones(3, device=self.dev)
            tk_c2 = time_kernel_stats(lambda: rend._shade(o, d, eng.w["dists"], eng.w["mid_z"], None, bgc, 0.5))
            c2_launches = getattr(rend, "shade_launches", lambda: 4 if self.wdepth else 3)()
        # the training-mode launch over ALL rows (what the step launches on a scene whose samples all lie inside the relaxed
        # sphere, and in the all-samples leg): no work list
        fgc, eng._fg_compact = eng._fg_compact, False
        tk_full = time_kernel_stats(lambda: eng._sdf_forward(o, d))
        eng._fg_compact = fgc
        eng._sdf_forward(o, d)                    # leave the workspace as the step left it

        def traffic_of(tag):
            tf = _profile_file("traffic_sdf_fwd_%s%s.json" % (self.precision, tag))
            if tf is None:
                return None, None
            j = json.load(open(tf))
            return j.get("hbm_bytes_per_launch"), os.path.relpath(tf, ROOT)
        tr_train, src_train = traffic_of("_train")
        tr_inf, src_inf = traffic_of("")
        rows_mean, tk = rows_sum / float(n_lists), t_sum / n_lists
        tr_full = tr_train
        if tr_train is not None:
            tr_train *= situ["points"] / float(eng.P)      # PMC figure is for a 65 536-point launch; bytes scale with the rows
        F1 = F_SDF + F_GRAD
        fused_col = bool(getattr(eng, "_color_fused", False))
        F1t = F1 + (F_COL if fused_col else 0)          # the training launch: + the colour head where it rides in it
        pk = PEAK[dtype]
        fl_inf = F1 * eng.P
        name = ("sdf_fwd_kernel<F32,1,4,false>" if dtype == "f32" else
                ("sdf2::sdf_fwd2_kernel<3,true,4,3> (csrc/k_sdf_fwd2.h MODE 3: + the colour head, %d FLOP/row)" % F1t if fused_col else
                 "sdf2::sdf_fwd2_kernel<1,true,4,3> (csrc/k_sdf_fwd2.h)"))
        c2f = (F1 + F_COL + (F_VDN if self.wdepth else 0)) * eng.P
        return {"bound": "mfma",
                "kernel": name + ": fused PE + SDF MLP + gradient sweep, training-mode launch of the timed step over its foreground "
                                 "work list, timed IN SITU: HIP events around the launch inside %d consecutive training steps (%s)"
                                 % (in_situ_steps, "one-stream schedule, as the committed kernel trace" if one_stream is not None else "as scheduled"),
                "achieved": situ["flops"] / (situ["kernel_ms"] * 1e-3) / 1e12, "peak": pk / 1e12, "unit": "TFLOP/s",
                "frac": situ["flops"] / (situ["kernel_ms"] * 1e-3) / pk,
                "traffic": tr_train, "traffic_source": src_train, "kernel_ms": situ["kernel_ms"], "points": situ["points"],
                "in_situ": situ,
                "in_step_two_streams": dict(situ_two, frac=situ_two["flops"] / (situ_two["kernel_ms"] * 1e-3) / pk),
                "isolated": {"what": "the step's launch repeated 4 x back to back on an idle chip, mean over %d steps' lists (rounds 1-3 quoted "
                                     "this as roofline.frac)" % n_lists,
                             "kernel_ms": tk * 1e3, "points": rows_mean, "frac": F1t * rows_mean / tk / pk, "per_list": per_list},
                "inference_launch": {"kernel_ms": tk_inf["median"] * 1e3, "kernel_ms_min": tk_inf["min"] * 1e3, "kernel_ms_max": tk_inf["max"] * 1e3,
                                     "points": eng.P, "achieved": fl_inf / tk_inf["median"] / 1e12,
                                     "frac": fl_inf / tk_inf["median"] / pk, "traffic": tr_inf, "traffic_source": src_inf,
                                     "what": "median of 20 isolated launches"},
                "training_launch_full_rows": {"kernel_ms": tk_full["median"] * 1e3, "kernel_ms_min": tk_full["min"] * 1e3,
                                              "kernel_ms_max": tk_full["max"] * 1e3, "points": eng.P,
                                              "achieved": F1t * eng.P / tk_full["median"] / 1e12, "frac": F1t * eng.P / tk_full["median"] / pk,
                                              "flop_per_row": F1t,
                                              "traffic": tr_full, "what": "median of 20 isolated launches"},
                "c2_forward": {"what": "SURVEY.md 8d C2: PE + SDF MLP + gradient sweep + colour%s head + NeuS alpha / compositing on %d points, "
                                       "as render() launches them (%d launch%s), median of 20"
                                       % (" + VDN" if self.wdepth else "", eng.P, c2_launches, "" if c2_launches == 1 else "es"),
                               "launches": c2_launches,
                               "chain_ms": tk_c2["median"] * 1e3, "chain_ms_min": tk_c2["min"] * 1e3, "points": eng.P,
                               "points_per_s": eng.P / tk_c2["median"],
                               "achieved": c2f / tk_c2["median"] / 1e12, "frac": c2f / tk_c2["median"] / pk}}

    def sdf_roofline_in_step(self, in_situ_steps=40):
        """N > 1: the north-star kernel timed only where the job runs it - HIP events around the launch inside consecutive training
        steps of THIS leg as scheduled (every rank steps: the steps hold collectives); no isolated launches, no second leg."""
        in_situ_steps = max(1, int(round(in_situ_steps / float(len(self.batches))))) * len(self.batches)
        situ = self.sdf_in_situ(in_situ_steps)
        if self.rank != 0:
            return None
        dtype = "f32" if self.precision == "fp32" else "bf16"
        pk = PEAK[dtype]
        tf = _profile_file("traffic_sdf_fwd_%s_train.json" % self.precision)
        tr = json.load(open(tf)).get("hbm_bytes_per_launch") if tf is not None else None
        return {"bound": "mfma", "kernel": "fused PE + SDF MLP + gradient sweep, training-mode launch of the timed step over its foreground work "
                                           "list, HIP events around the launch inside %d consecutive steps of rank 0" % in_situ_steps,
                "achieved": situ["flops"] / (situ["kernel_ms"] * 1e-3) / 1e12, "peak": pk / 1e12, "unit": "TFLOP/s",
                "frac": situ["flops"] / (situ["kernel_ms"] * 1e-3) / pk,
                "traffic": tr * situ["points"] / float(self.trainer.engine.P) if tr is not None else None,
                "kernel_ms": situ["kernel_ms"], "points": situ["points"],
                "schedule": "default two-stream schedule of the data-parallel step (the schedule of ms_per_step), rank 0", "in_situ": situ}

    def dw_roofline(self):
        eng = self.trainer.engine
        dtype = "f32" if self.precision == "fp32" else "bf16"
        tdw = time_kernel(lambda: eng._launch_dw_groups())       # two launches: the SDF network's entries, then the rest
        dw_bytes, dw_flops = eng.dw_bytes(), eng.dw_flops()
        tdwf = _profile_file("traffic_dw_gemm_%s.json" % dtype)
        traffic = json.load(open(tdwf)).get("hbm_bytes_per_launch") if (tdwf is not None and not self.wdepth) else None
        return {"bound": "hbm", "kernel": "dw_gemm_%s_kernel (batched split-K weight-gradient GEMM, longest kernel of the step; its two launches - "
                                          "SDF entries, the rest - timed back to back)" % dtype,
                "achieved": dw_bytes / tdw / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": dw_bytes / tdw / 8e12,
                "traffic": traffic, "kernel_ms": tdw * 1e3, "tflops": dw_flops / tdw / 1e12}


def runner_flow(args, dev, precision, steps, warmup=5, wdepth=False):
    """What an UNCHANGED dpt_runner.py executes on the drop-in classes (dpt_runner.py:117-144, 197-257): networks from the conf's
    kwargs, torch.optim.Adam over .parameters(), per iteration render() under grad (one autograd node) -> the runner's torch
    loss -> zero_grad -> loss.backward() -> optimizer.step(); the jitter from torch.rand inside render(). The path of
    tests/test_gpu_runner_flow.py, timed: K steps bracketed by synchronize, median of >= 5 regions."""
    import torch.nn.functional as F
    from vdn_train import synth, factory
    seed = 0
    rend = factory.build_renderer(wdepth=wdepth, device=dev, states=synth.make_all_states(seed, wdepth=wdepth), precision=precision)
    params = rend._all_parameters()
    opt = torch.optim.Adam(params, lr=5e-4)
    cams = synth.make_cameras(seed)
    g = lambda x: torch.tensor(x).to(dev)
    B = args.batch
    batches = []
    for s in range(16):
        o, d = synth.random_pixel_batch(seed, s, s % len(cams), B, cams=cams)
        near, far = synth.near_far_from_sphere(o, d)
        batches.append((g(o), g(d), g(near), g(far), g(synth.target_colors(o, d))))
    bg = torch.ones([1, 3], device=dev)
    gt_feats = torch.rand(B, 96, device=dev) if wdepth else None

    def step(i):
        rays_o, rays_d, near, far, true_rgb = batches[i % len(batches)]
        mask = torch.ones(B, 1, device=dev)
        mask_sum = mask.sum() + 1e-5
        out = rend.render(rays_o, rays_d, near, far, background_rgb=bg, cos_anneal_ratio=0.5, depth_before_color=False)
        color_error = (out["color_fine"] - true_rgb) * mask
        color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction="sum") / mask_sum
        mask_loss = F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
        loss = color_fine_loss + out["gradient_error"] * 0.1 + mask_loss * 0.0
        if wdepth:          # womsk_white_wdepth behind depth_start_iter (dpt_runner.py:239-243)
            depth_feat_error = (out["render_feats"] - gt_feats) * mask
            loss = loss + F.l1_loss(depth_feat_error, torch.zeros_like(depth_feat_error), reduction="sum") / mask_sum * 0.5
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    for i in range(warmup):
        step(i)
    regions = []
    while len(regions) < 5 or (sum(regions) < 1.0 and len(regions) < 200):
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(steps):
            loss = step(len(regions) * steps + i)
        torch.cuda.synchronize()
        regions.append(time.time() - t0)
    med = float(np.median(regions))
    # the runner's image loops (dpt_runner.py:439-445, 540-546) call render() on 512-ray chunks with autograd ON and never call
    # backward: the forward of the step above, node and saves included
    fwd = []
    while len(fwd) < 5 or (sum(fwd) < 0.5 and len(fwd) < 200):
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(steps):
            b = batches[i % len(batches)]
            out = rend.render(b[0], b[1], b[2], b[3], background_rgb=bg, cos_anneal_ratio=0.5)
            keep = out["color_fine"].detach()
            del out
        torch.cuda.synchronize()
        fwd.append(time.time() - t0)
    fmed = float(np.median(fwd))
    return {"rays_per_s": B * steps / med, "ms_per_step": med / steps * 1e3, "regions": len(regions), "final_loss": float(loss.item()),
            "image_loop_rays_per_s": B * steps / fmed, "image_loop_ms_per_batch": fmed / steps * 1e3,
            "dtype": "f32" if precision == "fp32" else "bf16"}


def precision_gap(head, dev, n_batches=4):
    """bf16 against fp32 on the SAME rays, weights and jitter: the headline leg's networks as its timed steps left them, copied into
    a second renderer on the exact-fp32 kernels; render() without grad on `n_batches` of the leg's resident batches, the jitter
    injected (t_rand / t_rand_out) so that both draw the same samples. Per output: max |bf16 - fp32| / max |fp32| and
    mean |bf16 - fp32| / mean |fp32| over all rays of all batches; gradient_error is one scalar per batch (its relative error)."""
    from vdn_train import factory
    r16 = head.rend
    head.trainer.join()                 # (the heads' / background network's parameters are updated on the Trainer's side stream)
    r32 = factory.build_renderer(wdepth=head.wdepth, device=dev, states=None, precision="fp32")
    with torch.no_grad():
        for name in ("nerf", "sdf_network", "deviation_network", "color_network", "depth_network"):
            src, dst = getattr(r16, name), getattr(r32, name)
            if src is not None:
                dst.load_state_dict({k: v.detach().clone() for k, v in src.state_dict().items()})
    gen = torch.Generator(device="cpu").manual_seed(1234)
    bg = torch.ones(1, 3, device=dev)
    keys = ("color_fine", "weight_sum", "gradients", "weights") + (("render_feats",) if head.wdepth else ())
    num_max, den_max, num_sum, den_sum = ({k: 0.0 for k in keys} for _ in range(4))
    eik = []
    with torch.no_grad():
        for i in range(n_batches):
            o, d, near, far = head.batches[i % len(head.batches)][:4]
            t1 = torch.rand(o.shape[0], 1, generator=gen).to(dev)
            t2 = torch.rand(o.shape[0], r16.n_outside, generator=gen).to(dev)
            a = r16.render(o, d, near, far, background_rgb=bg, cos_anneal_ratio=0.5, t_rand=t1, t_rand_out=t2)
            b = r32.render(o, d, near, far, background_rgb=bg, cos_anneal_ratio=0.5, t_rand=t1, t_rand_out=t2)
            for k in keys:
                x, y = a[k].double(), b[k].double()
                num_max[k] = max(num_max[k], float((x - y).abs().max()))
                den_max[k] = max(den_max[k], float(y.abs().max()))
                num_sum[k] += float((x - y).abs().sum())
                den_sum[k] += float(y.abs().sum())
            ea, eb = float(a["gradient_error"]), float(b["gradient_error"])
            eik.append(abs(ea - eb) / max(abs(eb), 1e-30))
    inv_s = float(torch.exp(r16.deviation_network.variance.detach() * 10.0).item())
    out = {"what": "bf16 kernels against the fp32 parity kernels on the same %d x %d rays, weights (the headline leg's, after its timed "
                   "steps) and jitter: max |diff| / max |fp32| and mean |diff| / mean |fp32| per render() output" % (n_batches, head.B),
           "inv_s": inv_s, "gradient_error_rel_err_max": float(max(eik)), "gradient_error_rel_err_mean": float(np.mean(eik))}
    for k in keys:
        out[k + "_max_rel_err"] = num_max[k] / max(den_max[k], 1e-30)
        out[k + "_mean_rel_err"] = num_sum[k] / max(den_sum[k], 1e-30)
    del r32
    torch.cuda.empty_cache()
    return out


def real_cameras():
    """A camera rig the reference SHIPS: the 33 learned poses and the focal coefficient of pretrained-models/pixiu/
    womsk_learn_white_colmap/pnf_300000.pth, as the reference's own LearnPose / LearnIntrin return them (tests/golden/pnf_rays.npz,
    written by tests/golden/make_golden.py::pnf_fixture) -> (c2w [33,4,4], focal in pixels of an 800-px-wide frame)."""
    f = os.path.join(ROOT, "tests", "golden", "pnf_rays.npz")
    if not os.path.exists(f):
        return None
    d = np.load(f, allow_pickle=False)
    tag = "pixiu.womsk_learn_white_colmap"
    return d[tag + "__c2w"].astype(np.float64), float(d[tag + "__fx"]) ** 2 * 800.0         # poses.py:80-84: focal = fx^2 * W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="SURVEY.md 8d's whole CPU protocol (3 warm-up + 5 timed batches of "
                    "512 rays per figure: minutes of CPU time) instead of the bounded sample")
    ap.add_argument("--headline-only", action="store_true", help="only the headline leg: keeps a rocprof trace of this command to one "
                    "population of launches per kernel (no all-samples / fp32 / wdepth legs, no forward-only renders)")
    ap.add_argument("--no-all-samples", action="store_true", help="skip the leg with the zero-weight work lists off")
    ap.add_argument("--no-roofline", action="store_true", help="skip the single-kernel timing launches (with --headline-only a rocprof "
                    "trace of this command then holds nothing but the timed step's own launches)")
    ap.add_argument("--crop", type=int, default=None, help="draw pixels from a centred crop x crop window (object-centric capture) "
                    "instead of the full 800 x 800 frame")
    ap.add_argument("--config", choices=["womsk_white", "womsk_white_wdepth"], default="womsk_white",
                    help="womsk_white = BASELINE.json configs[1] (the headline); womsk_white_wdepth = configs[4] (VDN head + depth-feature loss)")
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

    wdepth = args.config == "womsk_white_wdepth"
    K, W = args.steps, args.warmup
    nb = W + K                                          # distinct resident batches; regions cycle through them
    head = Leg(args, dev, world, rank, args.precision, wdepth, nb, crop=args.crop)
    res = head.measure(W, K)
    extras = {}
    if world > 1:
        # K more steps with HIP events around every collective (dp.Collectives.sum_now / finish): how long each stream actually
        # spent in or waiting for RCCL - the three in-stream gradient slices and the exposed part of the two small overlapped
        # sums - per step, rank 0's view. (Kept out of the timed regions above: `value` carries no event records.)
        head.trainer.coll.timing = True
        head.region(W + K, K)
        head.trainer.coll.timing = False
        # communicators this job made beyond the world group, max over ranks (vdn_train/dp.py: ONE shared side group per process)
        from vdn_train import dp as _dp
        made = torch.tensor([_dp.groups_created], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(made, op=torch.distributed.ReduceOp.MAX)
        extras["dp_process_groups_created_max_over_ranks"] = int(made.item())
        extras["allreduce_exposed_ms"] = dict(head.trainer.coll.exposed_ms(), note="per call, bracketed by HIP events on the issuing stream: "
                                              "the gradient slices are summed IN that stream (the bracket is the collective itself: "
                                              "grad_sdf on the critical path, grad_nerf / grad_heads on the side stream); fg_count and "
                                              "eikonal overlap other work and the bracket is only the stream's wait for them")
    if world > 1:
        # N > 1 is the scaling measurement: the headline leg, the exposed all-reduce time and the north-star kernel inside the
        # job's own steps - no further legs (each would be another Trainer and minutes of box time on every rank), no per-rank
        # micro-benchmarks. Everything else is reported at N = 1.
        args.headline_only = True
    if not args.headline_only:
        # The same K steps with every sample evaluated, as the reference does: the default path skips samples that enter the loss
        # only through exact zeros ("Work lists") - identical results, reported side by side for transparency.
        if not args.no_all_samples:
            os.environ["VDN_FG_COMPACT"] = os.environ["VDN_BG_COMPACT"] = "0"
            r = head.measure(2, K)
            del os.environ["VDN_FG_COMPACT"], os.environ["VDN_BG_COMPACT"]
            extras["all_samples_evaluated"] = {k: r[k] for k in ("value", "ms_per_step", "trials", "executed_model_flops_per_s")}
            extras["all_samples_evaluated"]["note"] = "same steps with VDN_FG_COMPACT=0 VDN_BG_COMPACT=0 (no zero-weight samples skipped)"
        # forward-only render() throughput on the same rays (inference path)
        bg = torch.ones(1, 3, device=dev)
        with torch.no_grad():
            for i in range(2):
                head.rend.render(*head.batches[i][:4], background_rgb=bg, cos_anneal_ratio=0.5)
            head.fence()
            nf = max(100, K)
            # (eager render() needs ~300 us of host time per call for ~350 us of device time: a busy host shows directly, so three
            # regions, the median, and the host's enqueue time beside it; RenderPlan replays need 75 us of host time)
            rates, host_us = [], []
            for region in range(3):
                t1 = time.time()
                for i in range(nf):
                    head.rend.render(*head.batches[i % nb][:4], background_rgb=bg, cos_anneal_ratio=0.5)
                t2 = time.time()
                head.fence()
                rates.append(world * args.batch * nf / (time.time() - t1))
                host_us.append((t2 - t1) / nf * 1e6)
            extras["forward_only_rays_per_s"] = float(np.median(rates))
            extras["forward_only_detail"] = {"regions_rays_per_s": rates, "host_enqueue_us_per_call": float(np.median(host_us)), "calls_per_region": nf}
            # the image loops may hand render() any batch size: 4 of the resident batches at once
            if nb >= 4:
                big = [torch.cat([head.batches[(4 * j + k) % nb][c] for k in range(4)]) for j in range(2) for c in range(4)]
                big = [big[0:4], big[4:8]]
                for i in range(2):
                    head.rend.render(*big[i], background_rgb=bg, cos_anneal_ratio=0.5)
                head.fence()
                t1 = time.time()
                for i in range(nf // 2):
                    head.rend.render(*big[i % 2], background_rgb=bg, cos_anneal_ratio=0.5)
                head.fence()
                extras["forward_only_rays_per_s_batch_%d" % (4 * args.batch)] = world * 4 * args.batch * (nf // 2) / (time.time() - t1)
    roof = roof_dw = None
    if not args.no_roofline and world > 1:
        roof = head.sdf_roofline_in_step()
    elif not args.no_roofline:
        # the north-star kernel in situ on the one-stream schedule (what the committed rocprofv3 kernel trace shows): a second
        # leg with the same seed, batches and step count (same work lists), built with the side stream and the overlap off
        env1 = {"VDN_SIDE_STREAM": "0", "VDN_OVERLAP": "0"}
        old_env = {k: os.environ.get(k) for k in env1}
        os.environ.update(env1)
        one = Leg(args, dev, world, rank, args.precision, wdepth, nb, crop=args.crop)
        for k, v in old_env.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
        for i in range(head.steps_done):             # to the same training state (the work lists shrink over the first ~600 steps)
            one.step(i)
        one.fence()
        r1 = one.measure(0, K, min_trials=3, min_seconds=0.3)      # the one-stream schedule's own step time, beside its kernel figure
        roof = head.sdf_kernel_roofline(one_stream=one)   # every rank takes part (steps hold collectives); rank 0 gets the numbers
        if roof is not None:
            roof["schedule"] = ("one stream (VDN_SIDE_STREAM=0 VDN_OVERLAP=0: every launch of the step in order; the schedule of the committed "
                                "rocprofv3 kernel trace). NOT the schedule of the line's ms_per_step: see roofline_in_step_two_streams")
            roof["ms_per_step_of_this_schedule"] = r1["ms_per_step"]
        del one
        torch.cuda.empty_cache()
        roof_dw = head.dw_roofline() if rank == 0 else None

    def other_leg(precision, wd, crop=None, roofline=True):
        leg = Leg(args, dev, world, rank, precision, wd, nb, crop=crop)
        r = leg.measure(W, K)
        if roofline:
            r["roofline"] = leg.sdf_kernel_roofline(n_lists=3)
        if world > 1:
            leg.fence()
        del leg
        torch.cuda.empty_cache()
        return r
    if not args.headline_only:
        if args.precision == "bf16":
            # the path that holds the north-star 1e-4 tolerance (tests/test_gpu_parity.py), driver-timed in the same run
            extras["parity_path"] = dict(other_leg("fp32", wdepth), dtype="f32",
                                         note="same step on the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32): the path the 1e-4 parity tests hold on")
        if not wdepth:
            extras["wdepth"] = dict(other_leg(args.precision, True), config="womsk_white_wdepth (VDN head 4x256->96 + depth-feature loss, BASELINE.json configs[4])")
        # what an unchanged dpt_runner.py executes: render() + loss.backward() + torch.optim.Adam through the drop-in classes
        if world == 1:
            extras["runner_flow"] = {"what": "render() under grad -> torch loss -> loss.backward() -> torch.optim.Adam.step() through the drop-in "
                                             "classes (dpt_runner.py:214-257), 512 rays per step, default jitter; image_loop_*: "
                                             "render() with autograd on and no backward, the runner's validate_image / val_img chunks "
                                             "(dpt_runner.py:439-445); bf16_wdepth: womsk_white_wdepth (VDN head + the depth-feature loss, "
                                             "dpt_runner.py:239-243)",
                                     "bf16": runner_flow(args, dev, "bf16", K), "fp32": runner_flow(args, dev, "fp32", max(4, K // 4)),
                                     "bf16_wdepth": runner_flow(args, dev, "bf16", K, wdepth=True)}
            torch.cuda.empty_cache()
        if args.precision == "bf16" and world == 1:
            extras["bf16_vs_fp32"] = precision_gap(head, dev)
        rc = real_cameras()
        if rc is not None and args.crop is None:
            leg = Leg(args, dev, world, rank, args.precision, wdepth, nb, cams=rc[0], focal=rc[1])
            r = leg.measure(W, K)
            del leg
            torch.cuda.empty_cache()
            extras["real_cameras"] = dict({k: r[k] for k in ("value", "ms_per_step", "trials", "work_list_rows_mean", "executed_model_flops_per_s")},
                                          note="the same step with the camera rig of a scene the reference ships (33 learned poses + focal of "
                                               "pretrained-models/pixiu/womsk_learn_white_colmap/pnf_300000.pth, via tests/golden/pnf_rays.npz), "
                                               "pixels uniform over the 800 x 800 frame; networks and targets as in the headline leg")
        if args.crop is None:
            # the same step on an object-centric capture (pixels from the central 420-px window: the object fills the frame, as in
            # the DTU scenes the shipped configs train on): nearly every foreground sample lies inside the relaxed sphere, so the
            # SDF-side kernels get no relief from the work lists, while more background samples drop out
            oc = other_leg(args.precision, wdepth, crop=420)
            oc["note"] = "pixels drawn from the central 420 x 420 window of the 800 x 800 frames"
            extras["object_centric"] = oc

    if rank == 0:
        dtype = "f32" if args.precision == "fp32" else "bf16"
        line = {
            "metric": "rays/sec (512-ray batch, 128 samples/ray)", "value": res["value"], "unit": "rays/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "training step of %s (hierarchical sampling + render forward + backward + gradient "
                                   "all-reduce + Adam): SDF 8x256 + colour 4x256 %s+ NeRF 8x256, 512 rays x (64 coarse + 64 importance "
                                   "+ 32 outside) per GPU per step; pixels uniform over %s; samples that enter the loss through exact "
                                   "zeros are skipped: %.0f %% of the foreground and %.0f %% of the background points evaluated%s"
                                   % (args.config, "+ VDN head 4x256->96 " if wdepth else "",
                                      "the full 800 x 800 frame" if args.crop is None else "the central %d-px window" % args.crop,
                                      100.0 * res["foreground_points_evaluated_last_step"] / res["foreground_points_total"],
                                      100.0 * res["background_points_evaluated_last_step"] / max(res["background_points_total"], 1),
                                      (" (object_centric leg: %.0f %% / %.0f %%)" % (
                                          100.0 * extras["object_centric"]["foreground_points_evaluated_last_step"] / extras["object_centric"]["foreground_points_total"],
                                          100.0 * extras["object_centric"]["background_points_evaluated_last_step"] / max(extras["object_centric"]["background_points_total"], 1)))
                                      if "object_centric" in extras else ""),
                       "rays_per_gpu": args.batch, "samples_per_ray": 128, "outside_samples": 32, "parallelism": "dp%d" % world,
                       "flop_per_ray": flop_per_ray(wdepth), "allreduce_bytes": head.trainer.param_flat.numel() * 4,
                       # flop_per_ray is SURVEY.md 8d's algorithmic count (every sample evaluated); samples that render_core
                       # multiplies by exact zeros are not evaluated: the executed count is below
                       "executed_flop_per_ray": res["executed_flop_per_ray"],
                       "background_points_evaluated_last_step": res["background_points_evaluated_last_step"],
                       "background_points_total": res["background_points_total"],
                       "foreground_points_evaluated_last_step": res["foreground_points_evaluated_last_step"],
                       "foreground_points_total": res["foreground_points_total"]},
            "trials": res["trials"], "work_list_rows_mean": res["work_list_rows_mean"],
            # FLOPs of the points the step actually evaluated (not the all-samples count)
            "model_flops_per_s": res["executed_model_flops_per_s"],
            "final_loss": res["final_loss"],
            "roofline": roof, "roofline_dw_gemm": roof_dw,
            # the north-star kernel inside the steps of THIS line's schedule (default: two streams; what ms_per_step is measured on)
            "roofline_in_step_two_streams": (dict({k: roof["in_step_two_streams"][k] for k in ("frac", "kernel_ms", "points", "steps")},
                                                  schedule="default two-stream schedule = the schedule of ms_per_step: the background network's "
                                                           "kernels share the chip with the launch", ms_per_step=res["ms_per_step"])
                                             if (roof is not None and "in_step_two_streams" in roof) else None),
        }
        line.update(extras)
        if roof is not None:
            # flat copies of the nested figures (a reader that keeps only the scalars of `roofline` still sees them)
            for k in ("inference_launch", "training_launch_full_rows", "c2_forward", "in_step_two_streams"):
                if isinstance(roof.get(k), dict) and "frac" in roof[k]:
                    roof[k + "_frac"] = roof[k]["frac"]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.batch, 0, wdepth, full=args.cpu_baseline_full)
        else:
            line["cpu_baseline"] = None
        # the figures of the other legs once more, short and LAST: whoever keeps only the end of this (long) line still has them
        g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
        rnd = lambda x: None if x is None else (round(x, 1) if abs(x) >= 100 else float("%.4g" % x))
        summary = {"value_rays_per_s": rnd(res["value"]), "ms_per_step": rnd(res["ms_per_step"]),
                   # (`roofline.frac` is timed on the one-stream schedule at N = 1, inside the job's own two-stream steps at N > 1)
                   ("roofline_frac_in_step_one_stream" if world == 1 else "roofline_frac_in_step_two_streams"): rnd(g(roof, "frac")),
                   **({"roofline_frac_in_step_two_streams": rnd(g(roof, "in_step_two_streams", "frac"))} if world == 1 else {}),
                   "roofline_frac_inference_launch": rnd(g(roof, "inference_launch", "frac")),
                   "roofline_frac_training_launch_full_rows": rnd(g(roof, "training_launch_full_rows", "frac")),
                   "roofline_frac_c2_forward_one_launch": rnd(g(roof, "c2_forward", "frac")),
                   "dw_gemm_hbm_frac": rnd(g(roof_dw, "frac")),
                   "runner_flow_bf16_rays_per_s": rnd(g(extras, "runner_flow", "bf16", "rays_per_s")),
                   "runner_flow_bf16_ms_per_step": rnd(g(extras, "runner_flow", "bf16", "ms_per_step")),
                   "runner_flow_fp32_rays_per_s": rnd(g(extras, "runner_flow", "fp32", "rays_per_s")),
                   "runner_flow_bf16_wdepth_rays_per_s": rnd(g(extras, "runner_flow", "bf16_wdepth", "rays_per_s")),
                   "wdepth_rays_per_s": rnd(g(extras, "wdepth", "value")), "object_centric_rays_per_s": rnd(g(extras, "object_centric", "value")),
                   "all_samples_rays_per_s": rnd(g(extras, "all_samples_evaluated", "value")),
                   "real_cameras_rays_per_s": rnd(g(extras, "real_cameras", "value")),
                   "fp32_parity_path_rays_per_s": rnd(g(extras, "parity_path", "value")),
                   "forward_only_rays_per_s": rnd(g(extras, "forward_only_rays_per_s")),
                   "bf16_vs_fp32_color_max_rel_err": rnd(g(extras, "bf16_vs_fp32", "color_fine_max_rel_err")),
                   "bf16_vs_fp32_color_mean_rel_err": rnd(g(extras, "bf16_vs_fp32", "color_fine_mean_rel_err")),
                   "bf16_vs_fp32_weight_sum_max_rel_err": rnd(g(extras, "bf16_vs_fp32", "weight_sum_max_rel_err")),
                   "bf16_vs_fp32_gradient_error_rel_err_max": rnd(g(extras, "bf16_vs_fp32", "gradient_error_rel_err_max")),
                   "allreduce_exposed_ms": ({k: rnd(v["mean_ms"]) for k, v in extras["allreduce_exposed_ms"].items() if isinstance(v, dict)}
                                            if "allreduce_exposed_ms" in extras else None),
                   "cpu_baseline_rays_per_s": rnd(g(line, "cpu_baseline", "value"))}
        line["summary"] = {k: v for k, v in summary.items() if v is not None}
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        from vdn_train import dp
        dist.barrier()                 # rank 0 is still printing: leave together
        dp.shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
