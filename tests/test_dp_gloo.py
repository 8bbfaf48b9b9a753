"""Data-parallel recipe (vdn_train/dp.py) on CPU with gloo, world_size 2: ray shards + eikonal
numerator/denominator all-reduce + the gradient all-reduced in the Trainer's three slices over its two
communicators (dp.Collectives, the object the Trainer itself drives) == one process on the concatenated
batch. Compute is the oracle (this is a test of the host-side DP logic, not of the kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle.neus_oracle as orc
from vdn_train import dp, synth

B_PER_RANK, WORLD, SEED = 6, 2, 17


def _batch(rank):
    cams = synth.make_cameras(SEED)
    px = np.floor(synth.uniform(SEED, "dp/x%d" % rank, (B_PER_RANK,)) * 400) + 200
    py = np.floor(synth.uniform(SEED, "dp/y%d" % rank, (B_PER_RANK,)) * 400) + 200
    o, d = synth.pixel_rays(cams[0], px, py)
    near, far = synth.near_far_from_sphere(o, d)
    t1, t2 = synth.jitter(SEED, 0, B_PER_RANK, rank=rank)
    return [torch.tensor(x) for x in (o, d, near, far, t1, t2, synth.target_colors(o, d))]


def _render(nets, b):
    o, d, near, far, t1, t2, rgb = b
    return orc.render(nets, o, d, near, far, background_rgb=torch.ones(1, 3), cos_anneal_ratio=0.5, t_rand=t1, t_rand_out=t2)


def _feats(rank):
    return torch.tensor(synth.uniform(SEED, "dp/feats%d" % rank, (B_PER_RANK, 96)).astype(np.float32))


RAMP = 0.37        # depth_iter_weight() of the step (dpt_runner.py:167-171, 242)


def _worker(rank, port, q, wdepth):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    nets = orc.nets_from_numpy(synth.make_all_states(SEED, wdepth=wdepth), requires_grad=True)
    b = _batch(rank)
    out = _render(nets, b)
    nd = torch.stack([out["eik_num"].detach(), out["eik_den"].detach()])
    local_num = nd[0].clone()
    if wdepth:                                              # (one of the two cases takes round 4's form of the gradient sums:
        os.environ["VDN_DP_INSTREAM"] = "0"                 # every sum through begin / finish on the backend's stream)
    coll = dp.Collectives(WORLD)                            # the Trainer's own collectives object (vdn_train/trainer.py:97)
    assert coll.enabled and coll.side_group is not coll.group
    # a second Trainer of the same process (bench.py builds one per leg) shares the side communicator: one new_group per process
    coll2 = dp.Collectives(WORLD)
    assert coll2.side_group is coll.side_group and dp.groups_created == 1
    coll.finish(coll.begin([nd]), tag="eikonal")            # in place: global (num, den)
    eik = (out["eik_num"] + (nd[0] - local_num)) / (nd[1] + 1e-5)     # gradient flows through the local numerator only
    loss = (out["color_fine"] - b[6]).abs().sum() / (B_PER_RANK + 1e-5) / WORLD + 0.1 * eik / WORLD * WORLD
    # note: the eikonal term is already the GLOBAL value; each rank differentiates only its own numerator
    named = orc.all_params(nets)
    local = (out["color_fine"] - b[6]).abs().sum() / (B_PER_RANK + 1e-5) / WORLD + 0.1 * eik
    if wdepth:      # the VDN depth-feature term is a per-rank mean like the colour term: scaled by 1/W (dpt_runner.py:239-242)
        local = local + RAMP * (out["render_feats"] - _feats(rank)).abs().sum() / (B_PER_RANK + 1e-5) / WORLD
    gs = torch.autograd.grad(local, [p for _, p in named], allow_unused=True)
    flat = torch.cat([(torch.zeros_like(p) if gr is None else gr).reshape(-1) for (_, p), gr in zip(named, gs)])
    # remove the double-counted constant part: d/dtheta of (nd[0]-local_num) is zero, so nothing to fix
    # the gradient as the Trainer reduces it: three slices of the flat buffer - [background network] and [heads] through the
    # side communicator, [SDF network + variance] through the main one (trainer.py: update_rest / update_sdf)
    names = [n for n, _ in named]
    sizes = np.cumsum([0] + [p.numel() for _, p in named])
    sb = int(sizes[next(i for i, n in enumerate(names) if n.startswith("sdf."))])
    se = int(sizes[names.index("variance") + 1])
    # (in the order the Trainer's host thread issues them: each slice summed in place, in the stream that made it)
    assert coll.instream == (not wdepth)
    coll.sum_now([flat[:sb]], side=True, tag="grad_nerf")
    coll.sum_now([flat[sb:se]], tag="grad_sdf")
    coll.sum_now([flat[se:]], side=True, tag="grad_heads")
    if rank == 0:
        q.put((flat.numpy(), float(eik)))
    dp.shutdown()                                           # the side communicator goes before the world group
    assert dp._shared_side_group is None
    dist.destroy_process_group()


@pytest.mark.parametrize("wdepth", [False, True])
def test_dp_equals_single_process_on_concatenated_batch(wdepth):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q, wdepth)) for r in range(WORLD)]
    for p in procs:
        p.start()
    flat, eik = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    # single process, concatenated batch
    nets = orc.nets_from_numpy(synth.make_all_states(SEED, wdepth=wdepth), requires_grad=True)
    bs = [_batch(r) for r in range(WORLD)]
    cat = [torch.cat([bs[r][i] for r in range(WORLD)], 0) for i in range(7)]
    out = _render(nets, cat)
    # mean of the per-rank colour means == colour mean of the union for equal shards (mask_sum = B + 1e-5 each)
    col = sum((out["color_fine"][r * B_PER_RANK:(r + 1) * B_PER_RANK] - bs[r][6]).abs().sum() / (B_PER_RANK + 1e-5) for r in range(WORLD)) / WORLD
    loss = col + 0.1 * out["gradient_error"]
    if wdepth:
        loss = loss + RAMP * sum((out["render_feats"][r * B_PER_RANK:(r + 1) * B_PER_RANK] - _feats(r)).abs().sum() / (B_PER_RANK + 1e-5)
                                 for r in range(WORLD)) / WORLD
    named = orc.all_params(nets)
    gs = torch.autograd.grad(loss, [p for _, p in named], allow_unused=True)
    ref = torch.cat([(torch.zeros_like(p) if gr is None else gr).reshape(-1) for (_, p), gr in zip(named, gs)]).numpy()
    assert abs(eik - out["gradient_error"].item()) < 1e-6 * abs(eik)
    assert flat.shape == ref.shape == ((1718879,) if wdepth else (1409087,))     # the all-reduce payloads of SURVEY.md 8e
    if wdepth:      # the VDN head's gradient is in the payload and is not zero
        assert np.abs(ref[1409087 + 12384:]).max() > 0
    assert np.abs(flat - ref).max() < 1e-5 * np.abs(ref).max()
