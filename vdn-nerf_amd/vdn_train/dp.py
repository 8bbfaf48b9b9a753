"""Ray-sharded data parallelism: one process per GPU, full replica of all networks,
each rank renders its own 512 rays. Per step the ranks exchange
  * (numerator, denominator) of the eikonal term, which is a ratio of sums over the GLOBAL batch
    (renderer.py:313-315) - 8 bytes, reduced while the colour head, the background network and the
    compositor still run;
  * the gradient, as three slices of the flat buffer: [SDF network + variance] on the critical path
    (the next step's sampler needs the updated SDF weights first), [background network] and
    [colour + VDN heads] on the side stream, beside the next step's sampler.
RCCL over xGMI on the GPUs (backend "nccl"), gloo in the CPU tests.

Equivalence with one process on the concatenated batch:
  loss_global = (1/W) sum_r L1_r / mask_sum_r  +  igr * sum_r num_r / (sum_r den_r + 1e-5)
so each rank back-propagates its colour term scaled by 1/W and its eikonal numerator against the
GLOBAL denominator; summing the ranks' gradients gives d loss_global / d theta.
"""
import os

import torch
import torch.distributed as dist


# The second communicator (same ranks as the world) is made ONCE per process and shared by every Trainer on the world group
# (round 6: bench.py builds a Trainer per leg - six RCCL communicators per rank, none destroyed, was the round-5 state).
# dist.new_group is collective over the world, so the first data-parallel Trainer must be constructed by every rank - as before.
_shared_side_group = None
groups_created = 0          # process groups this module has created (tests/test_gpu_dp.py asserts <= 1 per process)


def shared_side_group():
    """The process-wide side communicator over all world ranks: created at the first call (collective: every rank calls it in the
    same order), reused afterwards."""
    global _shared_side_group, groups_created
    if _shared_side_group is None:
        _shared_side_group = dist.new_group(ranks=None)
        groups_created += 1
    return _shared_side_group


def shutdown():
    """Destroy the side communicator (before dist.destroy_process_group(), or when no Trainer of this process needs it again)."""
    global _shared_side_group
    if _shared_side_group is not None and dist.is_initialized():
        try:
            dist.destroy_process_group(_shared_side_group)
        except Exception:       # the world group went first: nothing left to destroy
            pass
    _shared_side_group = None


class Collectives:
    """The collectives of one Trainer, issued asynchronously on whatever torch stream is current at the call.

    torch.distributed orders a collective behind the work already queued on the current stream and runs it on the
    backend's own stream (RCCL) or thread (gloo); `wait()` on the returned handle orders the current stream behind it
    (RCCL: an event wait, the host does not block). So `begin_*` ... other launches ... `finish` overlaps the collective
    with those launches on the device.

    `enabled` is independent of the world size: a one-rank group runs the very same calls (an all-reduce over one rank is
    the identity), which is how the RCCL path is exercised on a single GPU (tests/test_gpu_dp.py).
    """

    def __init__(self, world_size, group=None, force=False, side_group=None):
        """group: the process group of this Trainer's replicas (None = the world). Every rank of the WORLD must construct its
        Trainer when the side communicator is made here (dist.new_group is collective over the world group): for a Trainer on a
        sub-group, create the second communicator yourself - in every world rank - and pass it as `side_group`; without one a
        sub-group Trainer runs all its collectives on `group`."""
        self.group = group
        self.enabled = bool(force) or world_size > 1
        if self.enabled and not dist.is_initialized():
            raise RuntimeError("data-parallel Trainer: torch.distributed is not initialised (init_process_group first)")
        # A process group's collectives run on ONE internal stream in host issue order. The Trainer issues, per step: the
        # background slice (side stream), the SDF slice (critical path: the next sampler waits for it), the heads' slice (side
        # stream), the next step's 8-byte eikonal sums (critical path). In one group the critical ones would queue behind the
        # side-stream slices and their GEMMs; the side-stream slices therefore get a communicator of their own (same ranks).
        # Created by every rank in the same order (dist.new_group is collective), once per process (shared_side_group above).
        # VDN_DP_SIDE_GROUP=0: one group for all.
        self.side_group = group if side_group is None else side_group
        whole_world = group is None or (dist.is_initialized() and group is dist.group.WORLD)
        if self.enabled and side_group is None and whole_world and os.environ.get("VDN_DP_SIDE_GROUP", "1") != "0":
            self.side_group = shared_side_group()
        # The gradient slices are summed IN the stream that made them (sum_now: torch.distributed's blocking form enqueues the
        # RCCL kernel on the current stream - no hop to the backend's stream and back, i.e. two marker packets and two queue
        # switches less per slice, on chains that wait for the sum at once anyway). The two small sums that DO overlap other
        # work (begin / finish: the foreground count under the SDF kernel, the logged eikonal pair) stay on the backend's
        # stream. One-rank RCCL group, same box, 3 alternating runs (profiles/r05_dp_one_rank_instream_ab.log): the step
        # without collectives 1 137 - 1 149 us, through the backend's stream 1 192 - 1 221, in-stream 1 174 - 1 190; a third
        # communicator for the small sums (1 215 - 1 227) and the small sums in-stream too (no different) are not kept.
        # VDN_DP_INSTREAM=0: every sum through begin / finish.
        self.instream = self.enabled and os.environ.get("VDN_DP_INSTREAM", "1") != "0"
        self.timing = False         # bench.py: HIP events around finish() -> exposed wait per tag
        self._timed = {}

    def begin(self, tensors, side=False):
        """Start summing each tensor (contiguous views of the flat buffers) over the ranks -> handles for finish().
        side: a slice that is issued and awaited on the Trainer's side stream (its own communicator)."""
        if not self.enabled:
            return []
        grp = self.side_group if side else self.group
        return [dist.all_reduce(t, group=grp, async_op=True) for t in tensors if t.numel()]

    def sum_now(self, tensors, side=False, tag=None):
        """Sum each tensor over the ranks in place, ordered in torch's current stream like a kernel launch (RCCL: the collective's
        kernel is enqueued on that stream; gloo: the host blocks until it is done). With `timing` on the call is bracketed by
        two events like finish()."""
        if not self.enabled:
            return
        if not self.instream:
            return self.finish(self.begin(tensors, side=side), tag=tag)
        grp = self.side_group if side else self.group
        ts = [t for t in tensors if t.numel()]
        timed = self.timing and tag is not None and ts
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for t in ts:
            dist.all_reduce(t, group=grp)
        if timed:
            e1.record()
            self._timed.setdefault(tag, []).append((e0, e1))

    def finish(self, handles, tag=None):
        """Order the current stream behind the collectives started by begin(). With `timing` on, the stream's wait is bracketed
        by two events: their distance is the EXPOSED part of the collective (what the stream could not hide behind its own
        launches), collected per tag for exposed_ms()."""
        if self.timing and tag is not None and handles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for h in handles:
                h.wait()
            e1.record()
            self._timed.setdefault(tag, []).append((e0, e1))
            return
        for h in handles:
            h.wait()

    def exposed_ms(self, reset=True):
        """Mean exposed wait per finish() tag in ms over the calls since the last reset (call after a device synchronise)."""
        out = {k: {"mean_ms": sum(a.elapsed_time(b) for a, b in v) / len(v), "calls": len(v)} for k, v in self._timed.items() if v}
        if reset:
            self._timed = {}
        return out

    def broadcast(self, flat, src=0):
        if self.enabled:
            dist.broadcast(flat, src, group=self.group)
