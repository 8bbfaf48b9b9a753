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
import torch
import torch.distributed as dist


def global_eikonal(num_den, group=None):
    """num_den: tensor [2] = local (numerator, denominator). All-reduced in place; returns gradient_error."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(num_den, group=group)
    return num_den[0] / (num_den[1] + 1e-5)


def allreduce_flat(flat, group=None):
    """Sum the flat gradient buffer over ranks (colour-term scaling by 1/W is applied upstream)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, group=group)
    return flat


class Collectives:
    """The collectives of one Trainer, issued asynchronously on whatever torch stream is current at the call.

    torch.distributed orders a collective behind the work already queued on the current stream and runs it on the
    backend's own stream (RCCL) or thread (gloo); `wait()` on the returned handle orders the current stream behind it
    (RCCL: an event wait, the host does not block). So `begin_*` ... other launches ... `finish` overlaps the collective
    with those launches on the device.

    `enabled` is independent of the world size: a one-rank group runs the very same calls (an all-reduce over one rank is
    the identity), which is how the RCCL path is exercised on a single GPU (tests/test_gpu_dp.py).
    """

    def __init__(self, world_size, group=None, force=False):
        self.group = group
        self.enabled = bool(force) or world_size > 1
        if self.enabled and not dist.is_initialized():
            raise RuntimeError("data-parallel Trainer: torch.distributed is not initialised (init_process_group first)")

    def begin(self, tensors):
        """Start summing each tensor (contiguous views of the flat buffers) over the ranks -> handles for finish()."""
        if not self.enabled:
            return []
        return [dist.all_reduce(t, group=self.group, async_op=True) for t in tensors if t.numel()]

    @staticmethod
    def finish(handles):
        """Order the current stream behind the collectives started by begin()."""
        for h in handles:
            h.wait()

    def broadcast(self, flat, src=0):
        if self.enabled:
            dist.broadcast(flat, src, group=self.group)
