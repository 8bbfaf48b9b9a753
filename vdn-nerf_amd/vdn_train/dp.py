"""Ray-sharded data parallelism: one process per GPU, full replica of all networks,
each rank renders its own 512 rays; per step ONE all-reduce of the flat gradient buffer (RCCL over
xGMI on the GPUs, gloo in the CPU tests) plus a 2-scalar all-reduce for the eikonal term, which is a
ratio of sums over the global batch (renderer.py:313-315).

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


def shard_seed(step, rank):
    """Disjoint pixel streams by rank from a shared seed: (step, rank) keys vdn_train.synth.random_pixel_batch."""
    return step, rank
