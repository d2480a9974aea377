"""Data parallelism for the renderer path: rays are independent units, so a global ray batch is split
contiguously across ranks (one process per GPU) and the only exchange step is one all-reduce of the
flat gradient buffer (RCCL over xGMI on the GPU box; gloo in the CPU tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_rays: int, rank: int, world_size: int):
    """Contiguous [begin, end) of the global batch owned by `rank` (rank r renders rays r*B/G..(r+1)*B/G)."""
    if n_rays % world_size != 0:
        raise ValueError(f"global ray batch {n_rays} is not divisible by world size {world_size}")
    per = n_rays // world_size
    return rank * per, (rank + 1) * per


def shard_batch(batch: dict, rank: int, world_size: int, n_rays: int | None = None):
    """Slices every per-ray tensor of a batch dict.  Tensors whose leading dim is n_rays are split on
    dim 0; [L, n_rays, ...] tensors (per-ray lights, targets) on dim 1; everything else is replicated."""
    if n_rays is None:
        n_rays = batch["rays_o"].shape[0]
    lo, hi = shard_range(n_rays, rank, world_size)
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n_rays:
            out[k] = v[lo:hi].contiguous()
        elif torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == n_rays:
            out[k] = v[:, lo:hi].contiguous()
        else:
            out[k] = v
    return out


# Optional timing of the step's collectives (bench.py, N > 1): while enabled, every all-reduce issued through
# `all_reduce_sum` is bracketed by two events on the current stream (the collective is waited for on that stream before
# the call returns, so the interval covers it).  Off by default: no events, no overhead.
_TIMED = None


def time_collectives(enable: bool):
    """Starts (True) or stops (False) recording event pairs around the data-parallel all-reduces of this process."""
    global _TIMED
    _TIMED = [] if enable else None


def collective_ms():
    """(summed milliseconds, number of collectives) recorded since `time_collectives(True)`; call after a device
    synchronisation.  Clears the record."""
    global _TIMED
    if not _TIMED:
        return 0.0, 0
    ms, n = sum(a.elapsed_time(b) for a, b in _TIMED), len(_TIMED)
    _TIMED = []
    return ms, n


def all_reduce_sum(t: torch.Tensor, group=None):
    """dist.all_reduce(SUM) in place; timed when `time_collectives(True)` is in effect and `t` lives on a GPU."""
    if _TIMED is not None and t.is_cuda:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        b.record()
        _TIMED.append((a, b))
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_mean_(flat: torch.Tensor, group=None):
    """In-place mean over ranks of one flat fp32 buffer (675,771 floats = 2.7 MB for the full model)."""
    all_reduce_sum(flat, group)
    flat.mul_(1.0 / dist.get_world_size(group))
    return flat


def allreduce_sum_(flat: torch.Tensor, group=None):
    """In-place sum over ranks (the exact large-batch gradient when every rank's loss is its additive share)."""
    return all_reduce_sum(flat, group)


def local_mask_count(mask: torch.Tensor, use_mask: bool) -> torch.Tensor:
    """This shard's [sum(mask > 0.5), rays] as a 2-float device tensor (the summands of `global_mask_count`;
    `rnb_loss(group=)` all-reduces them together with the eikonal partial sums in one collective)."""
    mk = mask.reshape(-1)
    n = torch.tensor(float(mk.numel()), device=mk.device)
    return torch.stack([(mk > 0.5).sum().to(torch.float32) if use_mask else n, n])


def global_mask_count(mask: torch.Tensor, use_mask: bool, group=None) -> torch.Tensor:
    """[sum(mask > 0.5), rays] of the WHOLE data-parallel batch as a 2-float tensor on the mask's device: the
    normalisers of the colour term (exp_runner.py:194, mask_sum) and of the BCE mean (exp_runner.py:251).  With
    `use_mask` False (mask_weight == 0, exp_runner.py:233-236) every ray counts.  One tiny all-reduce."""
    cnt = local_mask_count(mask, use_mask)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return cnt


def grid_slab(resolution: int, rank: int, world_size: int):
    """x-planes of an extract_fields grid owned by `rank`: (planes per rank, first plane, one past the last).  Every rank
    gets ceil(res / world) planes except the last ones, which get what is left — possibly nothing (models/renderer.py:10-25
    evaluates the grid in 64^3 blocks; here the unit is an x-slab)."""
    per = (resolution + world_size - 1) // world_size
    return per, min(rank * per, resolution), min((rank + 1) * per, resolution)


def gather_grid_slabs(slab: torch.Tensor, resolution: int, group=None):
    """All-gather of the ranks' x-slabs ([per, res, res], rows beyond a short slab zero) into the whole volume
    [res, res, res] on every rank."""
    world = dist.get_world_size(group)
    full = torch.empty((world * slab.shape[0],) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(full, slab.contiguous(), group=group)
    return full[:resolution]


def broadcast_parameters(modules, src=0, group=None):
    for m in modules:
        for p in m.parameters():
            dist.broadcast(p.data, src=src, group=group)
