"""The N>1 path on CPU: world_size-2 gloo processes exercise ray sharding and the flat-gradient
all-reduce helper used by the renderer's backward (RCCL on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import rnb_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rnb_neus_fork_amd.parallel as P
    torch.set_num_threads(1)
    batch = O.synthetic_batch(16, seed=3, step=0, warmup=False)
    mine = P.shard_batch(batch, rank, world)
    lo, hi = P.shard_range(16, rank, world)
    ok = mine["rays_o"].shape[0] == 8 and mine["lights_dir"].shape == (3, 8, 1, 3) \
        and torch.equal(mine["true_rgb"], batch["true_rgb"][:, lo:hi])
    # per-rank "gradient" = sum over the rank's rays of a ray-wise function; mean over ranks must equal
    # half of the single-process sum (the renderer's backward uses the same helper on its flat buffer)
    flat = torch.stack([mine["rays_d"].sum(), mine["near"].sum(), mine["true_rgb"].sum()]).float()
    ref = 0.5 * torch.stack([batch["rays_d"].sum(), batch["near"].sum(), batch["true_rgb"].sum()]).float()
    P.allreduce_mean_(flat)
    ok = ok and torch.allclose(flat, ref, rtol=1e-5, atol=1e-5)
    # exact large-batch semantics: the global normalisers every rank divides by, and the SUM of the
    # per-rank additive shares of the loss == the single-process loss of the whole batch (checked with the oracle's
    # loss terms: shard numerators over global denominators)
    cnt = P.global_mask_count(mine["mask"], True)
    ok = ok and float(cnt[0]) == float((batch["mask"] > 0.5).sum()) and float(cnt[1]) == 16.0
    cnt1 = P.global_mask_count(mine["mask"], False)
    ok = ok and float(cnt1[0]) == 16.0
    g = torch.Generator().manual_seed(5)
    color = torch.rand(3, 16, 3, generator=g)
    wsum = torch.rand(16, 1, generator=g)
    full_out = {"color_fine": color, "weight_sum": wsum, "gradient_error": torch.tensor(0.25)}
    full_loss = O.rnb_loss(full_out, batch["true_rgb"], batch["mask"])[0]
    m = (mine["mask"] > 0.5).float()
    share = ((color[:, lo:hi] - mine["true_rgb"]) * m[None]).abs().sum() / ((cnt[0] + 1e-5) * 3) \
        + 0.1 * 0.25 / world \
        + 0.1 * torch.nn.functional.binary_cross_entropy(wsum[lo:hi].clip(1e-3, 1 - 1e-3), m, reduction="sum") / cnt[1]
    tot = share.reshape(1).clone()
    P.allreduce_sum_(tot)
    ok = ok and torch.allclose(tot[0], full_loss, rtol=1e-5, atol=1e-6)
    w = torch.nn.Linear(4, 4)
    if rank == 1:
        with torch.no_grad():
            w.weight.add_(1.0)
    P.broadcast_parameters([w], src=0)
    gathered = [torch.empty_like(w.weight) for _ in range(world)]
    dist.all_gather(gathered, w.weight.data)
    ok = ok and torch.equal(gathered[0], gathered[1])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_sharding_and_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_rejects_ragged_batches():
    import rnb_neus_fork_amd.parallel as P
    with pytest.raises(ValueError):
        P.shard_range(10, 0, 4)
    assert P.shard_range(4096, 7, 8) == (3584, 4096)
