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


def _run(target, world, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_two_rank_sharding_and_allreduce():
    _run(_worker, 2)


def _worker_config4(rank, world, port, q):
    """BASELINE config 4's global batch (4096 rays) over `world` ranks, and the x-slabs of an SDF grid whose
    resolution does not divide by the world size (the last slab is shorter, at world 8 one more is shorter still)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rnb_neus_fork_amd.parallel as P
    torch.set_num_threads(1)
    n = 4096
    batch = O.synthetic_batch(n, seed=11, step=0, warmup=False)
    mine = P.shard_batch(batch, rank, world)
    lo, hi = P.shard_range(n, rank, world)
    per = n // world
    ok = (lo, hi) == (rank * per, (rank + 1) * per)
    for k, v in batch.items():
        if not torch.is_tensor(v):
            continue
        if v.dim() >= 1 and v.shape[0] == n:
            ok = ok and torch.equal(mine[k], v[lo:hi])
        elif v.dim() >= 2 and v.shape[1] == n:
            ok = ok and torch.equal(mine[k], v[:, lo:hi]) and mine[k].shape[0] == v.shape[0]
        else:
            ok = ok and mine[k] is v
    # the shards tile the batch: gathered back in rank order they are the batch
    got = [torch.empty_like(mine["rays_o"]) for _ in range(world)]
    dist.all_gather(got, mine["rays_o"])
    ok = ok and torch.equal(torch.cat(got), batch["rays_o"])
    cnt = P.global_mask_count(mine["mask"], True)
    ok = ok and float(cnt[0]) == float((batch["mask"] > 0.5).sum()) and float(cnt[1]) == float(n)
    # extract_fields' slabs: resolution 70 (the committed grid fixture's) and 61 (prime)
    for res in (70, 61, 5):
        per_x, x0, x1 = P.grid_slab(res, rank, world)
        ok = ok and per_x == -(-res // world) and 0 <= x0 <= x1 <= res and x1 - x0 <= per_x
        ix = torch.arange(res, dtype=torch.float32)
        vol = ix[:, None, None] * 10000 + ix[None, :, None] * 100 + ix[None, None, :]      # every voxel distinct
        slab = torch.zeros(per_x, res, res)
        slab[:x1 - x0] = vol[x0:x1]
        full = P.gather_grid_slabs(slab, res)
        ok = ok and full.shape == (res, res, res) and torch.equal(full, vol)
        spans = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(spans, torch.tensor([x0, x1]))
        edges = torch.stack(spans)
        ok = ok and int(edges[0, 0]) == 0 and int(edges[-1, 1]) == res and torch.equal(edges[1:, 0], edges[:-1, 1])
        if res == 70 and world == 8:
            ok = ok and [int(e[1] - e[0]) for e in edges] == [9, 9, 9, 9, 9, 9, 9, 7]
        if res == 5 and world == 8:
            ok = ok and [int(e[1] - e[0]) for e in edges] == [1, 1, 1, 1, 1, 0, 0, 0]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_config4_batch_and_ragged_grid_slabs(world):
    _run(_worker_config4, world)


def _worker_unequal(rank, world, port, q):
    """Exact large-batch semantics do not need equal shards: 24 rays split 16 + 8.  Every rank's loss share is its
    numerators over the GLOBAL denominators; the shares add up to the one-process loss of the whole batch.  (A mean
    over ranks of per-rank losses would weight the 8-ray shard's rays twice as much.)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rnb_neus_fork_amd.parallel as P
    torch.set_num_threads(1)
    n = 24
    lo, hi = ((0, 16), (16, 24))[rank]
    batch = O.synthetic_batch(n, seed=7, step=0, warmup=False)
    mask = batch["mask"][lo:hi]
    rgb = batch["true_rgb"][:, lo:hi]
    g = torch.Generator().manual_seed(9)
    color = torch.rand(3, n, 3, generator=g)
    wsum = torch.rand(n, 1, generator=g)
    # eikonal term: per-point error and relaxation weight (models/renderer.py:1079-1081) — a ratio of two sums
    # over all points of all rays, so the share is the local numerator over the global denominator
    perr = torch.rand(n, 12, generator=g)
    pw = (torch.rand(n, 12, generator=g) > 0.3).float()
    gerr = (perr * pw).sum() / (pw.sum() + 1e-5)
    full_out = {"color_fine": color, "weight_sum": wsum, "gradient_error": gerr}
    full_loss = O.rnb_loss(full_out, batch["true_rgb"], batch["mask"])[0]
    cnt = P.global_mask_count(mask, True)
    ok = float(cnt[0]) == float((batch["mask"] > 0.5).sum()) and float(cnt[1]) == float(n)
    den = pw[lo:hi].sum().reshape(1).clone()
    P.allreduce_sum_(den)
    m = (mask > 0.5).float()
    share = ((color[:, lo:hi] - rgb) * m[None]).abs().sum() / ((cnt[0] + 1e-5) * 3) \
        + 0.1 * (perr[lo:hi] * pw[lo:hi]).sum() / (den[0] + 1e-5) \
        + 0.1 * torch.nn.functional.binary_cross_entropy(wsum[lo:hi].clip(1e-3, 1 - 1e-3), m, reduction="sum") / cnt[1]
    tot = share.reshape(1).clone()
    P.allreduce_sum_(tot)
    ok = ok and torch.allclose(tot[0], full_loss, rtol=1e-5, atol=1e-6)
    # and the rank-mean of per-rank losses is NOT that loss (why exact mode sums shares instead)
    local_out = {"color_fine": color[:, lo:hi], "weight_sum": wsum[lo:hi],
                 "gradient_error": (perr[lo:hi] * pw[lo:hi]).sum() / (pw[lo:hi].sum() + 1e-5)}
    lm = O.rnb_loss(local_out, rgb, mask)[0].reshape(1).clone().float()
    P.allreduce_mean_(lm)
    ok = ok and abs(float(lm[0]) - float(full_loss)) > 1e-4
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_unequal_shards_sum_to_the_whole_batch_loss():
    _run(_worker_unequal, 2)


def test_shard_range_rejects_ragged_batches():
    import rnb_neus_fork_amd.parallel as P
    with pytest.raises(ValueError):
        P.shard_range(10, 0, 4)
    assert P.shard_range(4096, 7, 8) == (3584, 4096)
