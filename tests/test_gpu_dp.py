"""Data-parallel path on the device: two ranks render their shards of one ray batch with `set_data_parallel()`.
exact=True (default): every rank ends up with the gradient — and the loss value — of the SINGLE-PROCESS step on the
whole batch (the batch-global normalisers of SURVEY 8e travel in ONE 4-float all-reduce inside `rnb_loss(group=)`:
the eikonal ratio's two sums, mask_sum, the ray count).  exact=False: the DDP convention, mean of the ranks' local
gradients.  Transport: on the one-GPU test box the two ranks share device 0 over gloo; with >= 2 GPUs visible (the
driver's 8-GPU node) the same workers run one rank per GPU over the `nccl` backend = RCCL
(`test_exact_data_parallel_over_rccl`, skipped on one-GPU boxes)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port, backend="gloo"):
    """gloo: both ranks on device 0 (one-GPU box).  nccl: one rank per GPU over RCCL / xGMI."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dev


def _worker(rank, world, port, q):
    dev = _init(rank, world, port)
    import rnb_neus_fork_amd as R
    from rnb_neus_fork_amd import parallel as P
    from oracle import rnb_oracle as O
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    torch.manual_seed(0)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, dev)
    P.broadcast_parameters([sdf, devn, col])
    batch = O.synthetic_batch(16, seed=9, step=2, warmup=False)
    mine = {k: v.to(dev) for k, v in P.shard_batch(batch, rank, world).items()}
    leaves = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())

    def grads(dp):
        ren.set_data_parallel(enabled=dp, exact=False)
        for x in leaves:
            x.grad = None
        out = ren.render_rnb(mine["rays_o"], mine["rays_d"], mine["near"], mine["far"], mine["lights_dir"],
                             cos_anneal_ratio=1.0, t_rand=mine["t_rand"])
        O.rnb_loss(out, mine["true_rgb"], mine["mask"])[0].backward()
        torch.cuda.synchronize()
        return torch.cat([x.grad.reshape(-1) for x in leaves]).clone()

    g_local = grads(False)
    g_dp = grads(True)
    gathered = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(gathered, g_local)
    expect = sum(gathered) / world
    rel = float((g_dp - expect).norm() / expect.norm())
    differs = float((gathered[0] - gathered[1]).norm() / expect.norm())
    q.put((rank, rel, differs))
    dist.destroy_process_group()


def test_two_rank_gradients_are_the_mean_of_local_gradients():
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, rel, differs in res:
        assert rel < 1e-5, f"rank {rank}: DP gradient differs from the mean of local gradients by {rel:.2e}"
        assert differs > 1e-3, "the two shards should produce different local gradients"


def _exact_worker(rank, world, port, q, backend="gloo"):
    dev = _init(rank, world, port, backend)
    import rnb_neus_fork_amd as R
    from rnb_neus_fork_amd import parallel as P
    from oracle import rnb_oracle as O
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    torch.manual_seed(0)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, dev)
    P.broadcast_parameters([sdf, devn, col])
    # a batch whose shards have DIFFERENT mask counts and eikonal counts (otherwise per-shard normalisers would do)
    batch = O.synthetic_batch(24, seed=9, step=2, warmup=False)
    batch["mask"][:9] = 1.0
    batch["mask"][9:] = (torch.arange(15) % 4 == 0).float()[:, None]
    full = {k: v.to(dev) for k, v in batch.items()}
    mine = {k: v.to(dev) for k, v in P.shard_batch(batch, rank, world).items()}
    leaves = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())

    def run(b, dp, loss_group):
        ren.set_data_parallel(enabled=dp, exact=True)
        ren.set_variant(deterministic=True)       # ordered reductions: no atomic noise in the comparison
        for x in leaves:
            x.grad = None
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"])
        loss, parts = R.rnb_loss(out, b["true_rgb"], b["mask"], group=loss_group)
        loss.backward()
        torch.cuda.synchronize()
        return (torch.cat([x.grad.reshape(-1) for x in leaves]).clone(), float(loss), float(parts["eikonal_loss"]),
                ren.last_z_vals.clone())

    g_single, l_single, ge_single, z_single = run(full, False, None)         # the whole batch in one process
    # between the two steps ONLY RANK 0 renders (forward-only, as a rank-0 validation would): forwards are not
    # collective, so this must neither hang nor pair with the other rank's next all-reduce
    ren.set_data_parallel(enabled=True, exact=True)
    if rank == 0:
        with torch.no_grad():
            ev = ren.render_rnb(mine["rays_o"], mine["rays_d"], mine["near"], mine["far"], mine["lights_dir"],
                                cos_anneal_ratio=1.0, t_rand=mine["t_rand"])
        assert not hasattr(ev["gradient_error"], "rnb_dp_token") and bool(torch.isfinite(ev["gradient_error"]))
    g_dp, l_dp, ge_dp, z_dp = run(mine, True, dist.group.WORLD)            # my shard, exact data parallel
    # the pairing is checked both ways: exact render + loss without group, and group= on a non-exact render
    pairing_ok = True
    out = ren.render_rnb(mine["rays_o"], mine["rays_d"], mine["near"], mine["far"], mine["lights_dir"],
                         cos_anneal_ratio=1.0, t_rand=mine["t_rand"])
    try:
        R.rnb_loss(out, mine["true_rgb"], mine["mask"])
        pairing_ok = False
    except RuntimeError:
        pass
    try:
        O.rnb_loss(out, mine["true_rgb"], mine["mask"])[0].backward()      # the reference's torch ops: unpaired
        pairing_ok = False
    except RuntimeError:
        pass
    ren.set_data_parallel(enabled=True, exact=False)
    out = ren.render_rnb(mine["rays_o"], mine["rays_d"], mine["near"], mine["far"], mine["lights_dir"],
                         cos_anneal_ratio=1.0, t_rand=mine["t_rand"])
    try:
        R.rnb_loss(out, mine["true_rgb"], mine["mask"], group=dist.group.WORLD)
        pairing_ok = False
    except RuntimeError:
        pass
    lo = rank * (24 // world)
    same_samples = bool(torch.equal(z_dp, z_single[lo:lo + 24 // world]))
    rel = float((g_dp - g_single).norm() / g_single.norm())
    # the DDP convention on the same shards differs visibly (unequal mask counts)
    ren.set_data_parallel(enabled=True, exact=False)
    for x in leaves:
        x.grad = None
    out = ren.render_rnb(mine["rays_o"], mine["rays_d"], mine["near"], mine["far"], mine["lights_dir"],
                         cos_anneal_ratio=1.0, t_rand=mine["t_rand"])
    R.rnb_loss(out, mine["true_rgb"], mine["mask"])[0].backward()
    g_ddp = torch.cat([x.grad.reshape(-1) for x in leaves])
    rel_ddp = float((g_ddp - g_single).norm() / g_single.norm())
    q.put((rank, rel, abs(l_dp - l_single), abs(ge_dp - ge_single), same_samples and pairing_ok, rel_ddp))
    dist.destroy_process_group()


def test_exact_data_parallel_step_equals_the_single_process_step():
    """"so that G-GPU results equal the 1-GPU result on the same rays" — gradient rel-L2 <= 1e-5."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exact_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, rel, dl, dge, same_samples, rel_ddp in res:
        assert same_samples, ("a shard must sample exactly the depths the whole batch samples for those rays, and the "
                              "loss / renderer pairing checks must raise")
        assert rel <= 1e-5, f"rank {rank}: exact-DP gradient differs from the whole-batch gradient by {rel:.2e}"
        assert dl <= 2e-6 and dge <= 1e-7, f"rank {rank}: loss / gradient_error differ ({dl:.2e}, {dge:.2e})"
        assert rel_ddp > 1e-3, "with unequal mask counts the DDP mean must differ from the whole-batch gradient"
    print(f"exact DP vs single process: gradient rel-L2 {max(r[1] for r in res):.2e}; DDP-mean convention: {res[0][5]:.2e}")


def _run_exact(backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exact_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


def test_exact_data_parallel_over_rccl():
    """The same exact-DP check with one rank per GPU over the `nccl` backend (RCCL over xGMI): the first RCCL call this
    code makes is here, in a test, not in the scaling bench.  Needs two GPUs: skipped on the one-GPU boxes, runs on the
    driver's 8-GPU node.  `device_count()` does not initialise the GPU in this (parent) process."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (one rank per GPU over RCCL)")
    for rank, rel, dl, dge, ok, rel_ddp in _run_exact("nccl"):
        assert ok
        assert rel <= 1e-5, f"rank {rank}: exact-DP gradient over RCCL differs from the whole-batch gradient by {rel:.2e}"
        assert dl <= 2e-6 and dge <= 1e-7
        assert rel_ddp > 1e-3


def _rccl_one_rank_worker(port, q):
    """world_size 1 over the `nccl` backend: RCCL is loaded, a communicator is created on the device and the two
    collectives of a training step (the 4-float normaliser all-reduce, the flat-gradient SUM) run through it."""
    dev = _init(0, 1, port, "nccl")
    import rnb_neus_fork_amd as R
    from oracle import rnb_oracle as O
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    torch.manual_seed(0)
    sdf, devn, col, ren = R.build_from_named_params(mc, O.init_params(mc), dev)
    b = {k: v.to(dev) for k, v in O.synthetic_batch(16, seed=9, step=2).items()}
    leaves = list(sdf.parameters()) + list(devn.parameters()) + list(col.parameters())
    res = []
    for dp in (False, True):
        ren.set_data_parallel(enabled=dp, exact=True)
        ren.set_variant(deterministic=True)
        for x in leaves:
            x.grad = None
        out = ren.render_rnb(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                             t_rand=b["t_rand"])
        loss, _ = R.rnb_loss(out, b["true_rgb"], b["mask"], group=dist.group.WORLD if dp else None)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), torch.cat([x.grad.reshape(-1) for x in leaves]).clone()))
    same = bool(torch.equal(res[0][1], res[1][1])) and abs(res[0][0] - res[1][0]) < 1e-7
    q.put((same, dist.get_backend()))
    dist.destroy_process_group()


def test_rccl_single_rank_smoke():
    """The one-GPU boxes cannot run two RCCL ranks, but they can run ONE: the exact data-parallel step over a
    world-size-1 `nccl` group must equal the plain single-process step bit for bit (the collectives are identities),
    which proves RCCL loads, builds a communicator and reduces on this software stack before the scaling bench needs it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    same, backend = q.get(timeout=300)
    p.join(timeout=60)
    assert backend == "nccl" and same


def _grid_worker(rank, world, port, q):
    dev = _init(rank, world, port)
    import rnb_neus_fork_amd as R
    from oracle import rnb_oracle as O
    mc = O.ModelConf(sdf=O.SDFConf(d_out=65, d_hidden=64), color=O.ColorConf(d_feature=64, d_hidden=64),
                     render=O.RenderConf(n_samples=16, n_importance=16))
    torch.manual_seed(0)
    p = O.init_params(mc)
    sdf, devn, col, ren = R.build_from_named_params(mc, p, dev)
    lo, hi = torch.tensor([-1.0, -1.0, -1.0]), torch.tensor([1.0, 1.0, 1.0])
    single = ren.extract_fields(lo, hi, 24, chunk=8)               # 3 x-slabs, no group
    ren.set_data_parallel()
    sharded = ren.extract_fields(lo, hi, 24, chunk=8)              # slabs dealt to the 2 ranks + all-reduce
    q.put((rank, float(abs(single - sharded).max()), float(abs(single).max())))
    dist.destroy_process_group()


def test_extract_fields_sharded_over_ranks_equals_single_rank():
    """the SDF grid shards by x-slab; every rank ends up with the full volume."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, diff, mag in res:
        assert diff == 0.0 and mag > 0.1, f"rank {rank}: sharded grid differs by {diff}"
