"""bench.py — training rays/s of the RNb-NeuS renderer hot path on MI355X.

One "step" = one train_rnb iteration (exp_runner.py:174-263) on one synthetic ray batch already resident
in HBM: weight-norm materialisation, hierarchical sampling (64 coarse + 4x16 importance samples), fine
pass forward (SDF net, analytic normal, albedo net, composite), the R9 loss, the explicit backward,
[RCCL all-reduce of the flat gradient buffer when N > 1] and Adam.  Workload = BASELINE.json configs[1]:
wmask_rnb.conf, 512 rays x (64+64) samples per GPU, 3 lights, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "hbm_traffic.json")


def measured_traffic(n_gpus, rays):
    """HBM bytes per MFMA-family launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    see profiles/README.md).  PMC collection needs its own rocprofv3 runs, so bench.py reports the stored
    measurement of the same workload (N=1, 512 rays) and null for any other shape."""
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        if n_gpus == 1 and rays == t.get("rays", 512):
            return round(t["hbm_bytes_per_launch"]), {"hbm_bytes_per_step": round(t["hbm_bytes_per_step"]),
                                                       "launches_per_step": t["launches_per_step"],
                                                       "source": "profiles/hbm_traffic.json"}
    except Exception:
        pass
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=512, help="rays per GPU per step")
    ap.add_argument("--warmup-mode", action="store_true", help="render_rnb_warmup instead of render_rnb")
    ap.add_argument("--no-albedo", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--torch-train-ops", action="store_true",
                    help="loss as the reference's chain of torch ops and torch.optim.Adam(fused=True) instead of "
                         "the library's one-launch loss and flat Adam")
    return ap.parse_args()


def torch_rnb_loss(render_out, true_rgb, mask, igr_weight=0.1, mask_weight=0.1):
    """The loss exactly as exp_runner.py:229-258 writes it (torch ops on the renderer's outputs)."""
    import torch.nn.functional as F
    n_lights = true_rgb.shape[0]
    mask = (mask > 0.5).float() if mask_weight > 0.0 else torch.ones_like(mask)
    mask_sum = mask.sum() + 1e-5
    err = ((render_out["color_fine"] - true_rgb) * mask[None, :, :]).reshape(-1, true_rgb.shape[-1])
    color_loss = F.l1_loss(err, torch.zeros_like(err), reduction="sum") / (mask_sum * n_lights)
    eik = render_out["gradient_error"]
    mask_loss = F.binary_cross_entropy(render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    return color_loss + eik * igr_weight + mask_loss * mask_weight, {}


def make_oracle_conf():
    from oracle import rnb_oracle as O
    return O, O.ModelConf()


def cpu_baseline(rays, steps, warmup_mode, no_albedo):
    """The oracle (a PyTorch-CPU restatement with the reference's op structure: two fine SDF forwards,
    autograd double backward, Adam) timed on this box's host cores on a bounded sample."""
    O, mc = make_oracle_conf()
    # the GPU box grants 16 host cores per GPU; os.cpu_count() reports the whole host
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("RNB_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    p = O.init_params(mc)
    for v in p.values():
        v.requires_grad_(True)
    names = O.param_order(mc, no_albedo)
    opt = torch.optim.Adam([p[k] for k in names], lr=5e-4)
    times = []
    for it in range(steps + 1):
        b = O.synthetic_batch(rays, seed=0, step=it, warmup=warmup_mode)
        t0 = time.perf_counter()
        out = O.render_rnb(p, mc, b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"],
                           t_rand=b["t_rand"], cos_anneal_ratio=1.0, no_albedo=no_albedo, warmup=warmup_mode)
        loss, _ = O.rnb_loss(out, b["true_rgb"], b["mask"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        print(f"[bench] cpu baseline step {it}: {dt:.2f} s", file=sys.stderr, flush=True)
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": rays / med, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of {rays} rays x (64+64) samples, full train step "
                      f"(render_rnb + loss + backward + Adam) with oracle/rnb_oracle.py, torch {torch.__version__} "
                      f"CPU fp32, {threads} threads; median {med:.3f} s/step"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the renderer has no CPU path)"
    # one process per GPU; RNB_SHARE_GPU=1 (functional rehearsal of the N>1 path on a one-GPU box, with
    # RNB_DIST_BACKEND=gloo) puts every rank on device 0
    if os.environ.get("RNB_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("RNB_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: using {world}", file=sys.stderr)

    # the measured leg uses the product only (package + librnbneus_hip.so); oracle/ is touched by
    # cpu_baseline() alone
    import rnb_neus_fork_amd as R
    from rnb_neus_fork_amd import parallel as P
    from rnb_neus_fork_amd.synthetic import synthetic_batch
    lib = R.native.load()

    # confs/wmask_rnb.conf:53-90, constructed in the order of exp_runner.py:95-100 under seed 0
    torch.manual_seed(0)
    sdf = R.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                       geometric_init=True, weight_norm=True).to(dev)
    devnet = R.SingleVarianceNetwork(init_val=0.3).to(dev)
    col = R.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                             weight_norm=True, multires_view=4, squeeze_out=True).to(dev)
    ren = R.NeuSRenderer(None, sdf, devnet, col, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4,
                         perturb=1.0)
    if world > 1:
        P.broadcast_parameters([sdf, devnet, col])
        ren.set_data_parallel()
    params = list(sdf.parameters()) + list(devnet.parameters())
    if not args.no_albedo:
        params += list(col.parameters())
    # exp_runner.py:115: Adam, lr 5e-4.  Default: the library's flat single-launch Adam (identical update rule,
    # tests/test_gpu_parity.py); --torch-train-ops keeps torch.optim.Adam and the reference's chain of torch ops
    # for the loss, i.e. exactly what exp_runner.py would run around the drop-in renderer.
    if args.torch_train_ops:
        opt = torch.optim.Adam(params, lr=5e-4, fused=True)
    else:
        opt = R.FlatAdam(params, lr=5e-4)

    B = args.rays
    n_batches = 8
    # inputs resident in HBM before the timed region: each rank owns its contiguous shard of a global batch
    batches = []
    for i in range(n_batches):
        gb = synthetic_batch(B * world, seed=0, step=i, warmup=args.warmup_mode)
        mine = P.shard_batch(gb, rank, world, n_rays=B * world)
        batches.append({k: v.to(dev) for k, v in mine.items()})

    loss_fn = torch_rnb_loss if args.torch_train_ops else R.rnb_loss

    def step(i):
        b = batches[i % n_batches]
        fn = ren.render_rnb_warmup if args.warmup_mode else ren.render_rnb
        out = fn(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                 no_albedo=args.no_albedo, t_rand=b["t_rand"])
        loss, _ = loss_fn(out, b["true_rgb"], b["mask"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] warm-up done ({args.warmup} steps)", file=sys.stderr, flush=True)
    use_events = not args.no_gemm_events
    barrier()
    if use_events:
        lib.rnb_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    gemm_ms, gemm_n, gemm_fl = C.c_double(0), C.c_int64(0), C.c_double(0)
    if use_events:
        R.native.check(lib.rnb_profile_collect(C.byref(gemm_ms), C.byref(gemm_n), C.byref(gemm_fl)))
        lib.rnb_profile_enable(0)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.detach())
    if rank == 0:
        print(f"[bench] timed region done: {elapsed:.3f} s for {args.steps} steps", file=sys.stderr, flush=True)

    if rank == 0:
        flags = R.native.MODE_MVPS | (R.native.FLAG_NO_ALBEDO if args.no_albedo else 0)
        tf, ff = C.c_double(), C.c_double()
        R.native.check(lib.rnb_algorithmic_flops(C.byref(ren.desc), B, flags, C.byref(tf), C.byref(ff)))
        ms_per_step = 1e3 * elapsed / args.steps
        value = B * world * args.steps / elapsed
        roof = None
        traffic, traffic_detail = measured_traffic(world, B)
        if use_events and gemm_n.value > 0:
            ach = gemm_fl.value / (gemm_ms.value * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                    "traffic_detail": traffic_detail,
                    "kernel": "fp32-MFMA family: fused_forward/reverse/ra/fb_kernel, gemm_dw_direct_kernel, "
                              "gemm_rows_kernel<*>",
                    "launches_per_step": gemm_n.value / args.steps,
                    "avg_launch_us": round(1e3 * gemm_ms.value / gemm_n.value, 2),
                    "flop_per_launch": round(gemm_fl.value / gemm_n.value, 1),
                    "gemm_ms_per_step": round(gemm_ms.value / args.steps, 3),
                    "step_algorithmic_tflops": round(tf.value / (ms_per_step * 1e-3) / 1e12, 3),
                    "step_frac": round(tf.value / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(B, args.cpu_steps, args.warmup_mode, args.no_albedo)
        line = {
            "metric": "training rays/sec at 512 rays x 128 samples/ray",
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DiLiGenT-MV-shaped synthetic rays, wmask_rnb.conf networks (8x256 SDF MLP + "
                                   "2x256 albedo MLP), train_rnb step "
                                   f"({'render_rnb_warmup' if args.warmup_mode else 'render_rnb'}), "
                                   f"{B} rays x (64+64) samples per GPU, 3 lights, geometric init, Adam",
                       "rays_per_gpu": B, "samples_per_ray": 128, "n_lights": 3,
                       "no_albedo": bool(args.no_albedo), "parallelism": f"dp{world}", "final_loss": final_loss,
                       "train_ops": ("torch ops loss + torch.optim.Adam(fused)" if args.torch_train_ops
                                     else "rnb_loss_rnb + rnb_adam_step (one launch each)")},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
