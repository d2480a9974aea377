"""bench.py — training rays/s of the RNb-NeuS renderer hot path on MI355X.

One "step" = one train_rnb iteration (exp_runner.py:174-263) on one synthetic ray batch already resident
in HBM: weight-norm materialisation, hierarchical sampling (64 coarse + 4x16 importance samples), fine
pass forward (SDF net, analytic normal, albedo net, composite), the R9 loss, the explicit backward,
[RCCL all-reduce of the flat gradient buffer when N > 1] and Adam.  Default workload = BASELINE.json
configs[1]: wmask_rnb.conf, 512 rays x (64+64) samples per GPU, 3 lights, fp32.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            # WORLD_SIZE unset: this process SPAWNS N ranks (one per GPU) itself,
                                            # before it touches a GPU, and exits with their status
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # the driver's form: ranks come from the env
    python bench.py --gpus 8 --scaling strong --global-rays 4096       # BASELINE config 4 (strong scaling)
    python bench.py --dtype bf16 --samples 256                         # BASELINE config 5's arithmetic (bf16 sweeps)
    python bench.py --mode mesh --resolution 512                       # validate_mesh's SDF grid (forward only)

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0           # same guide, "HBM3E peak BW 8.0 TB/s spec"
PROFILES = os.path.join(ROOT, "profiles")


def measured_traffic(name, n_gpus, rays, samples, build_id):
    """HBM bytes per MFMA-family launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    see profiles/README.md).  PMC collection needs its own rocprofv3 runs, so bench.py reports the stored
    measurement — only for the same workload AND the same build of the library: the profile carries the `build_id`
    (rnb_build_id(): hash of csrc/ + flags) of the library it was collected with, and bytes measured on other kernels are
    not quoted.  Returns (profile or None, note or None)."""
    try:
        with open(os.path.join(PROFILES, name)) as f:
            t = json.load(f)
    except Exception:
        return None, f"profiles/{name}: not found"
    if not (n_gpus == 1 and rays == t.get("rays", 512) and samples == t.get("samples", 128)):
        return None, f"profiles/{name} holds another workload"
    if t.get("build_id") != build_id:
        return None, (f"profiles/{name} was collected on build {t.get('build_id', '(none recorded)')}, this library is "
                      f"{build_id}: stored bytes not quoted")
    return t, None


def measured_mfma_busy(build_id, name="sq_counters.json"):
    """Matrix-pipe busy fraction per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)) from the stored
    rocprofv3 PMC pass (tools/sq_summary.py writes profiles/sq_counters.json with the library's build id); quoted only for
    the build it was collected on.  Returns {kernel name: {...}} or None."""
    try:
        with open(os.path.join(PROFILES, name)) as f:
            t = json.load(f)
    except Exception:
        return None
    if t.get("build_id") != build_id:
        return None
    return t.get("per_kernel")


# kernel class tag (rnb_profile_report) -> substring of the kernel names whose PMC traffic belongs to it
CLASS_KERNELS = {"RA_sweep": "fused_ra_kernel", "FB_sweep": "fused_fb_", "R_sweep": "fused_reverse_kernel",
                 "F_sweep(save)": "fused_forward_kernel<2, true", "dW(x3: 256x256 + narrow jobs)": "gemm_dw_x3_kernel",
                 "dW(all)": "bf_dw_kernel", "dW(other)": "gemm_dw_direct_kernel", "layer_gemm": ("EpiReluMask", "EpiStore"),
                 "layer_gemm(forward)": "EpiRelu,", "albedo_fwd": "color_fwd_h2_kernel", "albedo_bwd": "color_bwd_h2_kernel"}
HBM_BOUND_TBS = 4.0    # a class that moves more than this (PMC bytes / event time) is labelled hbm-bound: half of the 8 TB/s
                       # spec, ~2/3 of what a plain copy reaches (6.3 TB/s)


# matrix terms per fp32 product of every kernel class under the default arithmetic (RNB_VARIANT_X3 + X2H): three fp16 terms
# except the RA sweep, which keeps the six bf16 terms (state-traffic bound: DESIGN 4)
X2H_TERMS = {"F_sweep(save)": 3, "F_sweep(forward_only)": 3, "R_sweep": 3, "FB_sweep": 3,
             "dW(x3: 256x256 + narrow jobs)": 3, "layer_gemm(forward)": 6, "RA_sweep": 6, "layer_gemm": 6, "albedo_fwd": 3,
             "albedo_bwd": 3}


def class_peak(tag, default_peak, terms):
    """MFMA ceiling of one kernel class in algorithmic fp32 TFLOP/s: the dense 16-bit peak over its terms per product."""
    return BF16_MFMA_PEAK_TFLOPS / terms[tag] if terms and tag in terms else default_peak


def kernel_classes(lib, steps, peak_tflops, traffic=None, terms=None):
    """Per kernel class of the last rnb_profile_collect (HIP events on the launch stream): ms per step, launches per step,
    algorithmic TFLOP/s and its fraction of the class's MFMA ceiling (`peak_tflops`, or 2500 / terms[class]); with the
    stored PMC traffic of the same workload also the HBM rate of the class and the roof that bounds it (`bound`: "hbm"
    above HBM_BOUND_TBS, else "mfma"); sorted by time."""
    need = lib.rnb_profile_report(None, 0)
    buf = C.create_string_buffer(int(need) + 16)
    lib.rnb_profile_report(buf, len(buf))
    out = []
    for ln in buf.value.decode().splitlines():
        tag, ms, n, fl = ln.rsplit(" ", 3)
        ms, n, fl = float(ms), int(n), float(fl)
        if ms <= 0:
            continue
        tf = fl / (ms * 1e-3) / 1e12
        pk = class_peak(tag, peak_tflops, terms)
        d = {"kernel_class": tag, "ms_per_step": round(ms / steps, 4), "launches_per_step": n / steps,
             "tflops": round(tf, 2), "frac": round(tf / pk, 4), "flop_per_step": fl / steps}
        if terms:
            d["terms"] = terms.get(tag)
            d["peak"] = round(pk, 1)
        key = CLASS_KERNELS.get(tag)
        if traffic and key:
            b = sum((2.0 * v["fetch_size_raw_kb_per_launch"] + v["write_size_kb_per_launch"]) * 1024.0 * v["launches_per_step"]
                    for k, v in traffic.get("per_kernel", {}).items()
                    if any(kk in k for kk in (key if isinstance(key, tuple) else (key,))))
            if b > 0:
                tbs = b / (ms / steps * 1e-3) / 1e12
                d.update(hbm_gb_per_step=round(b / 1e9, 3), hbm_tb_per_s=round(tbs, 2),
                         hbm_frac=round(tbs * 1e3 / HBM_PEAK_GBS, 3), bound="hbm" if tbs > HBM_BOUND_TBS else "mfma")
        out.append(d)
    out.sort(key=lambda d: -d["ms_per_step"])
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 10)")
    ap.add_argument("--rays", type=int, default=512, help="rays per GPU per step (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-rays", type=int, default=4096, help="rays per step of the whole job (strong scaling)")
    ap.add_argument("--samples", type=int, default=128, help="samples per ray (half coarse, half importance)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="arithmetic of the SDF-network sweeps (bf16: bf16 operands, fp32 accumulate; config 5)")
    ap.add_argument("--mode", choices=("train", "render", "mesh"), default="train",
                    help="train: full train_rnb step (the metric); render: forward-only NeuSRenderer.render under no_grad "
                         "(SURVEY 8d 'also'); mesh: validate_mesh's SDF grid + marching cubes")
    ap.add_argument("--device-rays", action="store_true",
                    help="--mode train: draw pixels and gather rays / targets / lights INSIDE the timed step with "
                         "DeviceRays.sample() on HBM-resident synthetic image stacks (SURVEY 8f-2) instead of pre-staged "
                         "batches; the line also carries the pre-staged time of the same run")
    ap.add_argument("--stack", default="20x512x612", help="--device-rays: views x H x W of the synthetic capture "
                                                          "(config 5: 200x1024x1024, ~22 GB of HBM)")
    ap.add_argument("--resolution", type=int, default=512, help="--mode mesh: grid points per axis")
    ap.add_argument("--no-marching-cubes", action="store_true", help="--mode mesh: time the SDF grid only")
    ap.add_argument("--warmup-mode", action="store_true", help="render_rnb_warmup instead of render_rnb")
    ap.add_argument("--no-albedo", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed steps of the CPU baseline (median reported; BASELINE.md 3)")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="untimed warm-up steps of the CPU baseline")
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--deterministic", action="store_true", help="ordered reductions instead of fp32 atomics")
    ap.add_argument("--x3", action="store_true", help="request RNB_VARIANT_X3 explicitly (it is the default for the 256-wide shape)")
    ap.add_argument("--f32-mfma", action="store_true", help="A/B knob: native fp32 MFMA sweeps (RNB_VARIANT_F32_MFMA) instead of x3")
    ap.add_argument("--dw-staged", action="store_true", help="A/B knob: LDS-DMA staged dW kernel for the 256x256 jobs")
    ap.add_argument("--fwd-ti", type=int, default=0, help="A/B knob: tile rows per wave of the forward sweep (1 | 2)")
    ap.add_argument("--bwd-ti", type=int, default=0, help="A/B knob: tile rows per wave of the backward sweeps (1 | 2)")
    ap.add_argument("--fwd-nw", type=int, default=0, help="A/B knob: waves per workgroup of the forward sweep (4 | 8)")
    ap.add_argument("--bwd-nw", type=int, default=0, help="A/B knob: waves per workgroup of the backward sweeps (4 | 8)")
    ap.add_argument("--reg-tile", action="store_true", help="A/B knob: M/V kernels for the large forward-only sweeps (RNB_VARIANT_REG_TILE)")
    ap.add_argument("--lds-tile", action="store_true", help="A/B knob: force the LDS-tile sweep kernels (RNB_VARIANT_LDS_TILE)")
    ap.add_argument("--no-x2h", dest="x2h", action="store_false", default=None,
                    help="A/B knob: forward-type sweeps on six bf16 terms like the backward ones (RNB_VARIANT_NO_X2H)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short extra legs (no-albedo, bf16 x 256 samples, render, fp32 MFMA) that the default "
                         "--gpus 1 train run appends under `also`")
    ap.add_argument("--also-steps", type=int, default=20, help="timed steps of each `also` leg")
    ap.add_argument("--torch-train-ops", action="store_true",
                    help="loss as the reference's chain of torch ops and torch.optim.Adam(fused=True) instead of "
                         "the library's one-launch loss and flat Adam")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 3 if a.mode == "mesh" else 50
    if a.warmup is None:
        a.warmup = 1 if a.mode == "mesh" else 10
    return a


# -------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N without a torchrun environment
# -------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """Starts one child process per rank and waits for them.  This launcher process never imports torch and makes no
    GPU-adjacent call at all; each child decides by itself (init_distributed) whether the box has a device per rank
    or the ranks must share device 0 over gloo (a one-GPU development box: a functional rehearsal of the N > 1 path,
    flagged `"rehearsal": true` in the JSON line — never a scaling measurement)."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc != 0:
        print(f"[bench] a rank failed (exit {rc}): fewer than {n} ranks completed", file=sys.stderr)
    return rc


def torch_rnb_loss(render_out, true_rgb, mask, igr_weight=0.1, mask_weight=0.1):
    """The loss exactly as exp_runner.py:229-258 writes it (torch ops on the renderer's outputs)."""
    import torch
    import torch.nn.functional as F
    n_lights = true_rgb.shape[0]
    mask = (mask > 0.5).float() if mask_weight > 0.0 else torch.ones_like(mask)
    mask_sum = mask.sum() + 1e-5
    err = ((render_out["color_fine"] - true_rgb) * mask[None, :, :]).reshape(-1, true_rgb.shape[-1])
    color_loss = F.l1_loss(err, torch.zeros_like(err), reduction="sum") / (mask_sum * n_lights)
    eik = render_out["gradient_error"]
    mask_loss = F.binary_cross_entropy(render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
    return color_loss + eik * igr_weight + mask_loss * mask_weight, {}


def cpu_threads():
    # the GPU box grants 16 host cores per GPU; os.cpu_count() reports the whole host
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return int(os.environ.get("RNB_CPU_THREADS", min(avail, 16)))


def cpu_baseline(rays, samples, steps, warmup_mode, no_albedo, warm=3):
    """The oracle (a PyTorch-CPU restatement with the reference's op structure: two fine SDF forwards,
    autograd double backward, Adam) timed on this box's host cores on a bounded sample.  The oracle computes in
    fp32 whatever --dtype says: the reference has no reduced-precision path."""
    import torch
    from oracle import rnb_oracle as O
    mc = O.ModelConf(render=O.RenderConf(n_samples=samples // 2, n_importance=samples // 2))
    threads = cpu_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    p = O.init_params(mc)
    for v in p.values():
        v.requires_grad_(True)
    names = O.param_order(mc, no_albedo)
    opt = torch.optim.Adam([p[k] for k in names], lr=5e-4)
    times = []
    for it in range(steps + warm):
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
        if it >= warm:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": rays / med, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"median of {steps} timed steps (after {warm} warm-up steps) of {rays} rays x ({samples // 2}+{samples // 2}) samples, "
                      f"full train step (render_rnb + loss + backward + Adam) with oracle/rnb_oracle.py, torch "
                      f"{torch.__version__} CPU fp32, {threads} threads; median {med:.3f} s/step"}


def cpu_baseline_render(rays, samples, steps, warm=3):
    """CPU leg of --mode render: the oracle's `render` (sampling + fine pass with the autograd normal, as the reference
    computes it even for a forward-only image, exp_runner.py:389-472)."""
    import torch
    from oracle import rnb_oracle as O
    mc = O.ModelConf(render=O.RenderConf(n_samples=samples // 2, n_importance=samples // 2))
    threads = cpu_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    p = O.init_params(mc)
    times = []
    for it in range(steps + warm):
        b = O.synthetic_batch(rays, seed=0, step=it)
        t0 = time.perf_counter()
        O.render(p, mc, b["rays_o"], b["rays_d"], b["near"], b["far"], t_rand=b["t_rand"], cos_anneal_ratio=1.0)
        dt = time.perf_counter() - t0
        print(f"[bench] cpu baseline render {it}: {dt:.2f} s", file=sys.stderr, flush=True)
        if it >= warm:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": rays / med, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"median of {steps} timed calls (after {warm} warm-up calls) of oracle/rnb_oracle.py::render on {rays} rays x ({samples // 2}+"
                      f"{samples // 2}) samples, torch {torch.__version__} CPU fp32, {threads} threads; median {med:.3f} s"}


def cpu_baseline_mesh(n_points):
    """CPU leg of --mode mesh: the oracle's SDF forward (models/fields.py:82-108) on a bounded sample of grid points."""
    import torch
    from oracle import rnb_oracle as O
    mc = O.ModelConf()
    threads = cpu_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    p = O.init_params(mc)
    pts = torch.rand(n_points, 3) * 2 - 1
    with torch.no_grad():
        O.sdf_only(p, mc.sdf, pts[:4096])
        t0 = time.perf_counter()
        O.sdf_only(p, mc.sdf, pts)
        dt = time.perf_counter() - t0
    return {"value": n_points / dt, "unit": "points/s", "cores": threads, "kind": "port",
            "sample": f"one no-grad SDF forward of {n_points} points with oracle/rnb_oracle.py::sdf_only, torch "
                      f"{torch.__version__} CPU fp32, {threads} threads: {dt:.2f} s"}


def init_distributed(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the renderer has no CPU path)"
    # one process per GPU; RNB_SHARE_GPU=1 (functional rehearsal of the N>1 path on a one-GPU box, with
    # RNB_DIST_BACKEND=gloo) puts every rank on device 0
    # (also chosen by every rank on its own when the box has fewer devices than ranks; device_count() does not
    # initialise the GPU, and every rank of one box sees the same count, so the ranks agree)
    rehearsal = bool(os.environ.get("RNB_SHARE_GPU")) or torch.cuda.device_count() < world
    if rehearsal:
        local_rank = 0
        os.environ.setdefault("RNB_DIST_BACKEND", "gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        backend = os.environ.get("RNB_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != world:
            raise RuntimeError(f"process group has {dist.get_world_size()} ranks, expected {world}")
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: using {world}", file=sys.stderr)
    # a box with a device per rank must give an RCCL measurement: anything else (RNB_SHARE_GPU / RNB_DIST_BACKEND forced by
    # the environment) would print a rehearsal line where a scaling point is expected — refuse instead
    if world > 1 and torch.cuda.device_count() >= world and (rehearsal or backend != "nccl") and not os.environ.get("RNB_ALLOW_REHEARSAL"):
        raise SystemExit(f"[bench] {torch.cuda.device_count()} devices for {world} ranks, but the ranks would not talk RCCL "
                         f"(backend {backend}, shared device {rehearsal}): not a scaling measurement "
                         "(set RNB_ALLOW_REHEARSAL=1 to run the functional rehearsal anyway)")
    return world, rank, dev, backend, rehearsal


def synthetic_capture(R, dev, n_views, H, W, n_lights=3):
    """HBM-resident stacks shaped like the reference's Dataset tensors (models/dataset.py:219-239): images and
    per-pixel light directions [V, L, H, W, 3], masks [V, H, W, 1], inverse intrinsics and poses [V, 4, 4].  Cameras
    sit on the radius-3 sphere looking at the origin with a field of view that just holds the unit sphere; the mask
    is the silhouette of the radius-0.5 ball (the geometric init).  Generated on the device, view by view."""
    import math
    import torch
    g = torch.Generator(device=dev).manual_seed(0)
    gv = torch.Generator("cpu").manual_seed(0)
    c = torch.randn(n_views, 3, generator=gv)
    c = 3.0 * c / c.norm(dim=-1, keepdim=True)
    fwd = -c / c.norm(dim=-1, keepdim=True)
    up0 = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd).clone()
    up0[fwd[:, 2].abs() > 0.9] = torch.tensor([1.0, 0.0, 0.0])
    right = torch.linalg.cross(up0, fwd)
    right = right / right.norm(dim=-1, keepdim=True)
    up = torch.linalg.cross(fwd, right)
    pose = torch.eye(4).repeat(n_views, 1, 1)
    pose[:, :3, 0], pose[:, :3, 1], pose[:, :3, 2], pose[:, :3, 3] = right, up, fwd, c
    focal = 0.5 * min(H, W) / math.tan(math.asin(1.0 / 3.0))
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    Kinv = torch.inverse(K).repeat(n_views, 1, 1)
    images = torch.empty(n_views, n_lights, H, W, 3, device=dev)
    lights = torch.empty(n_views, n_lights, H, W, 3, device=dev)
    masks = torch.empty(n_views, H, W, 1, device=dev)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32),
                            torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], -1)                          # [H, W, 3]
    for v in range(n_views):
        images[v].uniform_(0.0, 1.0, generator=g)
        lights[v].normal_(generator=g)
        lights[v] /= lights[v].norm(dim=-1, keepdim=True)
        d = pix @ Kinv[v, :3, :3].T.to(dev)
        d = d / d.norm(dim=-1, keepdim=True)
        d = d @ pose[v, :3, :3].T.to(dev)
        o = c[v].to(dev)
        closest = o + d * (-(o * d).sum(-1, keepdim=True))
        masks[v] = (closest.norm(dim=-1, keepdim=True) < 0.5).float()
    lw = torch.randn(n_views, n_lights, 3, generator=gv)
    lw = lw / lw.norm(dim=-1, keepdim=True)
    return R.DeviceRays(images, None, masks, lights, lw, Kinv, pose, dev)


def build_model(R, dev, samples, dtype, deterministic, fwd_ti=0, bwd_ti=0, dw_staged=False, x3=False, fwd_nw=0, bwd_nw=0, f32_mfma=False,
                reg_tile=False, lds_tile=False, x2h=None):
    import torch
    # confs/wmask_rnb.conf:53-90, constructed in the order of exp_runner.py:95-100 under seed 0
    torch.manual_seed(0)
    sdf = R.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                       geometric_init=True, weight_norm=True).to(dev)
    devnet = R.SingleVarianceNetwork(init_val=0.3).to(dev)
    col = R.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                             weight_norm=True, multires_view=4, squeeze_out=True).to(dev)
    ren = R.NeuSRenderer(None, sdf, devnet, col, n_samples=samples // 2, n_importance=samples // 2, n_outside=0,
                         up_sample_steps=4, perturb=1.0)
    ren.set_variant(bf16=(dtype == "bf16"), deterministic=deterministic, fwd_ti=fwd_ti, bwd_ti=bwd_ti, dw_staged=dw_staged, x3=x3,
                    fwd_nw=fwd_nw, bwd_nw=bwd_nw, f32_mfma=f32_mfma, reg_tile=reg_tile, lds_tile=lds_tile, x2h=x2h)
    return sdf, devnet, col, ren


def measure_train(args, ctx, with_cpu=True):
    """One measurement of `args` on the process group `ctx` (init_distributed): warm-up, K timed steps, the line as a dict on
    rank 0 (None elsewhere).  Emits nothing and leaves the process group alone."""
    import torch
    import torch.distributed as dist
    world, rank, dev, backend, rehearsal = ctx

    # the measured leg uses the product only (package + librnbneus_hip.so); oracle/ is touched by
    # cpu_baseline() alone
    import rnb_neus_fork_amd as R
    from rnb_neus_fork_amd import parallel as P
    from rnb_neus_fork_amd.synthetic import synthetic_batch
    lib = R.native.load()

    S = args.samples
    sdf, devnet, col, ren = build_model(R, dev, S, args.dtype, args.deterministic, args.fwd_ti, args.bwd_ti, args.dw_staged, args.x3,
                                        args.fwd_nw, args.bwd_nw, args.f32_mfma, args.reg_tile, args.lds_tile, args.x2h)
    exact_dp = not args.torch_train_ops     # the reference's torch-op loss knows nothing about shards
    if world > 1:
        P.broadcast_parameters([sdf, devnet, col])
        ren.set_data_parallel(exact=exact_dp)
    params = list(sdf.parameters()) + list(devnet.parameters())
    if not args.no_albedo:
        params += list(col.parameters())
    # exp_runner.py:115: Adam, lr 5e-4.  Default: the library's flat single-launch Adam (identical update rule,
    # tests/test_gpu_train_ops.py); --torch-train-ops keeps torch.optim.Adam and the reference's chain of torch ops
    # for the loss, i.e. exactly what exp_runner.py would run around the drop-in renderer.
    if args.torch_train_ops:
        opt = torch.optim.Adam(params, lr=5e-4, fused=True)
    else:
        opt = R.FlatAdam(params, lr=5e-4)

    if args.scaling == "strong":
        if args.global_rays % world:
            raise SystemExit(f"--global-rays {args.global_rays} is not divisible by {world} ranks")
        B = args.global_rays // world
    else:
        B = args.rays
    n_batches = 8
    # inputs resident in HBM before the timed region: each rank owns its contiguous shard of a global batch
    batches = []
    for i in range(n_batches):
        gb = synthetic_batch(B * world, seed=0, step=i, warmup=args.warmup_mode)
        mine = P.shard_batch(gb, rank, world, n_rays=B * world)
        batches.append({k: v.to(dev) for k, v in mine.items()})

    group = dist.group.WORLD if (world > 1 and exact_dp) else None
    forward_only = args.mode == "render"
    capture = None
    if args.device_rays:
        if forward_only or args.warmup_mode:
            raise SystemExit("--device-rays measures the train_rnb step in main mode")
        V, Hh, Ww = (int(x) for x in args.stack.lower().split("x"))
        capture = synthetic_capture(R, dev, V, Hh, Ww)
        torch.cuda.synchronize()
        if rank == 0:
            gb = sum(t.numel() * 4 for t in (capture.images, capture.light_directions, capture.masks)) / 1e9
            print(f"[bench] synthetic capture {V} x {Hh} x {Ww} resident in HBM: {gb:.2f} GB", file=sys.stderr, flush=True)
    use_capture = [capture is not None]
    last_out = [None]      # the dict of the last render (sizes only: SURVEY 8(d)'s algorithmic bytes)

    def step(i):
        if forward_only:
            # NeuSRenderer.render under no_grad (models/renderer.py:556-648; validate_image's call): sampling + fine
            # forward + composite, nothing saved for a backward (FLAG_FORWARD_ONLY)
            b = batches[i % n_batches]
            with torch.no_grad():
                out = ren.render(b["rays_o"], b["rays_d"], b["near"], b["far"], cos_anneal_ratio=1.0, t_rand=b["t_rand"])
            last_out[0] = out
            return out["color_fine"].sum()
        if use_capture[0]:
            # exp_runner.py:174-220 on the device: pixel draw, ray / target / per-pixel light gather, near / far — one
            # launch (+ two torch.randint) inside the step; the perturbation draw is the renderer's own torch.rand
            b = capture.sample(i % capture.n_images, B)
            b["t_rand"] = None
        else:
            b = batches[i % n_batches]
        fn = ren.render_rnb_warmup if args.warmup_mode else ren.render_rnb
        out = fn(b["rays_o"], b["rays_d"], b["near"], b["far"], b["lights_dir"], cos_anneal_ratio=1.0,
                 no_albedo=args.no_albedo, t_rand=b["t_rand"])
        last_out[0] = out
        if args.torch_train_ops:
            loss, _ = torch_rnb_loss(out, b["true_rgb"], b["mask"])
        else:
            loss, _ = R.rnb_loss(out, b["true_rgb"], b["mask"], group=group, report_global=False)   # (no per-step logging here)
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
    use_events = not args.no_gemm_events and rank == 0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    if use_events:
        lib.rnb_profile_enable(1)
    if world > 1:
        P.time_collectives(True)     # event pairs around the step's all-reduces (normalisers + flat gradient buffer)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step(args.warmup + i)
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    gemm_ms, gemm_n, gemm_fl = C.c_double(0), C.c_int64(0), C.c_double(0)
    if use_events:
        R.native.check(lib.rnb_profile_collect(C.byref(gemm_ms), C.byref(gemm_n), C.byref(gemm_fl)))
        lib.rnb_profile_enable(0)
    coll_ms, coll_n = (0.0, 0)
    if world > 1:
        coll_ms, coll_n = P.collective_ms()
        P.time_collectives(False)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2]
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    final_loss = float(loss.detach())
    if rank == 0:
        print(f"[bench] timed region done: {elapsed:.3f} s for {args.steps} steps", file=sys.stderr, flush=True)
    prestaged_ms = None
    if capture is not None:
        # the same steps on pre-staged batches, in the same process: what the in-step ray generation costs
        use_capture[0] = False
        for i in range(3):
            step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        barrier()
        prestaged_ms = 1e3 * (time.perf_counter() - t1) / args.steps

    if rank == 0:
        flags = (R.native.MODE_CORE if forward_only else
                 R.native.MODE_MVPS | (R.native.FLAG_NO_ALBEDO if args.no_albedo else 0))
        tf, ff = C.c_double(), C.c_double()
        R.native.check(lib.rnb_algorithmic_flops(C.byref(ren.desc), B, flags, C.byref(tf), C.byref(ff)))
        if forward_only:
            tf = ff           # forward-only: (112 + 2 x 128) SDF sweeps + 128 albedo evaluations per ray
        ms_per_step = 1e3 * elapsed / args.steps
        value = B * world * args.steps / elapsed
        bf16 = args.dtype == "bf16"
        roof = None
        traffic_file = ("hbm_traffic_bf16.json" if bf16 else
                        ("r02_hbm_traffic_f32_mfma.json" if args.f32_mfma else "hbm_traffic.json"))
        bid = R.native.build_id()
        tr, tr_note = (None, None) if (forward_only or capture is not None) else measured_traffic(traffic_file, world, B, S, bid)
        if use_events and gemm_n.value > 0:
            ach = gemm_fl.value / (gemm_ms.value * 1e-3) / 1e12
            step_tf = tf.value / (ms_per_step * 1e-3) / 1e12
            common = {"launches_per_step": gemm_n.value / args.steps,
                      "avg_launch_us": round(1e3 * gemm_ms.value / gemm_n.value, 2),
                      "flop_per_launch": round(gemm_fl.value / gemm_n.value, 1),
                      "gemm_ms_per_step": round(gemm_ms.value / args.steps, 3),
                      "step_algorithmic_tflops": round(step_tf, 3)}
            if not bf16:
                # dtype f32.  SURVEY 8(d): the path is MFMA-bound (~7 MB of algorithmic HBM traffic per 530-GFLOP step), so
                # `roofline` is the MATRIX roof of the dominant kernel class: algorithmic fp32 FLOPs (2 M N K of the real layer
                # shapes, never the split terms) / its device time (HIP events on the launch stream) / (dense peak of the MFMA
                # dtype it issues / terms per fp32 product).  The family, the whole step and the design's saved-state HBM
                # figure are kept beside it (`family`, `step`, `hbm_state`).
                x3 = not args.f32_mfma
                x2h = x3 and args.x2h is not False
                terms = X2H_TERMS if x2h else None
                base_peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if x3 else FP32_MFMA_PEAK_TFLOPS
                by = kernel_classes(lib, args.steps, base_peak, tr, terms)
                fam_peak = base_peak
                if x2h and by:
                    # mixed arithmetic: the ceiling of the family is the rate at which its classes would finish if each ran
                    # at its own MFMA peak (flop-weighted harmonic mean): frac = ideal matrix time / measured time
                    ideal_ms = sum(k["flop_per_step"] / (class_peak(k["kernel_class"], base_peak, terms) * 1e12) * 1e3 for k in by)
                    fam_peak = sum(k["flop_per_step"] for k in by) / (ideal_ms * 1e-3) / 1e12
                dom = by[0] if by else None
                busy = measured_mfma_busy(bid)
                dom_kernel = CLASS_KERNELS.get(dom["kernel_class"]) if dom else None
                dom_terms = (terms or {}).get(dom["kernel_class"]) if dom else None
                dom_terms = dom_terms or (6 if x3 else 1)
                dom_peak = class_peak(dom["kernel_class"], base_peak, terms) if dom else fam_peak
                dom_busy = None
                if busy and dom_kernel:
                    keys = dom_kernel if isinstance(dom_kernel, tuple) else (dom_kernel,)
                    hits = [v for k, v in busy.items() if any(kk in k for kk in keys)]
                    if hits:
                        dom_busy = round(max(h["mfma_busy"] for h in hits), 4)
                roof = {"bound": "mfma",
                        "achieved": dom["tflops"] if dom else round(ach, 3), "peak": round(dom_peak, 1), "unit": "TFLOP/s",
                        "frac": round((dom["tflops"] if dom else ach) / dom_peak, 4),
                        "kernel": (f"{dom['kernel_class']}: " + (" / ".join(dom_kernel) if isinstance(dom_kernel, tuple) else str(dom_kernel))) if dom else None,
                        "terms": dom_terms,
                        "mfma_dtype": ("fp32 (v_mfma_f32_32x32x2_f32)" if not x3 else
                                       "fp16 (v_mfma_f32_32x32x16_f16)" if dom_terms == 3 else "bf16 (v_mfma_f32_32x32x16_bf16)"),
                        "peak_basis": (f"{BF16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s dense 16-bit MFMA / {dom_terms} matrix terms per fp32 product"
                                       if x3 else "157.3 TFLOP/s dense fp32 MFMA"),
                        "frac_of_fp32_mfma_peak": round((dom["tflops"] if dom else ach) / FP32_MFMA_PEAK_TFLOPS, 4),
                        "avg_launch_us": round(1e3 * dom["ms_per_step"] / dom["launches_per_step"], 2) if dom else None,
                        "flop_per_launch": round(dom["flop_per_step"] / dom["launches_per_step"], 1) if dom else None,
                        "launches_per_step": dom["launches_per_step"] if dom else None,
                        "mfma_busy_pmc": dom_busy,
                        "mfma_busy_note": (None if dom_busy is not None else
                                           "no SQ_VALU_MFMA_BUSY_CYCLES pass of this build under profiles/sq_counters.json"),
                        "traffic": (round(dom["hbm_gb_per_step"] * 1e9 / dom["launches_per_step"]) if dom and "hbm_gb_per_step" in dom else None),
                        "traffic_unit": "HBM bytes per launch of this kernel (PMC: 2 x FETCH_SIZE + WRITE_SIZE)",
                        "traffic_note": tr_note,
                        "family": {"kernels": ("split-operand MFMA family: fused_forward/reverse_kernel<.., true, true>, fused_fb_h2_kernel, "
                                               "gemm_dw_x3_kernel<0, 2>, color_fwd/bwd_h2_kernel (three fp16 terms); fused_ra_kernel<.., true> "
                                               "(six bf16 terms)" if x2h else
                                               "x3 MFMA family: fused_forward/reverse/ra/fb_kernel<.., true>, gemm_dw_x3_kernel, "
                                               "gemm_rows_x3m_kernel<*>" if x3 else
                                               "fp32-MFMA family: fused_forward/reverse/ra/fb_kernel, gemm_dw_direct_kernel, gemm_rows_kernel<*>"),
                                   "achieved": round(ach, 3), "peak": round(fam_peak, 1), "unit": "TFLOP/s",
                                   "frac": round(ach / fam_peak, 4),
                                   "peak_basis": ("2500 / terms per kernel class (three fp16 terms: 833.3; RA sweep, six bf16 terms: 416.7), "
                                                  "flop-weighted harmonic mean: frac = ideal matrix time / measured time" if x2h else
                                                  "2500 / 6 bf16 terms per fp32 product" if x3 else "157.3 TFLOP/s dense fp32 MFMA"),
                                   "frac_of_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4)},
                        "step": {"algorithmic_flop": tf.value, "algorithmic_tflops": round(step_tf, 3),
                                 "frac": round(step_tf / fam_peak, 4),
                                 "frac_of_fp32_mfma_peak": round(step_tf / FP32_MFMA_PEAK_TFLOPS, 4)}}
                roof["family"].update(common)
                if by:
                    roof["by_kernel_class"] = by
                # SURVEY 8(d)'s algorithmic HBM bytes: what a fully fused step must move — the ray inputs, the returned tensors,
                # one read of the weights and one write of their gradients — against what the PMC counters saw
                try:
                    io = sum(v.numel() * v.element_size() for v in batches[0].values() if torch.is_tensor(v))
                    io += sum(v.numel() * v.element_size() for v in last_out[0].values() if torch.is_tensor(v)) if last_out[0] else 0
                    io += 2 * sum(q.numel() * 4 for q in params)
                    roof["algorithmic_bytes_per_step"] = io
                    roof["algorithmic_bytes_basis"] = ("SURVEY 8(d): ray inputs + returned tensors + one read of the trained leaves "
                                                       "+ one write of their gradients")
                    if tr:
                        roof["traffic_bytes_per_step"] = round(tr["hbm_bytes_per_step"])
                        roof["traffic_ratio"] = round(tr["hbm_bytes_per_step"] / io, 1)
                        roof["traffic_source"] = "profiles/" + traffic_file
                except Exception as e:      # (never let bookkeeping kill the line)
                    roof["algorithmic_bytes_note"] = repr(e)
                if x3 and not forward_only:
                    # the DESIGN's own floor: the fp32 per-point saved state written once and read once per consumer between
                    # the launches of the unfused decomposition (not SURVEY 8(d)'s algorithmic bytes)
                    try:
                        ab = C.c_double()
                        R.native.check(lib.rnb_algorithmic_bytes(C.byref(ren.desc), B, flags, C.byref(ab)))
                        gbs = ab.value * args.steps / (gemm_ms.value * 1e-3) / 1e9
                        roof["hbm_state"] = {"design_state_bytes_per_step": ab.value, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                             "basis": ("rnb_algorithmic_bytes: every saved-state matrix of the design written once and "
                                                       "read once per consumer / device time of the MFMA-family launches")}
                    except Exception:
                        pass
            else:
                # bf16 sweeps: 1/16 of the fp32 matrix time, so the per-point saved state decides: the bound is HBM.
                # achieved = algorithmic bytes of the MFMA-family launches (each saved-state matrix written once and
                # read by each consumer once: DESIGN 4b) / their device time.
                alg = None
                try:
                    ab = C.c_double()
                    R.native.check(lib.rnb_algorithmic_bytes(C.byref(ren.desc), B, flags, C.byref(ab)))
                    alg = ab.value
                except Exception:
                    alg = None
                gbs = (alg * args.steps / (gemm_ms.value * 1e-3) / 1e9) if alg else None
                roof = {"bound": "hbm", "achieved": round(gbs, 1) if gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4) if gbs else None,
                        "traffic": round(tr["hbm_bytes_per_launch"]) if tr else None,
                        "traffic_unit": "HBM bytes per launch",
                        "traffic_note": tr_note,
                        "algorithmic_bytes_per_step": alg,
                        "kernel": "bf16-MFMA family: bf_forward/reverse/ra/fb_kernel, bf_color_fwd/bwd_kernel, bf_dw_kernel",
                        "mfma_tflops": round(ach, 2), "mfma_frac_of_bf16_peak": round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
                        "step_frac_of_bf16_peak": round(step_tf / BF16_MFMA_PEAK_TFLOPS, 4)}
                roof.update(common)
                if tr:   # the same rate on the bytes the PMC counters saw (independent of the design's own byte count)
                    pg = tr["hbm_bytes_per_step"] * args.steps / (gemm_ms.value * 1e-3) / 1e9
                    roof["pmc"] = {"hbm_bytes_per_step": round(tr["hbm_bytes_per_step"]), "achieved": round(pg, 1), "unit": "GB/s",
                                   "frac": round(pg / HBM_PEAK_GBS, 4), "source": "profiles/" + traffic_file}
                by = kernel_classes(lib, args.steps, BF16_MFMA_PEAK_TFLOPS, tr)
                if by:
                    roof["by_kernel_class"] = by
        cpu = None
        if world == 1 and not args.no_cpu_baseline and with_cpu:
            cpu = (cpu_baseline_render(B, S, args.cpu_steps, args.cpu_warmup) if forward_only else
                   cpu_baseline(B, S, args.cpu_steps, args.warmup_mode, args.no_albedo, args.cpu_warmup))
        line = {
            "metric": (f"forward-only rays/sec (NeuSRenderer.render, no_grad) at {B} rays x {S} samples/ray" if forward_only
                       else f"training rays/sec at {B} rays x {S} samples/ray"),
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_median": round(median_ms, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            # `dtype` is the type of the results and of everything stored; the products themselves:
            "arithmetic": ("bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16); masters, gradients, epilogues fp32" if bf16 else
                           ("fp32 MFMA (v_mfma_f32_32x32x2_f32)" if args.f32_mfma else
                            "x3: every fp32 operand as 3 bf16 terms (hi + mid + lo = x exactly), 6 of the 9 cross terms per product "
                            "on v_mfma_f32_32x32x16_bf16, fp32 accumulate; dropped terms < 2^-26 |ab|" if args.x2h is False else
                            "x2h: fp32 operands as 2 fp16 terms after a power-of-two scale taken from the data (per matrix for weights, per "
                            "64-point tile and layer for activations / Jacobian rows / adjoints: no operand range), 3 of the 4 cross terms per product on "
                            "v_mfma_f32_32x32x16_f16, fp32 accumulate; operand representation <= 2^-22 (rms 2^-23.6); the RA sweep "
                            "keeps x3 (3 bf16 terms per operand, 6 cross terms)")),
            "build_id": bid,
            "rccl_ranks": world if backend == "nccl" else (1 if world == 1 else 0),
            # the flat gradient buffer is all-reduced after the backward sweeps have finished (renderer.py: one all_reduce on
            # the compute stream): nothing overlaps it — 2.7 MB, latency-bound on xGMI
            "grad_allreduce_overlap": (False if world > 1 else None),
            # device time of rank 0's collectives per step (HIP events on the compute stream around each all-reduce: the 4-float
            # normalisers in the loss + the flat gradient buffer in the backward), so an N > 1 line decomposes itself
            "allreduce_ms_per_step": (round(coll_ms / args.steps, 4) if world > 1 else None),
            "allreduces_per_step": (coll_n / args.steps if world > 1 else None),
            "config": {"workload": "DiLiGenT-MV-shaped synthetic rays, wmask_rnb.conf networks (8x256 SDF MLP + "
                                   "2x256 albedo MLP), train_rnb step "
                                   f"({'render_rnb_warmup' if args.warmup_mode else 'render_rnb'}), "
                                   f"{B} rays x ({S // 2}+{S // 2}) samples per GPU, 3 lights, geometric init, Adam"
                                   + (f"; strong scaling of a {args.global_rays}-ray global batch" if args.scaling == "strong" else ""),
                       "rays_per_gpu": B, "global_rays": B * world, "samples_per_ray": S, "n_lights": 3,
                       "no_albedo": bool(args.no_albedo), "parallelism": f"dp{world}", "final_loss": final_loss,
                       "dp_loss": ("exact large-batch (one 4-float all-reduce of the normalisers + SUM of gradients)" if (world > 1 and exact_dp)
                                   else ("DDP mean of per-rank losses" if world > 1 else "single process")),
                       "backend": backend, "deterministic": bool(args.deterministic),
                       # the products behind `dtype` (kept here too: the driver's parsed record keeps `config`)
                       "arithmetic": ("bf16" if bf16 else "fp32_mfma" if args.f32_mfma else "x3" if args.x2h is False else "x2h"),
                       "operand_bits": (8 if bf16 else 24 if (args.f32_mfma or args.x2h is False) else 22),
                       "train_ops": ("torch ops loss + torch.optim.Adam(fused)" if args.torch_train_ops
                                     else "rnb_loss_rnb + rnb_adam_step (one launch each)")},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if forward_only:
            line["config"]["workload"] = ("DiLiGenT-MV-shaped synthetic rays, wmask_rnb.conf networks, NeuSRenderer.render "
                                          f"under no_grad (sampling + fine forward + composite), {B} rays x "
                                          f"({S // 2}+{S // 2}) samples per GPU, geometric init")
            line["config"]["train_ops"] = None
        if capture is not None:
            line["config"]["device_rays"] = {
                "stack": args.stack, "hbm_gb": round(sum(t.numel() * 4 for t in (capture.images, capture.light_directions,
                                                                                  capture.masks)) / 1e9, 2),
                "ms_per_step_with_in_step_ray_generation": round(ms_per_step, 3),
                "ms_per_step_prestaged_batches": round(prestaged_ms, 3),
                "delta_ms": round(ms_per_step - prestaged_ms, 3),
                "note": "DeviceRays.sample() (2 x torch.randint + rnb_gen_rays_at_view) and the renderer's own torch.rand "
                        "inside the timed step; `value` is this configuration"}
        if rehearsal:
            line["rehearsal"] = True
            line["config"]["note"] = ("ranks share ONE GPU over gloo: functional rehearsal of the N > 1 path, not a "
                                      "scaling measurement")
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        return line
    return None


# the configurations the default --gpus 1 run measures beside the headline: BASELINE configs 3 and 5
# (1-GPU legs), the forward-only render, and the native-fp32-MFMA arithmetic — short passes, in this process, under `also`
ALSO_LEGS = (
    ("config 3: wmask_rnb_noalbedo.conf (normal-only loss path)", dict(no_albedo=True)),
    ("config 5 (1-GPU leg): bf16 sweeps, 256 samples per ray", dict(dtype="bf16", samples=256)),
    ("NeuSRenderer.render, forward only (no_grad)", dict(mode="render")),
    ("deterministic reductions (RNB_VARIANT_DETERMINISTIC: ordered slab sums instead of fp32 atomics, bit-reproducible)",
     dict(deterministic=True)),
    ("A/B: six bf16 terms in every product (RNB_VARIANT_NO_X2H: the round-3 arithmetic)", dict(x2h=False)),
    ("A/B: native fp32 MFMA arithmetic (RNB_VARIANT_F32_MFMA)", dict(f32_mfma=True)),
)


def run_train(args):
    import copy
    import torch.distributed as dist
    ctx = init_distributed(args)
    world, rank = ctx[0], ctx[1]
    line = measure_train(args, ctx, with_cpu=False)
    plain = (world == 1 and args.mode == "train" and not args.no_also and not args.device_rays and not args.no_albedo
             and args.dtype == "f32" and not args.f32_mfma and args.x2h is None and not args.warmup_mode and args.rays == 512
             and args.samples == 128)
    if plain:
        also = []
        for name, kw in ALSO_LEGS:
            la = copy.copy(args)
            for k, v in kw.items():
                setattr(la, k, v)
            la.steps, la.warmup = args.also_steps, 5
            import gc
            import torch
            gc.collect()
            torch.cuda.empty_cache()    # (each leg's workspaces have their own sizes: start it from an unfragmented pool)
            l = measure_train(la, ctx, with_cpu=False)
            r = l.get("roofline") or {}
            also.append({"leg": name, "metric": l["metric"], "value": l["value"], "unit": l["unit"], "ms_per_step": l["ms_per_step"],
                         "steps": l["steps"], "warmup": l["warmup"], "dtype": l["dtype"], "arithmetic": l["arithmetic"],
                         "config": {"workload": l["config"]["workload"], "no_albedo": l["config"]["no_albedo"],
                                    "samples_per_ray": l["config"]["samples_per_ray"]},
                         "roofline": {"bound": r.get("bound"), "frac": r.get("frac"), "achieved": r.get("achieved"),
                                      "peak": r.get("peak"), "unit": r.get("unit"), "kernel": r.get("kernel"),
                                      "family_frac": (r.get("family") or {}).get("frac"),
                                      "step_frac": (r.get("step") or {}).get("frac", r.get("step_frac")),
                                      "pmc": r.get("pmc")}})
            print(f"[bench] also: {name}: {l['value']:.0f} {l['unit']}, {l['ms_per_step']} ms/step", file=sys.stderr, flush=True)
        line["also"] = also
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            forward_only = args.mode == "render"
            B = args.rays if args.scaling == "weak" else args.global_rays // world
            cpu = (cpu_baseline_render(B, args.samples, args.cpu_steps, args.cpu_warmup) if forward_only else
                   cpu_baseline(B, args.samples, args.cpu_steps, args.warmup_mode, args.no_albedo, args.cpu_warmup))
            line["cpu_baseline"] = cpu
            line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_mesh(args):
    """validate_mesh's SDF grid (models/renderer.py:10-25, exp_runner.py:561-581): resolution^3 forward-only SDF
    evaluations, x-slabs sharded over the ranks, all-gather of the slabs.  A step = one whole grid."""
    import torch
    import torch.distributed as dist
    world, rank, dev, backend, rehearsal = init_distributed(args)
    import rnb_neus_fork_amd as R
    lib = R.native.load()
    sdf, devnet, col, ren = build_model(R, dev, 128, args.dtype, False, x3=args.x3, f32_mfma=args.f32_mfma, fwd_ti=args.fwd_ti,
                                        fwd_nw=args.fwd_nw, reg_tile=args.reg_tile, lds_tile=args.lds_tile, x2h=args.x2h)
    if world > 1:
        from rnb_neus_fork_amd import parallel as P
        P.broadcast_parameters([sdf, devnet, col])
        ren.set_data_parallel()
    res = args.resolution
    bmin = torch.tensor([-1.01, -1.01, -1.01])
    bmax = torch.tensor([1.01, 1.01, 1.01])

    def step():
        return ren.extract_fields(bmin, bmax, res, to_host=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    use_events = not args.no_gemm_events and rank == 0
    barrier()
    if use_events:
        lib.rnb_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        u = step()
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
    # the step after the grid in validate_mesh (models/renderer.py:31): marching cubes on the volume still in HBM
    # (csrc/mcubes.hip; every rank holds the whole volume after the all-gather, rank 0 meshes it)
    mc = None
    if rank == 0 and not args.no_marching_cubes:
        v, tri = R.marching_cubes(u, 0.0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            v, tri = R.marching_cubes(u, 0.0)
        torch.cuda.synchronize()
        mc_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        alg = 4.0 * res ** 3 + 24.0 * v.shape[0] + 12.0 * tri.shape[0]     # volume once + the two output arrays
        mc = {"ms": round(mc_ms, 3), "vertices": int(v.shape[0]), "triangles": int(tri.shape[0]),
              "bound": "hbm", "algorithmic_bytes": alg, "achieved_GBps": round(alg / (mc_ms * 1e-3) / 1e9, 1),
              "frac": round(alg / (mc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              "note": "4 passes over the volume (count, scan, vertices, triangles) + one 4-byte id word per grid point; "
                      "includes the host read of the two output sizes; parity with PyMCubes unpinned"}
    if rank == 0:
        n = res ** 3
        ms = 1e3 * elapsed / args.steps
        roof = None
        if use_events and gemm_n.value > 0:
            ach = gemm_fl.value / (gemm_ms.value * 1e-3) / 1e12
            x3 = args.dtype != "bf16" and not args.f32_mfma   # fp32 products as six bf16 MFMA terms (see bench_train)
            peak = (BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else
                    (BF16_MFMA_PEAK_TFLOPS / 6.0 if x3 else FP32_MFMA_PEAK_TFLOPS))
            roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None,
                    "peak_basis": ("2500 TFLOP/s dense bf16 MFMA" if args.dtype == "bf16" else
                                   ("2500 TFLOP/s dense bf16 MFMA / 6 bf16 terms per fp32 product (x3 arithmetic)" if x3
                                    else "157.3 TFLOP/s dense fp32 MFMA")),
                    "frac_of_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                    "kernel": "fused forward-only SDF sweep with in-kernel grid-point generation",
                    "launches_per_step": gemm_n.value / args.steps,
                    "avg_launch_us": round(1e3 * gemm_ms.value / gemm_n.value, 2),
                    "flop_per_launch": round(gemm_fl.value / gemm_n.value, 1),
                    "step_frac": round(gemm_fl.value / args.steps / (ms * 1e-3) / 1e12 / peak, 4)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline_mesh(1 << 20)
        line = {"metric": f"SDF grid points/sec of validate_mesh at {res}^3", "value": round(n * args.steps / elapsed, 1),
                "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "rccl_ranks": world if backend == "nccl" else (1 if world == 1 else 0),
                "config": {"workload": f"extract_fields: {res}^3 SDF evaluations of the wmask_rnb.conf SDF network "
                                       "(geometric init), x-slabs sharded over the ranks, volume resident in HBM",
                           "resolution": res, "parallelism": f"dp{world}", "grid_mean": float(u.mean())},
                "roofline": roof, "cpu_baseline": cpu, "marching_cubes": mc}
        if mc:
            line["config"]["mesh_ms_total"] = round(ms + mc["ms"], 3)     # grid + marching cubes = extract_geometry
        if rehearsal:
            line["rehearsal"] = True
        if cpu:
            line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


JSON_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries (gloo's connection banner, the ROCm runtime) print
    to file descriptor 1 behind Python's back, so fd 1 is pointed at stderr for the life of the process and the JSON
    line goes to a private duplicate of the original stdout."""
    global JSON_OUT
    sys.stdout.flush()
    JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def emit(line):
    JSON_OUT.write(json.dumps(line) + "\n")
    JSON_OUT.flush()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    claim_stdout()
    if args.mode == "mesh":
        run_mesh(args)
    else:
        run_train(args)


if __name__ == "__main__":
    main()
