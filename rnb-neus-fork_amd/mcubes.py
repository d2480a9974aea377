"""Marching cubes on the device: the step of `extract_geometry` (models/renderer.py:27-36) that the reference
delegates to PyMCubes (`mcubes.marching_cubes(u, threshold)`).  All arithmetic is in librnbneus_hip.so
(csrc/mcubes.hip); there is no CPU path.  Parity with PyMCubes is unpinned (not importable here): see DESIGN.md."""
from __future__ import annotations

import ctypes as C

import torch

from . import native


def marching_cubes(volume: torch.Tensor, threshold: float = 0.0):
    """volume: [nx, ny, nz] fp32 CUDA tensor (what `NeuSRenderer.extract_fields(..., to_host=False)` returns).
    Returns device tensors (vertices [V, 3] float64 in grid-index coordinates, triangles [T, 3] int32), like
    `mcubes.marching_cubes` returns arrays.  One host synchronisation (the two output sizes)."""
    if not volume.is_cuda:
        raise RuntimeError("marching_cubes: the volume must live on the GPU (there is no CPU path)")
    if volume.dim() != 3:
        raise ValueError(f"marching_cubes: expected a 3-D volume, got shape {tuple(volume.shape)}")
    lib = native.load()
    u = volume.detach().to(torch.float32).contiguous()
    nx, ny, nz = u.shape
    nbytes = C.c_int64()
    native.check(lib.rnb_marching_cubes_workspace_bytes(nx, ny, nz, C.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=u.device)
    counts = torch.empty(2, dtype=torch.int64, device=u.device)
    with native.on_device(u) as stream:
        native.check(lib.rnb_marching_cubes_count(native.ptr(u), nx, ny, nz, float(threshold), native.ptr(ws),
                                                  ws.numel(), native.ptr(counts), stream))
    nv, nt = (int(c) for c in counts.cpu())
    vertices = torch.empty(nv, 3, dtype=torch.float64, device=u.device)
    triangles = torch.empty(nt, 3, dtype=torch.int32, device=u.device)
    with native.on_device(u) as stream:
        native.check(lib.rnb_marching_cubes_emit(native.ptr(u), nx, ny, nz, float(threshold), native.ptr(ws),
                                                 ws.numel(), nv, nt, native.ptr(vertices), native.ptr(triangles),
                                                 stream))
    return vertices, triangles
