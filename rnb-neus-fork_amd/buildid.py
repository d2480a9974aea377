"""Build identity of librnbneus_hip.so: a hash of every source the library is compiled from and of the compiler flags.

The id is compiled into the library (`rnb_build_id()`), written into every profile JSON at collection time
(tools/traffic_summary.py, bench.py) and compared again when bench.py quotes stored PMC traffic: bytes measured on another
build of the kernels are not reported.  No dependencies (the build script loads this file by path, before the package)."""
from __future__ import annotations

import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

HIP_SOURCES = ["api.hip", "layout.hip", "mlp.hip", "weightnorm.hip", "sampling.hip", "composite.hip", "prof.hip", "fused.hip",
               "sweep_mv.hip", "fused_bwd.hip", "color_h2.hip", "bf16.hip", "train.hip", "raygen.hip", "mcubes.hip"]
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
EXTRA_FLAGS = {
    # sampling.hip must round like the reference's separate PyTorch ops (no fused multiply-add contraction)
    "sampling.hip": ["-ffp-contract=off"], "raygen.hip": ["-ffp-contract=off"],
    # the epilogue chains of the M/V sweeps' vector waves are written stage by stage (independent chains): the SLP
    # vectoriser would fuse neighbouring chains into packed fp32 instructions (slower beside the partner's MFMAs)
    "sweep_mv.hip": ["-fno-slp-vectorize"],
}


def source_files(pkg_dir: str = _HERE):
    """Every file the library is built from, as (name, absolute path), in a fixed order."""
    csrc = os.path.join(pkg_dir, "csrc")
    names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".h", ".inc")))
    files = [("csrc/" + n, os.path.join(csrc, n)) for n in names]
    files.append(("include/rnbneus.h", os.path.join(os.path.dirname(pkg_dir), "include", "rnbneus.h")))
    return files


def source_build_id(pkg_dir: str = _HERE, overrides: dict | None = None) -> str:
    """16 hex digits over (file names, file contents, flags).  `overrides` maps a file name (as in source_files) to the
    bytes to hash instead of the file's contents (tests)."""
    h = hashlib.sha256()
    for name, path in source_files(pkg_dir):
        data = overrides[name] if overrides and name in overrides else open(path, "rb").read()
        h.update(name.encode() + b"\0" + str(len(data)).encode() + b"\0" + data)
    h.update(repr((HIP_SOURCES, COMMON_FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:16]
