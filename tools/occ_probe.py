"""Development probe (GPU box): how much do the two workgroups of a CU overlap in the forward-only sweep?  Times the 64-point-tile
kernel on 128 .. 4096 tiles (256 tiles = one workgroup per CU, 512 = two, 1024 = two rounds of two).
usage: python tools/occ_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rnb_neus_fork_amd as R  # noqa: E402
from rnb_neus_fork_amd import runtime  # noqa: E402
from tests.golden_util import Golden  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = Golden("full_main_b512")
    sdf, devn, col, ren = R.build_from_named_params(g.mc, g.params(), dev)
    ren.set_variant(fwd_ti=2)
    packed = ren._pack(True)
    big = (torch.rand(1 << 20, 3, device=dev) * 2 - 1) * 0.9
    for tiles in (128, 256, 384, 512, 768, 1024, 2048, 4096):
        pts = big[: tiles * 64].contiguous()
        for _ in range(5):
            runtime.sdf_forward(ren.desc, packed, pts, False)
        torch.cuda.synchronize()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            runtime.sdf_forward(ren.desc, packed, pts, False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{tiles:5d} tiles of 64 points: {dt * 1e6:8.1f} us per forward-only sweep  ({dt * 1e6 / max(1, tiles / 512):7.1f} us per 512 tiles)", flush=True)


if __name__ == "__main__":
    main()
