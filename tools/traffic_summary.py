"""Builds profiles/hbm_traffic.json from two rocprofv3 PMC passes (development / evidence tool).
usage: python tools/traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> <rays>
           [<samples per ray> [<output json under profiles/>]]"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
steps, rays = int(sys.argv[3]), int(sys.argv[4])
samples = int(sys.argv[5]) if len(sys.argv) > 5 else 128
out_name = sys.argv[6] if len(sys.argv) > 6 else "hbm_traffic.json"
fam = [k for k in set(f) | set(w) if "gemm_" in k or "fused_" in k or "bf_" in k or "color_fwd_h2" in k or "color_bwd_h2" in k]
per_kernel = {}
tot_f = tot_w = n = 0
for k in sorted(fam):
    nf = len(f.get(k, []))
    per_kernel[k] = {"launches_per_step": nf / steps,
                     "fetch_size_raw_kb_per_launch": sum(f.get(k, [0])) / max(nf, 1),
                     "write_size_kb_per_launch": sum(w.get(k, [0])) / max(len(w.get(k, [])), 1)}
    tot_f += sum(f.get(k, [0]))
    tot_w += sum(w.get(k, [0]))
    n += nf
# the build the passes ran on: the library in this tree right now (collect and summarise in one go, on one box)
import rnb_neus_fork_amd as R  # noqa: E402

out = {"build_id": R.native.build_id(), "rays": rays, "samples": samples, "steps": steps, "launches_per_step": n / steps,
       "fetch_size_raw_bytes_per_step": tot_f / steps * 1024,
       "fetch_bytes_per_step_corrected_x2": 2 * tot_f / steps * 1024,
       "write_bytes_per_step": tot_w / steps * 1024,
       "hbm_bytes_per_step": (2 * tot_f + tot_w) / steps * 1024,
       "hbm_bytes_per_launch": (2 * tot_f + tot_w) / steps * 1024 / (n / steps),
       "per_kernel": per_kernel,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; MFMA-family "
               "kernels only (gemm_*, fused_*, color_*_h2, bf_*). FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
               "requests at 64 B); WRITE_SIZE taken as is. Infinity-Cache hits are included in FETCH_SIZE."}
json.dump(out, open("profiles/" + out_name, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
