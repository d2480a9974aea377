"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc pass (development / evidence tool).
usage: python tools/sq_summary.py <sq_counter_collection.csv> [<kernel_trace.csv of the same run> [<out.json under profiles/>]]
With a third argument the per-kernel figures are also written as JSON with the build id of the library in this tree
(bench.py quotes `roofline.mfma_busy_pmc` from profiles/sq_counters.json for the build it was collected on).
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); with the kernel trace also the clock the chip
held: (GRBM_GUI_ACTIVE / 8) / duration."""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[re.sub(r"\(.*", "", r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("# matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), per kernel")
per_kernel = {}
for k, v in agg.items():
    if not v.get("SQ_VALU_MFMA_BUSY_CYCLES") or sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) == 0:
        continue
    busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
    clk = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) / 8.0
    wait = sum(v.get("SQ_WAIT_ANY", [0])) / max(len(v.get("SQ_WAIT_ANY", [0])), 1)
    wave = sum(v.get("SQ_WAVE_CYCLES", [1])) / max(len(v.get("SQ_WAVE_CYCLES", [1])), 1)
    line = f"{k[:64]:66s} matrix pipe busy {100 * busy / (clk * 1024):5.1f} %   waves parked {100 * wait / wave:5.1f} %   (GRBM_GUI_ACTIVE/8 = {clk:.0f} clk"
    per_kernel[k] = {"mfma_busy": busy / (clk * 1024), "waves_parked": wait / wave, "gui_active_clk": clk}
    if dur.get(k):
        d = sum(dur[k]) / len(dur[k])
        line += f", {d / 1e3:.1f} us under the counters => {clk / d:.2f} GHz"
        per_kernel[k].update(us_under_counters=d / 1e3, clock_ghz=clk / d)
    print(line + ")")
if len(sys.argv) > 3:
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import rnb_neus_fork_amd as R  # noqa: E402
    json.dump({"build_id": R.native.build_id(), "per_kernel": per_kernel,
               "note": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY ... over "
                       "bench.py --steps 4 --warmup 2; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"},
              open(os.path.join(root, "profiles", sys.argv[3]), "w"), indent=1)
