"""Vector-pipe utilisation per kernel from a rocprofv3 --pmc pass (development / evidence tool).
usage: python tools/valu_summary.py <counter_collection.csv>
Counters: GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES.
A wave64 vector instruction occupies its SIMD's issue for 4 cycles (16 lanes) — transcendental and packed forms longer —, an
MFMA 32x32x16 for 8 of its 32; `valu issue` = 4 x (SQ_INSTS_VALU - SQ_INSTS_MFMA) / (clocks x 1024 SIMDs) is therefore a LOWER bound of
the share of cycles the vector issue port is taken by non-matrix work."""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[re.sub(r"\(.*", "", r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else 0.0
for k, v in agg.items():
    if "rnb::" not in k or mean(v.get("SQ_INSTS_MFMA", [])) == 0:
        continue
    clk = mean(v["GRBM_GUI_ACTIVE"]) / 8.0
    valu, mfma = mean(v.get("SQ_INSTS_VALU", [])), mean(v.get("SQ_INSTS_MFMA", []))
    other = valu - mfma
    print(f"{k[:62]:64s} clk {clk:9.0f}  VALU insts/wave-instr total {valu:12.0f} (MFMA {mfma:11.0f})  "
          f"non-MFMA VALU issue >= {100 * 4 * other / (clk * 1024):5.1f} %  MFMA issue {100 * 8 * mfma / (clk * 1024):5.1f} %  "
          f"ACTIVE_INST_VALU/4/clk/1024 {100 * mean(v.get('SQ_ACTIVE_INST_VALU', [0])) / 4 / (clk * 1024):6.1f} %  "
          f"LDS insts {mean(v.get('SQ_INSTS_LDS', [0])):10.0f}  VMEM wr {mean(v.get('SQ_INSTS_VMEM_WR', [0])):9.0f} rd {mean(v.get('SQ_INSTS_VMEM_RD', [0])):9.0f}")
