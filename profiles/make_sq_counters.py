#!/usr/bin/env python3
"""Summarise one rocprofv3 SQ-counter pass into per-kernel means for the MFMA kernels.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \\
        SQ_ACTIVE_INST_LDS --output-format csv -d OUT/pmc_sq -- python3 bench.py --serialize --no-cpu --no-roofline --steps 5 --warmup 2 --blocks 1
    python3 profiles/make_sq_counters.py OUT/pmc_sq COMMIT > profiles/r02_pmc_sq_counters.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); SQ_WAVE_CYCLES / SQ_WAIT_* are in quad-cycles, the wait fractions
are of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_gconv_up4|k_gconv|k_wgrad)<(\d+)(?:, (\d+))?", name.replace("siggan::", ""))
    if not m:
        return None
    return f"{m.group(1)}<{m.group(2)}{',' + m.group(3) if m.group(3) else ''}>"


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": __doc__.split("\n\n")[1].strip().replace("\\\n        ", "") + "  " + __doc__.split("\n\n")[2].strip().replace("\n", " "),
       "commit": sys.argv[2] if len(sys.argv) > 2 else None, "kernels": {}}
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CU_CYCLES", [0]))):
    m = {c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())}
    row = {"launches": len(next(iter(v.values())))}
    row.update(m)
    if m.get("SQ_BUSY_CU_CYCLES"):
        row["mfma_util"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * m["SQ_BUSY_CU_CYCLES"]), 4)
    if m.get("SQ_WAVE_CYCLES"):
        row["wait_any_frac"] = round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
        row["wait_inst_any_frac"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 4)
    out["kernels"][k] = row
print(json.dumps(out, indent=1))
