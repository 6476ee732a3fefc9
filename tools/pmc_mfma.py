#!/usr/bin/env python3
"""Matrix-pipe occupancy per kernel from a rocprofv3 --pmc pass (profiles/*_pmc_mfma.json, read by bench.py).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 \\
              SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d out/mfma -- \\
              python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-rotate
    python tools/pmc_mfma.py out/mfma profiles/rNN_pmc_mfma.json

Counter units on gfx950 (calibrated on tools/rt_probe.hip, whose forward kernel issues exactly 1024 waves x 11520
v_mfma_f32_16x16x4_f32): SQ_INSTS_MFMA = instructions; SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over
all SIMDs (= 32 x SQ_INSTS_MFMA for that instruction); SQ_BUSY_CU_CYCLES = busy cycles summed over the CUs;
SQ_BUSY_CYCLES = the same per shader engine (32 SEs, 8 CUs = 32 SIMDs each); SQ_WAVE_CYCLES = wave lifetimes in units
of 4 cycles.  Derived:

    mfma_busy_pct = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES)
                  = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (32 SIMDs per SE * SQ_BUSY_CYCLES)      (SURVEY 5.1's ratio, normalised)

i.e. the share of the kernel's duration, launch to last wave, in which a SIMD's matrix pipe was executing an MFMA,
averaged over the SIMDs of the busy CUs.  With SQ_INSTS_VALU in the pass (the K2 table): valu_issue_pct_min = 100 * 4 cycles *
SQ_INSTS_VALU / (4 SIMDs * SQ_BUSY_CU_CYCLES).  Keys are "<kernel name>|grid=<work-items>"."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_playground_pytorch_amd import _srchash  # noqa: E402  (no GPU, no library needed)


def main():
    dirs, dest = sys.argv[1:-1], sys.argv[-1]
    per = {}
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
                if "vqa" not in name:
                    continue
                key = "%s|grid=%s" % (name, row.get("Grid_Size", "?"))
                disp = per.setdefault(key, {}).setdefault(row["Dispatch_Id"], {})
                disp[row["Counter_Name"]] = disp.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    table = {}
    for key, dispatches in sorted(per.items()):
        names = sorted({c for d in dispatches.values() for c in d})
        entry = {"launches": len(dispatches)}
        for c in names:
            vals = [d[c] for d in dispatches.values() if c in d]
            entry[c] = sum(vals) / len(vals)
        busy, cu = entry.get("SQ_VALU_MFMA_BUSY_CYCLES"), entry.get("SQ_BUSY_CU_CYCLES")
        if busy is not None and cu:
            entry["mfma_busy_pct"] = round(100.0 * busy / (4.0 * cu), 2)
        if busy is not None and entry.get("SQ_BUSY_CYCLES"):
            entry["mfma_busy_over_sq_busy_per_simd"] = round(busy / (32.0 * entry["SQ_BUSY_CYCLES"]), 4)
        insts = entry.get("SQ_INSTS_VALU")
        if insts is not None and cu:
            # vector-ALU issue: a wave64 VALU instruction holds its SIMD's issue port for >= 4 cycles (quarter-rate ones --
            # v_mul_lo_u32, the hash multiplies -- longer), so this is a LOWER bound of the share of a busy CU's SIMD-cycles
            # spent issuing VALU (SQ_INSTS_VALU counts the MFMAs too)
            entry["valu_issue_pct_min"] = round(100.0 * 4.0 * insts / (4.0 * cu), 2)
        if (entry.get("SQ_INSTS_MFMA", 0) > 0 or "mfma" in key or "gemm" in key or "bilinear" in key or "linear" in key
                or "oda_" in key):
            table[key] = entry
    _srchash.stamp_table(table)     # per-row source fingerprints: bench.py drops a row measured on other sources
    json.dump(table, open(dest, "w"), indent=1, sort_keys=True)
    print("wrote %s: %d kernels" % (dest, len(table) - 1))
    for k, v in table.items():
        if k.startswith("__"):
            continue
        if v.get("mfma_busy_pct") or v.get("valu_issue_pct_min"):
            print("  %-110s launches %4d  mfma busy %5.1f %%  valu issue >= %5.1f %%"
                  % (k[:110], v["launches"], v.get("mfma_busy_pct") or 0.0, v.get("valu_issue_pct_min") or 0.0))


if __name__ == "__main__":
    main()
