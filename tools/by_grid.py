#!/usr/bin/env python3
"""Per-dispatch durations of the hand-written kernels, grouped by (kernel, grid work-items), from a rocprofv3
--kernel-trace CSV (the --stats summary merges launches of one template instance at different grids).
    python tools/by_grid.py <dir with *_kernel_trace.csv> <steps in the run> [header text] [table.json] > profiles/<name>.txt
With a fourth argument the same rows are also written as JSON ({"<kernel>|grid=<work-items>": {launches, median_us, mean_us,
source}}, `source` = the fingerprint of the files that define the kernel): bench.py prints a row's median next to its own
event-timed duration (`trace_ms`) as long as the stamp still matches the tree."""
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_playground_pytorch_amd import _srchash  # noqa: E402  (no GPU, no library needed)


def main():
    d, steps = sys.argv[1], float(sys.argv[2])
    header = sys.argv[3] if len(sys.argv) > 3 else ""
    files = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
    if not files:
        sys.exit("no kernel_trace.csv under " + d)
    groups = {}
    for row in csv.DictReader(open(files[0])):
        name = row["Kernel_Name"]
        if not name.startswith(("vqa::", "void vqa::")):
            continue
        name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if "Grid_Size" in row:
            grid = int(row["Grid_Size"])
        else:
            grid = int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1)) * int(row.get("Grid_Size_Z", 1))
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        groups.setdefault((name, grid), []).append(dur)
    if len(sys.argv) > 4:
        table = {"%s|grid=%d" % (name, grid): {"launches": len(ds), "median_us": round(statistics.median(ds), 2),
                                               "mean_us": round(sum(ds) / len(ds), 2)}
                 for (name, grid), ds in groups.items()}
        json.dump(_srchash.stamp_table(table), open(sys.argv[4], "w"), indent=1, sort_keys=True)
    if header:
        print("# " + header)
    print("%-84s %10s %8s %10s %10s %10s" % ("kernel", "grid", "launches", "median_us", "mean_us", "us/step"))
    for (name, grid), ds in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print("%-84s %10d %8d %10.1f %10.1f %10.1f" % (name[:84], grid, len(ds), statistics.median(ds),
                                                       sum(ds) / len(ds), sum(ds) / steps))


if __name__ == "__main__":
    main()
