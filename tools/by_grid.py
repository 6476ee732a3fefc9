#!/usr/bin/env python3
"""Per-dispatch durations of the hand-written kernels, grouped by (kernel, grid work-items), from a rocprofv3
--kernel-trace CSV (the --stats summary merges launches of one template instance at different grids).
    python tools/by_grid.py <dir with *_kernel_trace.csv> <steps in the run> [header text] > profiles/<name>.txt"""
import csv
import glob
import statistics
import sys


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
    if header:
        print("# " + header)
    print("%-84s %10s %8s %10s %10s %10s" % ("kernel", "grid", "launches", "median_us", "mean_us", "us/step"))
    for (name, grid), ds in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print("%-84s %10d %8d %10.1f %10.1f %10.1f" % (name[:84], grid, len(ds), statistics.median(ds),
                                                       sum(ds) / len(ds), sum(ds) / steps))


if __name__ == "__main__":
    main()
