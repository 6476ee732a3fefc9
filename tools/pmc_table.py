#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter CSVs into the per-kernel traffic table bench.py reads (profiles/*_pmc_traffic.json).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph
    python tools/pmc_table.py out/fetch out/write profiles/rNN_pmc_traffic.json

(separate passes, kernel by kernel -- counters are not collected inside graph replays.)  Keys are
"<kernel name without 'void ' and the parameter list>|grid=<total work-items>", values the mean FETCH_SIZE / WRITE_SIZE
per launch in KiB as the counters report them (FETCH_SIZE under-counts 16-byte-per-lane streaming reads by 2x on gfx950,
MI355X_MICROARCH.md; bench.py applies that correction per kernel).  Each row also carries `source`: the fingerprint of the
files that define its kernel (vqa_playground_pytorch_amd/_srchash.py), and the table `__source__.source_hash`."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqa_playground_pytorch_amd import _srchash  # noqa: E402  (no GPU, no library needed)


def load(directory, counter):
    out = {}
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0]
            if not name.startswith("vqa::") and "vqa" not in name:
                continue
            grid = row.get("Grid_Size") or row.get("Grid_Size_X") or "?"
            key = "%s|grid=%s" % (name, grid)
            # one row per (dispatch, counter[, dimension]): sum the dimensions of a dispatch, then average dispatches
            out.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
            out[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in out.items()}


def main():
    fetch_dir, write_dir, dest = sys.argv[1:4]
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    table = {}
    for key in sorted(set(fetch) | set(write)):
        n = fetch.get(key, write.get(key))[0]
        table[key] = {"launches": n, "FETCH_SIZE_KiB": fetch.get(key, (0, 0.0))[1], "WRITE_SIZE_KiB": write.get(key, (0, 0.0))[1]}
    # every row says which sources its kernel was compiled from (bench.py drops a row whose stamp differs from the tree's)
    _srchash.stamp_table(table)
    json.dump(table, open(dest, "w"), indent=1, sort_keys=True)
    print("wrote %s: %d kernels" % (dest, len(table) - 1))


if __name__ == "__main__":
    main()
