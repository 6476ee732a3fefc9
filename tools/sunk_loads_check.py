#!/usr/bin/env python3
"""Loops that wait for the loads they have just issued:  python3 tools/sunk_loads_check.py file.s [kernel-substring ...]

For every innermost loop of every kernel in a hipcc -S listing: the position (instruction index in the loop body) of its
vector-memory and LDS loads and of the s_waitcnt that follow, flagging
  * a `s_waitcnt vmcnt(0)` / `lgkmcnt(0)` within the first tenth of the body while loads sit in the last fifth (a prefetch the
    compiler sank to the end of the pass: every pass pays the latency), and
  * loads followed by a wait for them within 8 instructions.
Round 6 found three such loops by hand (K2's weight and data gradients, the GRU's batched product); this is the same look,
scripted."""
import re
import sys


def kernels(lines):
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l):
            start = (i, l.split(":")[0])
        elif "s_endpgm" in l and start:
            yield start[1], lines[start[0]:i]
            start = None


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            out.append((labels[m.group(1)], i))
    inner = [(a, b) for a, b in out if not any(a <= c and d <= b and (c, d) != (a, b) for c, d in out)]
    return inner


def instrs(seg):
    return [x.strip() for x in seg if x.strip() and not x.strip().startswith((";", ".")) and not x.strip().endswith(":")]


def loop_stats(lines, filters=()):
    """{kernel: [loop, ...]}: per innermost loop its instruction count, the positions of its MFMAs, vector-memory loads and LDS reads
    and its (position, text) s_waitcnt -- what tests/test_host_cpu.py asserts on for the loops round 6 repaired."""
    out = {}
    for name, body in kernels(lines):
        if filters and not any(f in name for f in filters):
            continue
        for a, b in loops(body):
            ins = instrs(body[a:b + 1])
            out.setdefault(name, []).append({
                "n": len(ins),
                "mfma": [i for i, x in enumerate(ins) if x.startswith("v_mfma")],
                "vmem": [i for i, x in enumerate(ins) if x.startswith(("buffer_load", "global_load"))],
                "ds": [i for i, x in enumerate(ins) if x.startswith("ds_read")],
                "waits": [(i, x) for i, x in enumerate(ins) if x.startswith("s_waitcnt")]})
    return out


def compile_to_asm(path, hipcc="/opt/rocm/bin/hipcc"):
    """Device-only -S listing of one csrc file (lines)."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", os.path.join(root, "include"),
                        "-I", os.path.dirname(path), "-o", out, path], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    for name, body in kernels(lines):
        if filters and not any(f in name for f in filters):
            continue
        for a, b in loops(body):
            ins = instrs(body[a:b + 1])
            n = len(ins)
            if n < 24:
                continue
            vm = [i for i, x in enumerate(ins) if x.startswith(("buffer_load", "global_load"))]
            ds = [i for i, x in enumerate(ins) if x.startswith("ds_read")]
            waits = [(i, x) for i, x in enumerate(ins) if x.startswith("s_waitcnt")]
            flags = []
            for i, x in waits:
                m = re.search(r"vmcnt\((\d+)\)", x)
                if m and int(m.group(1)) == 0 and vm and i < n * 0.15 and max(vm) > n * 0.7:
                    flags.append("vmcnt(0) at %d/%d with loads at %s" % (i, n, vm[-4:]))
                m = re.search(r"lgkmcnt\((\d+)\)", x)
                if m and int(m.group(1)) == 0 and ds and i < n * 0.15 and max(ds) > n * 0.7:
                    flags.append("lgkmcnt(0) at %d/%d with ds_reads at %s" % (i, n, ds[-4:]))
                near = [j for j in vm if 0 < i - j <= 8] if "vmcnt" in x else []
                if near and re.search(r"vmcnt\(0\)", x):
                    flags.append("vmcnt(0) at %d right behind loads at %s" % (i, near))
            mf = sum(1 for x in ins if x.startswith("v_mfma"))
            if flags:
                print("%-90s loop %d..%d (%d instrs, %d mfma, %d vmem loads, %d ds reads)" % (name[:90], a, b, n, mf, len(vm), len(ds)))
                for f in flags[:4]:
                    print("      " + f)


if __name__ == "__main__":
    main()
