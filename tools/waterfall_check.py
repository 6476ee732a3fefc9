#!/usr/bin/env python3
"""Waterfall loops in the library's ISA.  When hipcc cannot prove a buffer descriptor (or any SGPR operand of a memory instruction)
wave-uniform it keeps it in VGPRs and wraps the instruction in a loop -- v_readfirstlane x 4, v_cmp_eq, s_and_saveexec, the load,
s_xor exec, s_cbranch_execnz -- once per distinct value among the lanes (one, in every case met so far: the value WAS uniform).
Twice this cost tens of microseconds per launch before it was seen (relation_dgrad_split in round 5: +50 us; the K1 -> K5 fused
forward and the encoder's batched GEMM in round 6: +15 / +35 us): a descriptor made behind a divergent region becomes a PHI of it;
an index that went through a vector-ALU division is "divergent" until it is passed through __builtin_amdgcn_readfirstlane.

    python tools/waterfall_check.py [file.hip ...]        (default: every csrc/*.hip; compiles device code only, -S)

Prints, per kernel, the number of waterfall loops (a v_readfirstlane within 12 lines before an s_cbranch_execnz that jumps backwards
over a memory instruction) and how many of them sit between the first and the last MFMA.  tests/test_host_cpu.py asserts that the
hot kernels have none."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vqa_playground_pytorch_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernels_of(asm):
    lines = asm.split("\n")
    out = {}
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):\s*;\s*@", line)
        if not m:
            continue
        j = i
        while j < len(lines) and "s_endpgm" not in lines[j]:
            j += 1
        out[m.group(1)] = lines[i:j]
    return out


def waterfalls(body):
    """(loops, loops between MFMAs): a backward s_cbranch_execnz whose loop body holds a v_readfirstlane and a memory instruction."""
    labels = {m.group(1): k for k, x in enumerate(body) for m in [re.match(r"^(\.LBB[0-9_]+):", x)] if m}
    mf = [k for k, x in enumerate(body) if "v_mfma" in x]
    lo, hi = (mf[0], mf[-1]) if mf else (0, -1)
    total = inner = 0
    for k, x in enumerate(body):
        m = re.search(r"s_cbranch_execnz\s+(\.LBB[0-9_]+)", x)
        if not m or labels.get(m.group(1), k + 1) > k:
            continue
        loop = body[labels[m.group(1)]:k]
        if len(loop) <= 24 and any("v_readfirstlane" in y for y in loop) and any(re.search(r"\b(buffer_|global_|flat_|ds_|s_load|s_buffer)", y) for y in loop):
            total += 1
            inner += lo < k < hi
    return total, inner


def check(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out, path],
                       check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(path))
        return {name: waterfalls(body) for name, body in kernels_of(open(out).read()).items()}


def demangle(names):
    try:
        res = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, res))
    except Exception:       # noqa: BLE001
        return {n: n for n in names}


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    for path in files:
        res = check(os.path.abspath(path))
        names = demangle(list(res))
        for k, (total, inner) in sorted(res.items()):
            if total:
                bad += inner > 0
                print("%-28s %-110s waterfall loops %3d, between MFMAs %3d" % (os.path.basename(path), names[k].split("(")[0][:110], total, inner))
    print("kernels with a waterfall loop between their MFMAs: %d" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
