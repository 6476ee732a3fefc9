"""Hashes of the HIP sources: what ties a built libvqa_mi355x.so, and the committed counter tables under profiles/, to
the code they were made from.

    python _srchash.py            -> C string literal of source_hash() (csrc/Makefile writes it into build/source_hash.inc,
                                     api.hip returns it from vqa_source_hash())

source_hash()        one sha256 over every file the library is compiled from (csrc/*.hip, csrc/*.hpp, include/*.h), names and
                     contents, in sorted order -- `vqa_source_hash()` of a library built from this tree returns it;
                     __graft_entry__.smoke() and tests/test_host_cpu.py compare the two, so a stale binary on the GPU box is seen.
file_hashes()        {file name: sha256 of its contents}.
kernel_sources(k)    the .hip / .hpp files that DEFINE the __global__ kernel a profiler row names, plus every .hpp (a template
                     header change can reach any kernel; include/*.h holds declarations only and is left to source_hash()).  tools/pmc_table.py / pmc_mfma.py store `kernel_fingerprint(k)` per row;
                     bench.py recomputes it and drops a row's counters (`traffic_stale`) when the sources moved on.
This module imports nothing of the package (the Makefile runs it before the library exists)."""
import hashlib
import os
import re
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")


def source_files(csrc=CSRC, include=INCLUDE):
    files = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp"))]
    if os.path.isdir(include):
        files += [os.path.join(include, f) for f in os.listdir(include) if f.endswith(".h")]
    return sorted(files, key=os.path.basename)


def file_hashes(csrc=CSRC, include=INCLUDE):
    out = {}
    for path in source_files(csrc, include):
        with open(path, "rb") as fh:
            out[os.path.basename(path)] = hashlib.sha256(fh.read()).hexdigest()
    return out


def source_hash(csrc=CSRC, include=INCLUDE):
    h = hashlib.sha256()
    for name, digest in sorted(file_hashes(csrc, include).items()):
        h.update(("%s:%s\n" % (name, digest)).encode())
    return h.hexdigest()


_defs = {}


def _definitions(csrc=CSRC):
    """{kernel identifier: set of files with a `__global__ ... <identifier>(` definition}."""
    key = os.path.abspath(csrc)
    if key not in _defs:
        table = {}
        # `__global__ [__launch_bounds__(...)] void <identifier>(`
        # (one level of nested parentheses inside the launch bounds: `__launch_bounds__(kThreads, (TUNE & 4) ? 2 : 1)`)
        pat = re.compile(r"__global__\s+(?:__launch_bounds__\s*\((?:[^()]|\([^()]*\))*\)\s*)?(?:static\s+)?void\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(")
        for path in source_files(csrc, os.devnull):
            with open(path, "r", errors="replace") as fh:
                for m in pat.finditer(fh.read()):
                    table.setdefault(m.group(1), set()).add(os.path.basename(path))
        _defs[key] = table
    return _defs[key]


def kernel_identifier(row_key):
    """'vqa::sp::gemm_nt_kernel<9, 5, ...>|grid=65536' -> 'gemm_nt_kernel'; a mangled name (rocprofv3 leaves some template instances
    over __bf16 undemangled: '_ZN3vqa30bilinear_bwd_prep8_bf16_kernelILi2EE...') -> its last length-prefixed component."""
    name = row_key.split("|")[0].strip()
    if name.startswith("_Z"):
        rest, last = name[2:].lstrip("N"), None
        while rest and rest[0].isdigit():
            m = re.match(r"(\d+)", rest)
            n = int(m.group(1))
            last, rest = rest[m.end():m.end() + n], rest[m.end() + n:]
        return last or name
    name = name.split("<")[0].split("(")[0].strip()
    return name.split("::")[-1]


def kernel_sources(row_key, csrc=CSRC, include=INCLUDE):
    ident = kernel_identifier(row_key)
    files = set(_definitions(csrc).get(ident, ()))
    # every .hpp (templates: a header change can reach any kernel); NOT the C header of declarations -- a new entry point does not
    # change an existing kernel (it is part of source_hash(), the whole library's)
    files |= {os.path.basename(p) for p in source_files(csrc, include) if p.endswith(".hpp")}
    return sorted(files)


def kernel_fingerprint(row_key, csrc=CSRC, include=INCLUDE):
    """sha256 (16 hex digits) over the files kernel_sources() names; None when no file defines the kernel."""
    ident = kernel_identifier(row_key)
    if ident not in _definitions(csrc):
        return None
    hashes = file_hashes(csrc, include)
    h = hashlib.sha256()
    for name in kernel_sources(row_key, csrc, include):
        h.update(("%s:%s\n" % (name, hashes[name])).encode())
    return h.hexdigest()[:16]


def stamp_table(table, csrc=CSRC, include=INCLUDE):
    """Write `source` (the kernel's fingerprint) into every row of a counter table and the whole tree's hash under "__source__"."""
    for key, row in table.items():
        if isinstance(row, dict) and not key.startswith("__"):
            row["source"] = kernel_fingerprint(key, csrc, include)
    table["__source__"] = {"source_hash": source_hash(csrc, include)}
    return table


def row_is_stale(key, row, csrc=CSRC, include=INCLUDE):
    """True: the row says which sources it was measured on and they differ from the tree's (or the kernel is gone).
    A row without a stamp (tables older than round 6) is reported stale as well -- nothing ties it to the code."""
    stamp = row.get("source") if isinstance(row, dict) else None
    return stamp is None or stamp != kernel_fingerprint(key, csrc, include)


if __name__ == "__main__":
    sys.stdout.write('"%s"\n' % source_hash())
