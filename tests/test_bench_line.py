"""bench.py's stdout contract: ONE compact JSON line under 2000 bytes that still carries `roofline` and `cpu_baseline`
(the round-3 line grew to 21 KB and the driver recorded `parsed: null`), and the (kernel, grid) look-up of the committed
counter tables."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _canned(world=1):
    entry = {"kernel": "linear_act_fwd", "shape": [18432, 2048, 310, True], "launches": 10, "mean_ms": 0.21901,
             "bound": "mfma", "achieved": 106.83, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.6791, "traffic": 196012345,
             "ms_per_step": 0.21901, "mfma_busy_pct": 77.12,
             "mfma_counters": {"SQ_BUSY_CYCLES": 1.0e7, "SQ_INSTS_MFMA": 5.0e6, "SQ_VALU_MFMA_BUSY_CYCLES": 3.0e8,
                               "launches": 8, "key": "vqa::rt::gemm_nt_kernel<9, 5, 1, 2, 2, true, vqa::EpiBiasAct, 0>|grid=65536"},
             "device_kernels": [["(rt::gemm_nt_kernel<9, 5, 1, 2, 2, true, EpiBiasAct>)", 65536]]}
    sub = {"metric": "m" * 90, "value": 462553.0, "unit": "samples/s", "ms_per_step": 1.107, "dtype": "f32", "steps": 20,
           "warmup": 5, "roofline": dict(entry, kernel="lowrank_bilinear_fusion_bwd_bf16", shapes=[[1] * 6] * 3), "step_coverage": {"timed_ops_ms": 1.0},
           "workload": "w" * 200, "launch": "hipGraph replay (2 graphs + eager all-reduce)", "command": "bench.py --model oda",
           "wall_s": 9.1}
    full = {
        "metric": "VQA samples/sec (fwd+bwd), CoR2 batch 512, 36x2048 regions", "value": 214712.3, "unit": "samples/s",
        "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 2.385, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "CoR2 fwd+bwd fp32, batch 512/GPU, 36x2048 regions + 2400-d question, 2-step chain, 2000 "
                               "answers (BASELINE configs[1]; configs[3] at 8 GPUs)", "global_batch": 512 * world,
                   "step": "s" * 120, "parallelism": "dp%d" % world, "launch": "hipGraph replay (2 graphs + eager all-reduce)",
                   "relation_mode": "factored", "library_gemms": "x" * 300,
                   "f32_products": "3xbf16 split, 6 partial products, fp32 accumulate",
                   "inputs": "4 rotating resident batches (+1 device copy/step); 1 resident batch: 254000.1 (+2.1%)",
                   "inputs_note": "n" * 400},
        "final_loss": 970.1, "final_grad_norm": 3.2,
        "roofline": dict(entry, shapes=[[18432, 2048, 310, True], [18432, 2048, 310, False]],
                         device_kernels=["vqa::rt::gemm_nt_kernel<9, 5, 1, 2, 2"]),
        "traffic_source": "t" * 200, "step_coverage": {"timed_ops_ms": 2.6, "of_ms_per_step": 1.09},
        "roofline_all": [dict(entry, kernel="op%d" % i) for i in range(70)],
        "resident_inputs": {"batches": 1, "value": 254000.1, "unit": "samples/s", "ms_per_step": 2.016, "note": "n" * 100},
        "sub_records": {tag: dict(sub) for tag, _ in bench.SUB_RECORDS},
        "cpu_baseline": {"value": 49.13, "unit": "samples/s", "cores": 16, "kind": "port", "host_cores": 256,
                         "sample": "CoR2 fwd+bwd (KLD-sum loss, dropout on), batch 16 x 31 steps after 1 warm-up; "
                                   "reference-faithful torch-CPU port; 16 threads of 256 host cores",
                         "sweep": {"16": 49.13, "32": 41.0, "64": 30.2},
                         "all_runs": [{"cores": 16, "value": 49.13, "sample": "s" * 200}] * 3},
    }
    if world > 1:       # (a multi-GPU run carries no sub-records and no CPU baseline: bench.py attaches those at N = 1 only)
        del full["sub_records"]
        full["distributed"] = {"nranks": world, "backend": "nccl", "allreduce_payload_bytes": 47760976,
                               "allreduce_ms_alone": 0.412, "allreduce_busbw_GBs": 202.9, "per_rank_samples_per_s": 200000.0,
                               "overlap": False,
                               "schedules": {"single": {"value": 1600000.0, "ms_per_step": 2.56},
                                             "overlap": {"value": 1500000.0, "ms_per_step": 2.73}}}
    if world == 1:
        full["sub_records"]["oda_b512"] = {"error": "rc=1 " + "e" * 300}
    return full


def test_compact_line_is_small_and_complete(tmp_path, capsys):
    for world in (1, 8):
        full = _canned(world)
        assert len(json.dumps(full)) > 20000                     # the record that used to be printed whole
        bench.emit(full, str(tmp_path / "bench_detail.json"))
        out, err = capsys.readouterr()
        lines = [ln for ln in out.splitlines() if ln.strip()]
        assert len(lines) == 1 and len(lines[0]) < bench.COMPACT_LIMIT
        line = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in line, key
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "mean_ms", "launches",
                    "mfma_busy_pct")) <= set(line["roofline"])
        assert set(("value", "unit", "cores", "kind", "sample", "host_cores")) <= set(line["cpu_baseline"])
        assert line["config"]["workload"].startswith("CoR2 fwd+bwd fp32") and "relation_mode" in line["config"]
        assert line["resident_inputs"] == {"value": 254000.1, "ms_per_step": 2.016, "batches": 1}
        assert line["config"]["f32_products"].startswith("3xbf16 split") and line["cpu_baseline"]["sweep"]["32"] == 41.0
        if world == 1:
            assert set(line["sub_records"]) == {tag for tag, _ in bench.SUB_RECORDS}
            assert set(line["sub_records"]["cor2_bf16_n100_b128"]) == {"value", "ms_per_step", "kernel", "frac"}
            assert "error" in line["sub_records"]["oda_b512"]
            assert len(lines[0]) < bench.COMPACT_LIMIT - 60          # headroom for longer numbers and kernel names
        assert ("distributed" in line) == (world > 1)
        if world > 1:
            assert set(line["distributed"]["schedules"]) == {"single", "overlap"}
        # the full record went to the detail file and to stderr
        detail = json.load(open(tmp_path / "bench_detail.json"))
        assert len(detail["roofline_all"]) == 70 and detail["detail_file"] == "bench_detail.json"
        assert json.loads(err.strip().splitlines()[-1])["roofline_all"]


def test_compact_line_sheds_optional_parts_rather_than_overflowing():
    full = _canned(8)
    full["sub_records"] = {"tag_%03d" % i: {"value": 1.0, "ms_per_step": 1.0, "dtype": "f32",
                                            "roofline": {"kernel": "k" * 40, "frac": 0.5}} for i in range(40)}
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.COMPACT_LIMIT
    assert "roofline" in line and "cpu_baseline" in line and "sub_records" not in line


def test_pmc_rows_are_matched_by_kernel_and_grid():
    table = {
        "vqa::relation_apply_fwd_kernel<float, 256>|grid=262144": {"FETCH_SIZE_KiB": 12315.0, "WRITE_SIZE_KiB": 16384.0, "launches": 8},
        "vqa::relation_apply_fwd_kernel<float, 256>|grid=524288": {"FETCH_SIZE_KiB": 78073.0, "WRITE_SIZE_KiB": 147456.0, "launches": 8},
        "vqa::rt::gemm_tn_kernel<5, 2, false, false>|grid=65536": {"FETCH_SIZE_KiB": 84976.0, "WRITE_SIZE_KiB": 39699.5, "launches": 8},
        "vqa::rt::gemm_tn_kernel<5, 2, true, true>|grid=65536": {"FETCH_SIZE_KiB": 96178.0, "WRITE_SIZE_KiB": 39699.5, "launches": 8},
        "vqa::attention_pool_bwd_fused_kernel<float, 256, 4, 2>|grid=131072": {"FETCH_SIZE_KiB": 83178.0, "WRITE_SIZE_KiB": 288.0, "launches": 16},
    }
    big = bench.pmc_row(table, "relation_apply_fwd", ((524288, "(relation_apply_fwd_kernel<T, 256>)"),))
    small = bench.pmc_row(table, "relation_apply_fwd", ((262144, "(relation_apply_fwd_kernel<T, 256>)"),))
    assert big["WRITE_SIZE_KiB"] == 147456.0 and small["WRITE_SIZE_KiB"] == 16384.0
    assert bench.pmc_row(table, "relation_apply_fwd", ((1024, "x"),)) is None            # a grid the table never saw
    assert bench.pmc_row(table, "relation_apply_fwd", ()) is None                           # no launch log: no guess
    gated = bench.pmc_row(table, "linear_act_bwd", ((65536, "(rt::gemm_tn_kernel<5, 2, true, true>)"), (317440, "linear_dw_reduce_kernel")))
    plain = bench.pmc_row(table, "linear_act_bwd", ((65536, "(rt::gemm_tn_kernel<5, 2, false, false>)"), (317440, "linear_dw_reduce_kernel")))
    assert gated["FETCH_SIZE_KiB"] == 96178.0 and plain["FETCH_SIZE_KiB"] == 84976.0
    # K3's backward is the fused kernel at B = 512 (the stream kernel's row is simply absent)
    fused = bench.pmc_row(table, "softmax_attention_pool_drop_bwd", ((131072, "(attention_pool_bwd_fused_kernel<T, 256, 4, 2>)"),))
    assert fused["key"].startswith("vqa::attention_pool_bwd_fused_kernel")


def test_counter_rows_are_tied_to_the_kernel_sources(tmp_path):
    """a committed counter row is evidence only for the code it was measured on.  tools/pmc_table.py stamps
    every row with the fingerprint of the files that define its kernel; edit one byte of that kernel's source (in a temp copy of
    csrc/) and the row reads stale, while rows of kernels defined elsewhere stay valid; bench.py then reports `traffic_stale`
    and prints neither `traffic` nor `mfma_busy_pct`."""
    import shutil

    from vqa_playground_pytorch_amd import _srchash

    csrc = tmp_path / "csrc"
    shutil.copytree(_srchash.CSRC, csrc, ignore=shutil.ignore_patterns("build"))
    csrc, inc = str(csrc), _srchash.INCLUDE
    keys = ["vqa::relation_apply_fwd_kernel<float, 256>|grid=524288", "vqa::adam_kernel|grid=2985216",
            "vqa::sp::gemm_nt_kernel<9, 5, 1, 2, 2, true, vqa::SplitEpiBiasAct, 0, 3, false>|grid=65536"]
    table = _srchash.stamp_table({k: {"FETCH_SIZE_KiB": 1.0, "WRITE_SIZE_KiB": 2.0, "launches": 8} for k in keys}, csrc, inc)
    assert table["__source__"]["source_hash"] == _srchash.source_hash(csrc, inc) == _srchash.source_hash()
    assert all(len(table[k]["source"]) == 16 for k in keys)
    assert "pairwise_relation.hip" in _srchash.kernel_sources(keys[0], csrc, inc)
    assert "gemm_f32_split.hpp" in _srchash.kernel_sources(keys[2], csrc, inc)
    assert not any(_srchash.row_is_stale(k, table[k], csrc, inc) for k in keys)
    # one byte of the kernel's source file
    path = os.path.join(csrc, "pairwise_relation.hip")
    data = bytearray(open(path, "rb").read())
    at = data.index(b"relation_apply_fwd_kernel")
    data[at - 1:at - 1] = b" "
    open(path, "wb").write(bytes(data))
    _srchash._defs.clear()
    assert _srchash.row_is_stale(keys[0], table[keys[0]], csrc, inc)
    assert not _srchash.row_is_stale(keys[1], table[keys[1]], csrc, inc)          # optimizer.hip did not change
    assert _srchash.source_hash(csrc, inc) != table["__source__"]["source_hash"]
    # a template header reaches every kernel
    with open(os.path.join(csrc, "common.hpp"), "a") as fh:
        fh.write("\n")
    assert all(_srchash.row_is_stale(k, table[k], csrc, inc) for k in keys)
    _srchash._defs.clear()
    # every kernel a profiler row can name is found in the sources: mangled names (rocprofv3 leaves some __bf16 instances so) and
    # kernels whose launch bounds hold a parenthesised expression included
    assert _srchash.kernel_identifier("_ZN3vqa30bilinear_bwd_prep8_bf16_kernelILi2EEEvPKDF16bS2_PKfPDF16bPfS6_iii|grid=131072") == "bilinear_bwd_prep8_bf16_kernel"
    assert _srchash.kernel_identifier("_ZN3vqa12_GLOBAL__N_121relation_dgrad_kernelILi0EEEvNS0_12RelDgradArgsENS_7DropCfgE|grid=1") == "relation_dgrad_kernel"
    for k in ("vqa::relation_dgrad_kernel<4>|grid=262144", "_ZN3vqa17column_sum_kernelIDF16bLi2EEEvPKT_iPfiii|grid=131328"):
        assert _srchash.kernel_fingerprint(k) is not None, k
    # rows without a stamp (tables older than round 6) and rows of kernels that no longer exist are stale too
    assert _srchash.row_is_stale(keys[1], {"FETCH_SIZE_KiB": 1.0})
    assert _srchash.row_is_stale("vqa::no_such_kernel|grid=64", {"source": "0" * 16})

    # bench.py: a stale row is reported, not printed
    fresh = _srchash.stamp_table({keys[0]: {"FETCH_SIZE_KiB": 10.0, "WRITE_SIZE_KiB": 20.0, "launches": 8}})
    grids = ((524288, "(relation_apply_fwd_kernel<T, 256>)"),)
    try:
        bench._tables[("traffic", "")] = fresh
        bench._tables[("mfma", "")] = {}
        bench._tables[("trace", "")] = _srchash.stamp_table({keys[0]: {"launches": 64, "median_us": 48.6, "mean_us": 48.8}})
        ok = bench.roofline_entry("relation_apply_fwd", (512, 36, 2048, True), 10, 0.05, 512, grids=grids)
        assert ok["traffic"] == int((10.0 * 2 + 20.0) * 1024) and "traffic_stale" not in ok and ok["trace_ms"] == 0.0486
        bench._tables[("traffic", "")] = {keys[0]: dict(fresh[keys[0]], source="0" * 16)}
        bench._tables[("trace", "")] = {keys[0]: {"launches": 64, "median_us": 48.6, "source": "0" * 16}}
        old = bench.roofline_entry("relation_apply_fwd", (512, 36, 2048, True), 10, 0.05, 512, grids=grids)
        assert old["traffic"] is None and old["traffic_stale"] is True and "trace_ms" not in old
        assert bench.compact_line({"roofline": old, "config": {}})["roofline"]["traffic_stale"] is True
    finally:
        bench._tables.clear()


def test_committed_counter_tables_describe_this_tree():
    """The tables bench.py reads (profiles/<EVIDENCE_TAG>_pmc_*.json, _trace*.json) were collected on the kernels of THIS tree:
    no row is stale.  After a kernel edit and before the next tools/refresh_evidence.sh run the rows of the edited kernels ARE
    stale -- bench.py then says `traffic_stale` instead of printing them, and this test SKIPS with their number (a reminder,
    not a failure: the tree is allowed to be ahead of its evidence, it is not allowed to print old counters as new)."""
    import glob

    import pytest

    from vqa_playground_pytorch_amd import _srchash

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", bench.EVIDENCE_TAG + "_pmc_*.json"))
                   + glob.glob(os.path.join(ROOT, "profiles", bench.EVIDENCE_TAG + "_trace*.json")))
    if not paths:
        pytest.skip("no counter tables committed for %s yet" % bench.EVIDENCE_TAG)
    report = []
    for path in paths:
        table = json.load(open(path))
        assert "__source__" in table, path + " carries no source stamp"
        stale = [k for k, row in table.items() if not k.startswith("__") and _srchash.row_is_stale(k, row)]
        if stale:
            report.append("%s: %d of %d rows stale" % (os.path.basename(path), len(stale), len(table) - 1))
    if report:
        pytest.skip("counter tables older than the kernels they name (bench.py reports traffic_stale): " + "; ".join(report))
