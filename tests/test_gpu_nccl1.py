"""The RCCL code path on hardware that has ONE GPU: a one-rank "nccl" process group (RCCL on ROCm) with the trainer's
gradient all-reduce forced (VQA_FORCE_ALLREDUCE=1).  A SUM over one rank is the identity, so every step must equal the
same step without a collective (to the run-to-run noise of the float atomics the small-batch K1 / K3 backward kernels use:
two separate processes are compared) -- what runs for real is RCCL's init on the device, the all-reduce
launched eagerly between the replayed hipGraphs (2-graph step), the asynchronous all-reduce + work handle beside the second
backward graph (3-graph step, overlap) and the thread_local capture path the trainer takes whenever a collective library's
watchdog thread is alive.  (train.py:517's nn.DataParallel replaced by one process per GPU; SURVEY 5.8, 8e.  The scaling
run over xGMI is the driver's.)"""
import pytest

from test_gpu_dp2 import _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", ["cor2", "oda"])
def test_one_rank_rccl_allreduce_is_the_identity(model):
    for graph in ("0", "1"):
        for overlap in ("0", "1"):
            env = {"G": graph, "VQA_DP_OVERLAP": overlap, "MODEL": model}
            want = _run(1, env)
            got = _run(1, dict(env, NCCL1="1"))
            tag = (model, graph, overlap)
            assert not want["reduce"] and want["backend"] is None, tag
            assert got["reduce"] and got["backend"] == "nccl", tag
            assert got["graph"] == want["graph"] == (graph == "1"), tag
            assert got["overlap"] == want["overlap"] == (overlap == "1"), tag
            if graph == "1":
                assert got["graphs"] == (["front_a", "front_b", "tail"] if overlap == "1" else ["front", "tail"]), tag
            assert got["losses"][0] == want["losses"][0], tag          # (the first forward has no atomics in it: exact)
            for a, b in zip(got["losses"], want["losses"]):
                assert abs(a - b) <= 1e-6 * abs(b), (tag, got["losses"], want["losses"])
            for a, b in zip(got["norms"], want["norms"]):
                assert abs(a - b) <= 1e-5 * abs(b), (tag, got["norms"], want["norms"])
            assert abs(got["weight_digest"] - want["weight_digest"]) <= 1e-4 * abs(want["weight_digest"]) + 1e-5, tag


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_one_rank_rccl_with_a_graph_per_input_slot(overlap):
    """bench.py's form of the multi-GPU step: rotating resident batches, one forward + backward graph per input slot (trainer
    input_slots), the all-reduce (a one-rank RCCL group, forced) between / beside the replayed graphs of whichever slot is up.
    Three resident copies of one batch visited in turn must give the numbers of the single-buffer step."""
    env = {"G": "1", "VQA_DP_OVERLAP": overlap, "MODEL": "cor2"}
    want = _run(1, env)
    got = _run(1, dict(env, NCCL1="1", SLOTS="3"))
    assert got["reduce"] and got["backend"] == "nccl" and got["graph"] and got["slots"] == 3, got
    assert got["overlap"] == (overlap == "1")
    for a, b in zip(got["losses"], want["losses"]):
        assert abs(a - b) <= 1e-6 * abs(b), (got["losses"], want["losses"])
    for a, b in zip(got["norms"], want["norms"]):
        assert abs(a - b) <= 1e-5 * abs(b), (got["norms"], want["norms"])
