"""Library-GEMM solution table for the ``[B,*]``-sized layers.

The question-side projections, the per-glimpse linears, Mutan's question-side ranks and the classifier are plain fp32
GEMMs of 0.3-3 GFLOP that stay on rocBLAS / hipBLASLt (a hand-written family for them lost to the library).
The library's default heuristic picks poor tilings for several of these shapes (0.13-0.44 of peak); PyTorch's TunableOp
can time every rocBLAS / hipBLASLt solution for a shape once and remember the winner.  ``tuned_gemms_gfx950.csv`` is that
table for the BASELINE shapes (CoR2 / ODA at 512 per rank, the 100-region / 128-per-rank variant, the question encoder),
recorded on an MI355X with this image's libraries by ``tools/tune_library_gemms.sh``; the file carries the library versions
it is valid for and is ignored when they differ.

    VQA_TUNED_GEMMS=1 (default)  look shapes up in the shipped table, no tuning at run time (unknown shapes: library default)
    VQA_TUNED_GEMMS=tune         also tune unknown shapes on first use and write the table to VQA_TUNED_GEMMS_FILE
                                 (REQUIRED in this mode: the shipped table is never overwritten; with several ranks
                                 every rank writes its own file, the device ordinal inserted into the name)
    VQA_TUNED_GEMMS=0            leave the library's own heuristic alone
A PYTORCH_TUNABLEOP_ENABLED already set in the environment wins: the user is driving TunableOp themselves.
"""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemms_gfx950.csv")
_state = {"done": False, "mode": None, "loaded": None}


def enable():
    """Idempotent; called by the trainer.  -> the mode in effect ('0', '1', 'tune' or 'user').
    Note: TunableOp is a process-wide switch of torch -- once a trainer has called this, every torch GEMM of the process
    is looked up in the table (unknown shapes keep the library default), not only this model's."""
    if _state["done"]:
        return _state["mode"]
    mode = os.environ.get("VQA_TUNED_GEMMS", "1")
    if "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        mode = "user"
    elif mode not in ("0", "1", "tune"):
        raise ValueError("VQA_TUNED_GEMMS must be 0, 1 or tune, got %r" % (mode,))
    if mode == "tune" and not os.environ.get("VQA_TUNED_GEMMS_FILE"):
        raise ValueError("VQA_TUNED_GEMMS=tune needs VQA_TUNED_GEMMS_FILE=<path to write>: the shipped table (%s) is "
                         "never overwritten by a run" % TABLE)
    _state["done"] = True
    if mode in ("1", "tune") and torch.cuda.is_available():
        import torch.cuda.tunable as tn
        tn.enable(True)
        tn.tuning_enable(mode == "tune")
        if mode == "tune":
            multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
            tn.set_filename(os.environ["VQA_TUNED_GEMMS_FILE"], insert_device_ordinal=multi)   # one file per rank
        loaded = False
        if os.path.exists(TABLE):
            try:
                loaded = bool(tn.read_file(TABLE))     # False: validators (ROCm / rocBLAS / hipBLASLt / arch) differ
            except RuntimeError:
                loaded = False
        _state["loaded"] = loaded
    _state["mode"] = mode
    return mode


def describe():
    """What bench.py prints as config.library_gemms: the mode in effect AND whether the shipped table was accepted."""
    mode = enable()
    text = {"1": "rocBLAS / hipBLASLt solutions looked up in the recorded TunableOp table "
                 "(vqa_playground_pytorch_amd/tuned_gemms_gfx950.csv), no tuning at run time",
            "tune": "TunableOp, tuning unknown shapes in the warm-up steps",
            "0": "library default heuristic", "user": "TunableOp as set in the environment"}[mode]
    if mode in ("1", "tune"):
        if _state["loaded"]:
            text += "; table loaded"
        else:
            text += "; TABLE NOT LOADED (validator mismatch or file missing): every shape runs on the library default"
    return text
