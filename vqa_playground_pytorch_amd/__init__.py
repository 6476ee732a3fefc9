"""MI355X-native CoR2 / ODA hot path of bupt-cist/vqa-playground-pytorch.

Python host (PyTorch-ROCm owns memory, streams and torch.distributed) over hand-written gfx950 HIP
kernels in ``libvqa_mi355x.so`` (C ABI: include/vqa_mi355x.h).  ``config/CoR2.py`` and
``config/ODA.py`` at the repo root expose the reference's module surface on top of this package.
"""
from . import _lib, layers, ops  # noqa: F401
from .cor2 import Model as CoR2Model  # noqa: F401
from .encoder import BayesianGRU, SkipThoughts  # noqa: F401
from .oda import Model as ODAModel  # noqa: F401

__all__ = ["CoR2Model", "ODAModel", "SkipThoughts", "BayesianGRU", "layers", "ops"]
