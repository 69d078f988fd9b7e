"""seghiero_amd -- MI355X-native (gfx950) training hot path of SegHiero behind the reference's Python API.

The package holds only what the hot path needs: ``csrc/`` (HIP kernels + the C ABI of
``include/seghiero_hip.h``), ``ops`` (ctypes plumbing for torch tensors) and the host-side mirrors of the
reference interface (``ResNetBackbone``, ``DepthwiseSeparableASPPContrastHead``, ``HieraTripletLoss``,
``build_*`` helpers, the train step).  There is no CPU / PyTorch-op fallback: without
``libseghiero_hip.so`` and a GPU the compute entry points raise.
"""
from ._lib import LIB, SegHieroHipError  # noqa: F401
from .hierarchy import build_fine_to_coarse_map, build_fine_to_super_map, build_hiera_index  # noqa: F401


def __getattr__(name):
    # heavy modules are imported lazily so `import seghiero_amd` stays cheap on CPU-only boxes
    import importlib
    table = {
        "ResNetBackbone": "backbone", "DepthwiseSeparableASPPContrastHead": "head", "AuxHead": "head",
        "HieraTripletLoss": "loss", "TreeTripletLoss": "loss", "CrossEntropyLoss": "loss",
        "RMIHieraTripletLoss": "loss", "RMITreeTripletLoss": "loss",
        "SegHieroTrainer": "train_step", "FusedSGD": "sgd",
    }
    if name in table:
        return getattr(importlib.import_module("." + table[name], __name__), name)
    raise AttributeError(name)
