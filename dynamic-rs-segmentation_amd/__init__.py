"""MI355X-native implementation of the dilated-CNN multi-size patch training / sliding-window
inference path of keillernogueira/dynamic-rs-segmentation (import name: ``drs_amd``).

Only what that path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/drs.h) and the
host-side mirror of the reference's interface (net builders, step loops, CLI).
"""
from . import _lib            # noqa: F401
from .nets import Plan, known_net_types, resolve          # noqa: F401

__all__ = ["Plan", "known_net_types", "resolve"]
