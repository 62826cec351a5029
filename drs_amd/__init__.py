"""Import name of the package: the directory the task prescribes is called ``dynamic-rs-segmentation_amd``, which is not a Python
identifier, so ``drs_amd`` is a package whose search path IS that directory (its modules are imported from there, one copy)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dynamic-rs-segmentation_amd")]

from . import _lib            # noqa: E402,F401
from .nets import Plan, known_net_types, resolve          # noqa: E402,F401

__all__ = ["Plan", "known_net_types", "resolve"]
