"""Import alias: the package directory is named ``dynamic-rs-segmentation_amd`` (not a Python
identifier); ``import drs_amd`` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dynamic-rs-segmentation_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
