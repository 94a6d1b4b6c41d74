"""active_gym — MI355X-native drop-in for the Atari active-vision path of
elicassion/active-gym (reference active_gym/__init__.py:3-9,58-64 export the
same names).  The observation pipeline runs as hand-written HIP kernels in
libagx.so; there is no CPU fallback."""
from . import _native  # noqa: F401  (ctypes binding; loading the .so is deferred to first use)
from .pipeline import ObsPipeline  # noqa: F401

__version__ = "0.1.0"
