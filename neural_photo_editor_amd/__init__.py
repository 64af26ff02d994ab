"""MI355X-native compute path for the Introspective Adversarial Network of ajbrock/Neural-Photo-Editor.

Only what the hot path needs lives here: the C-ABI HIP library (csrc/, include/ian.h), its ctypes
binding (lib.py) and the host-side mirror of the reference's model facade (api.py, API.py:11-110).
"""
from .api import IAN  # noqa: F401

__all__ = ["IAN"]
