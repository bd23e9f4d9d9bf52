"""qpsim_amd -- MI355X-native drop-in for the time-stepping hot path of ``qpsim``.

Host-side modules (``models``, ``geometry``, ``tables``, ``initial_conditions``, ``precompute``,
``safe_eval``) are plain NumPy and import anywhere.  ``solver`` drives the HIP library
(``libqpsim_hip.so``) on PyTorch-ROCm device tensors and raises if the library or a GPU is missing.
"""
from __future__ import annotations

__version__ = "0.1.0"

from . import geometry, initial_conditions, models, precompute, safe_eval, tables  # noqa: F401
