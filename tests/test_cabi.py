"""The C-ABI library builds, loads and exports every symbol include/qpsim_hip.h declares (no GPU needed)."""
from __future__ import annotations

import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from qpsim_amd import _hip
    return _hip.load()


def _declared_symbols() -> list[str]:
    text = (ROOT / "include" / "qpsim_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from qpsim_amd import _hip
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in qpsim_hip.h but not exported"
        assert name in _hip.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_hip.SIGNATURES) == declared


def test_version_and_error_string(lib):
    assert lib.qp_version() >= 100
    assert isinstance(lib.qp_last_error(), bytes)


def test_argument_validation_happens_before_any_launch(lib):
    import ctypes as C
    from qpsim_amd import _hip
    g = _hip.GridDesc(0, 4, 1, 0, 0, 0, 0, 0, 0, 0)
    assert lib.qp_stencil_combine(C.byref(g), 0.1, 0, 0, 0, 1.0, 0.0, 0.0, 0.0, 0.0, 0) == -1
    assert b"positive" in lib.qp_last_error()
    assert lib.qp_axpy(0, 1.0, 0, 0, 0) == -1
    t = _hip.CollisionTables(4, 7, 2, 0, 0, 0, 0, 0, 0, 0)
    assert lib.qp_collision_step(C.byref(t), 0, 10, 0, 0, 0, 0, 1.0, 0.1, 1, 1, 1, 0) == -1


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from qpsim_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setenv("QPSIM_HIP_LIBRARY", str(tmp_path / "missing.so"))
    with pytest.raises(_hip.HipLibraryMissing):
        _hip.load()


def test_product_never_imports_the_oracle():
    pkg = ROOT / "quasiparticle-physics-simulation_amd" / "qpsim_amd"
    for path in pkg.glob("*.py"):
        src = path.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f"{path.name} mentions the oracle"


def test_solver_requires_a_gpu_when_none_is_present():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    from qpsim_amd.solver import run_2d_crank_nicolson
    mask = np.ones((2, 2), dtype=bool)
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        run_2d_crank_nicolson(mask, edges, bcs, np.ones((2, 2)), 1.0, 0.1, 0.2, 1.0)
