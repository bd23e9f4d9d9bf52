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


def _header_struct_members(name: str) -> list[str]:
    """Member names of `typedef struct <name> { ... }` in include/qpsim_hip.h, in declaration order."""
    text = (ROOT / "include" / "qpsim_hip.h").read_text()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [part.strip().lstrip("*").strip().split()[-1].lstrip("*") for part in decl.split(",")]
    return names


def _doc_fields(cls_name: str) -> list[tuple[str, str]]:
    """(member, ctypes type) pairs of the `_fields_` list of class `cls_name` in INTEGRATION.md's binding snippet."""
    text = (ROOT / "INTEGRATION.md").read_text()
    block = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n" % cls_name, text, flags=re.S).group(1)
    return re.findall(r'\("([a-z_0-9]+)",\s*C\.(c_[a-z0-9_]+)\)', block)


@pytest.mark.parametrize("cls_name,c_name", [("GridDesc", "qp_grid_desc"), ("CollisionTables", "qp_collision_tables")])
def test_documented_binding_matches_header_and_product_binding(cls_name, c_name):
    """INTEGRATION.md's reference-side ctypes structures = the header's structs = the binding the product uses."""
    import ctypes as C
    from qpsim_amd import _hip
    doc = _doc_fields(cls_name)
    ours = getattr(_hip, cls_name)._fields_
    assert [n for n, _ in doc] == _header_struct_members(c_name)
    assert [n for n, _ in doc] == [n for n, _ in ours]
    for (name, doc_type), (_, our_type) in zip(doc, ours):
        assert C.sizeof(getattr(C, doc_type)) == C.sizeof(our_type), name
    assert doc[0] == ("struct_size", "c_uint32")


def test_struct_size_mismatch_is_rejected(lib):
    """A binding built against another header revision (shorter struct) is refused before anything is read past it."""
    import ctypes as C
    from qpsim_amd import _hip
    t = _hip.CollisionTables.make(4, 7, 1, 8, 8, 8, 8, 8, 8, 0)
    t.struct_size = 64                     # what a 10-member revision of the struct would have said
    assert lib.qp_collision_step(C.byref(t), 8, 10, 8, 16, 8, 0, 1.0, 0.1, 1, 1, 1, 0) == -1
    assert b"struct_size" in lib.qp_last_error()
    g = _hip.GridDesc.make(4, 4, 1, 8, 8, 8, 8, 8, 8, 0)
    g.struct_size = 0
    assert lib.qp_stencil_combine(C.byref(g), 0.1, 8, 0, 8, 1.0, 0.0, 0.0, 0.0, 0.0, 0) == -1
    assert b"struct_size" in lib.qp_last_error()


def test_version_and_error_string(lib):
    assert lib.qp_version() >= 200
    assert isinstance(lib.qp_last_error(), bytes)


def test_argument_validation_happens_before_any_launch(lib):
    import ctypes as C
    from qpsim_amd import _hip
    g = _hip.GridDesc.make(0, 4, 1, 0, 0, 0, 0, 0, 0, 0)
    assert lib.qp_stencil_combine(C.byref(g), 0.1, 0, 0, 0, 1.0, 0.0, 0.0, 0.0, 0.0, 0) == -1
    assert b"positive" in lib.qp_last_error()
    assert lib.qp_axpy(0, 1.0, 0, 0, 0) == -1
    t = _hip.CollisionTables.make(4, 7, 2, 0, 0, 0, 0, 0, 0, 0)
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
