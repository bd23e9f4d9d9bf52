"""Host-side geometry compilation (mask/edges/BCs -> per-cell operator tables) vs the oracle's assembly."""
from __future__ import annotations

import json

import numpy as np
import pytest

from golden_utils import GOLDEN, bcs_from_json, edges_from_json
from oracle import qp_oracle as O
from qpsim_amd.engine import BoundaryAssignmentError, compile_geometry
from qpsim_amd.geometry import extract_edge_segments
from qpsim_amd.models import BoundaryCondition


def _case():
    z = np.load(GOLDEN / "operators.npz", allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    return z, meta, edges_from_json(meta["edges"]), bcs_from_json(meta["edge_conditions"])


def test_compiled_tables_match_oracle_grid_ops():
    z, meta, edges, bcs = _case()
    g = compile_geometry(z["mask"], edges, bcs, meta["dx"])
    ops = O.build_grid_ops(z["mask"], edges, bcs, meta["dx"])
    assert np.array_equal(g.ex, ops.e_x) and np.array_equal(g.ey, ops.e_y)
    assert np.array_equal(g.sx, ops.s_x) and np.array_equal(g.sy, ops.s_y)
    assert np.array_equal((g.flags & 1) > 0, ops.link_xm) and np.array_equal((g.flags & 2) > 0, ops.link_xp)
    assert np.array_equal((g.flags & 4) > 0, ops.link_ym) and np.array_equal((g.flags & 8) > 0, ops.link_yp)
    assert np.array_equal((g.flags & 16) > 0, z["mask"])


def test_missing_boundary_conditions_raise_like_the_reference():
    z, meta, edges, bcs = _case()
    partial = dict(bcs)
    partial.pop(edges[0].edge_id)
    with pytest.raises(BoundaryAssignmentError, match="Missing: 1"):
        compile_geometry(z["mask"], edges, partial, 1.0)
    with pytest.raises(BoundaryAssignmentError, match=r"Missing boundary condition for face at cell \(0, 0\) direction 'up'"):
        compile_geometry(z["mask"], [e for e in edges if e.normal != "up"], bcs, 1.0)
    with pytest.raises(ValueError):
        compile_geometry(z["mask"], edges, {**bcs, edges[0].edge_id: BoundaryCondition("dirichlet")}, 1.0)
    with pytest.raises(ValueError):
        compile_geometry(np.zeros((3, 3), dtype=bool), [], {}, 1.0)
    with pytest.raises(ValueError):
        compile_geometry(z["mask"], edges, bcs, 0.0)


def test_large_mask_compiles_quickly():
    import time
    mask = np.ones((1024, 1024), dtype=bool)
    mask[300:500, 400:700] = False
    t0 = time.time()
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("absorbing") for e in edges}
    g = compile_geometry(mask, edges, bcs, 1.0)
    assert time.time() - t0 < 20.0
    assert g.ex[0, 0] == 2.0 and g.ey[0, 0] == 2.0 and g.ex[5, 5] == 0.0
    assert g.ex[400, 399] == 2.0 and g.ey[299, 500] == 2.0
