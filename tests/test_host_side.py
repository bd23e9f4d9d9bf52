"""Host-side input builders vs golden vectors from the reference (CPU only)."""
from __future__ import annotations

import json

import numpy as np
import pytest

from golden_utils import GOLDEN, edges_from_json
from qpsim_amd import initial_conditions as ic
from qpsim_amd import precompute, tables
from qpsim_amd.geometry import create_intrinsic_geometry, extract_edge_segments
from qpsim_amd.models import (
    BoundaryCondition,
    ExternalGenerationSpec,
    InitialConditionSpec,
    SimulationParameters,
    normalize_collision_solver_name,
)
from qpsim_amd.safe_eval import compile_safe_expression


def _z(name):
    z = np.load(GOLDEN / name, allow_pickle=False)
    return z, json.loads(str(z["meta_json"]))


def test_tables_match_reference():
    z, meta = _z("tables.npz")
    for tag, m in meta.items():
        E, dE = tables.build_energy_grid(m["gap"], m["fmin"], m["fmax"], m["ne"])
        assert np.array_equal(E, z[f"{tag}_E"]) and dE == float(z[f"{tag}_dE"])
        assert np.array_equal(tables.dynes_density_of_states(E, m["gap"], m["gamma"]), z[f"{tag}_rho"])
        assert np.array_equal(tables.bcs_density_of_states(E, m["gap"]), z[f"{tag}_bcs"])
        assert np.array_equal(tables.thermal_qp_weights(E, m["gap"], m["T_b"], m["gamma"]), z[f"{tag}_qp_weights"])
        assert np.array_equal(tables.recombination_kernel_base(E, m["gap"], m["tau_r"], m["T_c"]), z[f"{tag}_Kr0"])
        assert np.array_equal(tables.scattering_kernel_base(E, m["gap"], m["tau_s"], m["T_c"]), z[f"{tag}_Ks0"])
        assert np.array_equal(tables.recombination_kernel(E, m["gap"], m["tau_r"], m["T_c"], m["T_b"]), z[f"{tag}_Kr"])
        assert np.array_equal(tables.scattering_kernel(E, m["gap"], m["tau_s"], m["T_c"], m["T_b"]), z[f"{tag}_Ks"])
        om, idx_d, idx_s, sg = tables.build_phonon_frequency_map(E)
        assert np.array_equal(om, z[f"{tag}_omega"])
        assert np.array_equal(idx_d, z[f"{tag}_idx_diff"]) and np.array_equal(idx_s, z[f"{tag}_idx_sum"])
        assert np.array_equal(sg, z[f"{tag}_sign"])
        assert np.array_equal(tables.thermal_phonon_occupation(om, m["T_b"]), z[f"{tag}_nph"])
        assert np.array_equal(tables.integration_widths_from_centers(om, fallback_width=dE), z[f"{tag}_widths"])
    E1, dE1 = tables.build_energy_grid(180.0, 1.5, 1.5, 1)
    assert np.array_equal(E1, z["single_E"]) and dE1 == float(z["single_dE"])
    assert np.array_equal(tables.integration_widths_from_centers(np.array([3.0]), fallback_width=0.7), z["widths_single"])


def test_table_argument_errors():
    with pytest.raises(ValueError):
        tables.build_energy_grid(0.0, 1, 2, 4)
    with pytest.raises(ValueError):
        tables.build_energy_grid(1.0, 2, 2, 4)
    with pytest.raises(ValueError):
        tables.integration_widths_from_centers(np.array([1.0, 1.0]))
    with pytest.raises(ValueError):
        tables.thermal_phonon_occupation(np.array([-1.0]), 0.1)


def test_edge_segments_match_reference_ids_and_faces():
    z, meta = _z("geometry_edges.npz")
    for name, ref_edges in meta.items():
        mine = extract_edge_segments(z[f"{name}_mask"])
        ref = edges_from_json(ref_edges)
        assert len(mine) == len(ref), name
        for a, b in zip(mine, ref):
            assert (a.edge_id, a.normal, a.x0, a.y0, a.x1, a.y1) == (b.edge_id, b.normal, b.x0, b.y0, b.x1, b.y1), name
            assert [(f.row, f.col, f.direction) for f in a.faces] == [(f.row, f.col, f.direction) for f in b.faces]
    assert np.array_equal(np.asarray(create_intrinsic_geometry().mask, dtype=bool), z["intrinsic_64x120_mask"])


def test_models_validation_rules():
    assert normalize_collision_solver_name(" Fischer_Catelani_Local ") == "fischer_catelani_local"
    with pytest.raises(ValueError):
        normalize_collision_solver_name("boltzphlow_relaxation")
    with pytest.raises(ValueError):
        BoundaryCondition("dirichlet").validate()
    with pytest.raises(ValueError):
        BoundaryCondition("periodic").validate()
    BoundaryCondition(" Reflective ").validate()
    p = SimulationParameters(diffusion_coefficient=6.0, dt=0.1, total_time=1.0, mesh_size=1.0, tau_0=300.0)
    assert (p.tau_s, p.tau_r, p.tau_0) == (300.0, 300.0, 300.0)
    p = SimulationParameters(diffusion_coefficient=6.0, dt=0.1, total_time=1.0, mesh_size=1.0, tau_s=250.0, tau_r=900.0)
    assert p.tau_0 == 575.0
    with pytest.raises(ValueError):
        SimulationParameters(diffusion_coefficient=6.0, dt=0.1, total_time=1.0, mesh_size=1.0,
                             external_generation=ExternalGenerationSpec(mode="constant", rate=-1.0))
    with pytest.raises(ValueError):
        SimulationParameters(diffusion_coefficient=6.0, dt=0.1, total_time=1.0, mesh_size=1.0, energy_gap=180.0,
                             num_energy_bins=1)


def test_safe_eval_accepts_and_rejects():
    f = compile_safe_expression("return np.where(x > 0.5, params['a'], params.get('b', 2.0)) + math.pi * 0",
                                variable_names=("x", "params"))
    assert np.array_equal(f(x=np.array([0.2, 0.8]), params={"a": 1.0}), np.array([2.0, 1.0]))
    assert compile_safe_expression("", variable_names=())() == 0.0
    for bad in ("__import__('os')", "x.__class__", "np.linalg.inv(x)", "open('f')", "[i for i in x]", "lambda: 1",
                "x.sum()", "np.load('a')", "math[0]", "f(**params)", "x = 1", "np.random.rand(3)"):
        with pytest.raises(ValueError):
            compile_safe_expression(bad, variable_names=("x", "params"))
    with pytest.raises(ValueError):
        compile_safe_expression("x + 1", variable_names=("x",))()


def test_precompute_matches_reference():
    z, meta = _z("host_side.npz")
    mask = z["pre_mask"]
    edges = edges_from_json(meta["pre_edges"])
    bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    for tag in ("uni", "non"):
        d = dict(meta[f"pre_{tag}"])
        d["external_generation"] = ExternalGenerationSpec(**d["external_generation"])
        p = SimulationParameters(**d)
        for kern in (0, 1):
            pre = precompute.precompute_arrays(mask, edges, bcs, p, include_collision_kernels=bool(kern))
            keys = {k.split("__", 1)[1] for k in z.files if k.startswith(f"pre_{tag}_{kern}__")}
            assert set(pre.keys()) == keys
            for k in keys:
                ref = z[f"pre_{tag}_{kern}__{k}"]
                assert np.allclose(np.asarray(pre[k], dtype=float), ref.astype(float), rtol=1e-14, atol=0), (tag, kern, k)
            assert precompute.validate_precomputed(pre, p, mask) is None
            broken = dict(pre)
            broken.pop("D_array")
            assert "D_array" in precompute.validate_precomputed(broken, p, mask)
        p2 = SimulationParameters(**{**d, "diffusion_coefficient": 7.0})
        assert "diffusion_coefficient" in precompute.validate_precomputed(pre, p2, mask)
    assert precompute._mask_hash(mask) == float(z["mask_hash"])
    assert precompute._gap_expression_hash("return 180 + 20 * x") == float(z["gap_expr_hash"])
    with pytest.raises(ValueError):
        precompute.precompute_arrays(mask, edges, bcs, SimulationParameters(
            diffusion_coefficient=6.0, dt=0.1, total_time=0.1, mesh_size=1.0, energy_gap=180.0,
            energy_max_factor=3.0, num_energy_bins=8, gap_expression="np.nan"))


def test_initial_condition_builders_match_reference():
    z, meta = _z("host_side.npz")
    mask, E, om = z["ic_mask"], z["ic_E"], z["ic_omega"]
    for nm in ("gauss", "uniform", "point_in", "point_hole", "custom", "default"):
        got = ic.build_initial_field(mask, InitialConditionSpec(**meta[f"ic_{nm}"]))
        assert np.allclose(got, z[f"ic_field_{nm}"], rtol=1e-15, atol=0), nm
    for nm in ("fd", "fd_default_T", "uni", "cust"):
        got = ic.build_initial_energy_weights(E, 180.0, 0.1, InitialConditionSpec(**meta[f"icw_{nm}"]), 0.2)
        assert np.allclose(got, z[f"ic_ew_{nm}"], rtol=1e-15, atol=0), nm
    assert ic.build_initial_energy_weights(E, 180.0, 0.1, InitialConditionSpec(), 0.2) is None
    for nm in ("be", "be_bath", "uni", "full"):
        got = ic.build_initial_phonon_energy_state(mask, om, InitialConditionSpec(**meta[f"icp_{nm}"]), 0.15)
        assert np.allclose(got, z[f"ic_ph_{nm}"], rtol=1e-15, atol=0), nm
    got = ic.build_initial_qp_energy_state(mask, E, InitialConditionSpec(**meta["icq_full"]))
    assert np.allclose(got, z["ic_qp_full"], rtol=1e-15, atol=0)
    assert ic.build_initial_qp_energy_state(mask, E, InitialConditionSpec()) is None
    assert np.allclose(ic.evaluate_gap_expression("return 180 + 20 * x - 3 * y", mask, 180.0), z["gap_values_expr"], rtol=1e-15)
    assert np.array_equal(ic.evaluate_gap_expression("", mask, 180.0), z["gap_values_default"])
    with pytest.raises(ValueError):
        ic.evaluate_gap_expression("return x - 0.5", mask, 180.0)


def test_host_operator_assembly_matches_reference_matrices():
    from golden_utils import bcs_from_json
    from qpsim_amd.solver import _mask_to_index, build_laplacian_with_boundaries, build_variable_diffusion_laplacian
    z, meta = _z("operators.npz")
    edges, bcs = edges_from_json(meta["edges"]), bcs_from_json(meta["edge_conditions"])
    L, src, index_map = build_laplacian_with_boundaries(z["mask"], edges, bcs, meta["dx"])
    assert np.allclose(L.toarray(), z["L"], rtol=1e-14, atol=1e-14) and np.allclose(src, z["source"], rtol=1e-14, atol=1e-14)
    assert np.array_equal(index_map, z["index_map"]) and np.array_equal(_mask_to_index(z["mask"])[0], z["index_map"])
    LD, srcD = build_variable_diffusion_laplacian(z["mask"], edges, bcs, meta["dx"], z["D_spatial"])
    assert np.allclose(LD.toarray(), z["L_D"], rtol=1e-14, atol=1e-14) and np.allclose(srcD, z["source_D"], rtol=1e-14, atol=1e-14)


def test_canonicalize_initial_condition_fills_defaults():
    c = ic.canonicalize_initial_condition(InitialConditionSpec())
    assert (c.spatial_kind, c.energy_kind, c.phonon_spatial_kind, c.phonon_energy_kind) == ("gaussian", "dos", "uniform", "bose_einstein")
    assert c.spatial_params == {"amplitude": 1.0, "x0": 0.5, "y0": 0.5, "sigma": 0.12} and c.phonon_spatial_params == {"value": 1.0}
    d = ic.canonicalize_initial_condition(InitialConditionSpec(spatial_kind=" Uniform ", spatial_params={"value": 2.0}))
    assert d.spatial_kind == "uniform" and d.spatial_params == {"value": 2.0}


def test_merged_bin_tagging_pairs_each_shared_bin_once():
    """Slot tags of merged phonon bins (register collision kernels): every bin shared by a diagonal and an anti-diagonal
    gets the same tag on both entries, tags are 1..n, untagged entries keep their bin."""
    from qpsim_amd import tables as T
    from qpsim_amd.engine import structured_bin_maps, tag_merged_bins
    for ne, fmax in ((12, 5.0), (18, 10.0), (50, 5.0), (12, 3.0)):
        E, _ = T.build_energy_grid(180.0, 1.0, fmax, ne)
        om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
        diag, anti = structured_bin_maps(idx_d, idx_s, sg, allow_shared=True)
        td, ta, n = tag_merged_bins(diag, anti)
        shared = sorted(set(diag[1:].tolist()) & set(anti.tolist()))
        assert n == len(shared) and (n == 0) == (structured_bin_maps(idx_d, idx_s, sg) is not None)
        assert np.array_equal(td & 0xffff, diag) and np.array_equal(ta & 0xffff, anti)
        tags_d = {int(v & 0xffff): int(v >> 16) for v in td[1:] if v >> 16}
        tags_a = {int(v & 0xffff): int(v >> 16) for v in ta if v >> 16}
        assert tags_d == tags_a and sorted(tags_d) == shared and sorted(tags_d.values()) == list(range(1, n + 1))
        assert td[0] >> 16 == 0


def test_peaceman_rachford_parameters_meet_their_bound_on_a_commuting_model_problem():
    """`engine.peaceman_rachford_parameters`: the cycle's worst-case factor on [alpha, beta] is what it says, and a cycle
    applied to an actual commuting pair H, V (1-D Laplacians with mixed boundary terms, tensor structure as on a full
    rectangle) reduces the error of A u = b, A = H + V, by at least that factor."""
    from qpsim_amd.engine import peaceman_rachford_parameters
    a = 0.3
    for reduction in (1e-6, 1e-11):
        ps, worst = peaceman_rachford_parameters(0.5, 0.5 + 4.0 * a, reduction)
        assert worst <= reduction and 2 <= len(ps) <= 12
        shorter, w2 = peaceman_rachford_parameters(0.5, 0.5 + 4.0 * a, reduction, jmax=len(ps) - 1)
        assert w2 > reduction      # the returned J is the smallest that meets the bound

    def lap(n, e_lo, e_hi):
        L = -2.0 * np.eye(n) + np.eye(n, k=1) + np.eye(n, k=-1)
        L[0, 0] = -(1.0 + e_lo)
        L[-1, -1] = -(1.0 + e_hi)
        return L

    nx, ny = 24, 17
    Hx = 0.5 * np.eye(nx) - a * lap(nx, 0.0, 2.0)
    Vy = 0.5 * np.eye(ny) - a * lap(ny, 0.4, 0.0)
    H, V = np.kron(np.eye(ny), Hx), np.kron(Vy, np.eye(nx))
    rng = np.random.default_rng(0)
    x = rng.random(nx * ny)
    b = (H + V) @ x
    ps, worst = peaceman_rachford_parameters(0.5, 0.5 + a * 4.0, 1e-9)
    u = np.zeros_like(x)
    I = np.eye(nx * ny)
    for p in ps:
        us = np.linalg.solve(H + p * I, b - (V - p * I) @ u)
        u = np.linalg.solve(V + p * I, b - (H - p * I) @ us)
    assert np.linalg.norm(u - x) <= 1.01 * worst * np.linalg.norm(x)


def test_fine_tile_block_mapping_is_a_bijection_that_keeps_super_tiles_on_one_xcd_label():
    """`fine_block` (csrc/qp_adi_fine.inc), restated: block -> (64 x 64 super-tile, half).  Every (super-tile, half) is hit
    exactly once, and both halves of a super-tile come from blocks with equal `blockIdx % 8` (the XCD label under round-robin
    dispatch) in every full group of 8 super-tiles - which is what keeps a tile on one XCD's L2 from pass to pass."""
    def fine_block(bid, ns):
        group, r = bid >> 4, bid & 15
        m = min(8, ns - group * 8)
        return group * 8 + r % m, r // m

    for ns in (1, 2, 5, 8, 9, 16, 23, 64, 1000):
        seen = {}
        for bid in range(2 * ns):
            S, sub = fine_block(bid, ns)
            assert 0 <= S < ns and sub in (0, 1) and (S, sub) not in seen
            seen[(S, sub)] = bid
        assert len(seen) == 2 * ns
        for S in range(8 * (ns // 8)):
            assert seen[(S, 0)] % 8 == seen[(S, 1)] % 8 == S % 8


def test_external_generation_matches_reference_and_every_strategy_agrees():
    """`solver.evaluate_external_generation` against the reference's own outputs (host_side.npz: constant, pulse inside /
    at the end of / before its window, a vectorisable and a scalar custom body), and its three evaluation strategies
    (one broadcast call, per bin, per cell) against each other."""
    from qpsim_amd.safe_eval import expression_is_elementwise
    from qpsim_amd.solver import _CustomGeneration, evaluate_external_generation
    z, meta = _z("host_side.npz")
    mask, E = z["ic_mask"], z["ic_E"]
    n = int(mask.sum())
    for nm in ("const", "pulse_in", "pulse_end", "pulse_before", "custom_vec", "custom_scalar"):
        spec = ExternalGenerationSpec(**meta[f"gen_{nm}"]["spec"])
        got = evaluate_external_generation(spec, E, n, meta[f"gen_{nm}"]["t"], mask)
        assert got.shape == (E.size, n) and np.allclose(got, z[f"gen_{nm}"], rtol=1e-15, atol=0), nm
    assert evaluate_external_generation(ExternalGenerationSpec(mode="none"), E, n, 0.0, mask) is None

    arrays = ("E", "x", "y")
    for body, elementwise in (("params['g'] * np.exp(-E / 400.0) * (1 + x) * (t < 1.0)", True),
                              ("np.where(x > 0.5, E, 2 * y)", True), ("3e-9", True), ("1.0 if t < 1 else 0.0", True),
                              ("1.0 if E > 300 else 0.0", False), ("(E > 300) and (t < 1)", False), ("x[0] + E", False),
                              ("x.size * E", False), ("len(x) * E", False), ("math.exp(-E / 400.0)", False),
                              ("np.arange(params['nx']) * E", True), ("np.zeros_like(x) + E", False), ("0.2 < x < 0.8", False)):
        assert expression_is_elementwise(body, array_variables=arrays) is elementwise, body

    # an expression only the per-bin strategy can vectorise (scalar E in a conditional), and one that needs the per-cell loop
    for body in ("(1e-8 if E > 300 else 2e-8) * (1 + x)", "math.exp(-E / 400.0) * 1e-8 * (1 + math.sin(y))"):
        spec = ExternalGenerationSpec(mode="custom", custom_body=body)
        gen = _CustomGeneration(spec, mask)
        want = gen._per_cell(np.asarray(E, dtype=float), 0.3, (E.size, n))
        assert np.allclose(evaluate_external_generation(spec, E, n, 0.3, mask), want, rtol=1e-15, atol=0)
    spec = ExternalGenerationSpec(mode="custom", custom_body="1e-8 * np.exp(-E / 400.0) * np.maximum(x - y, 0.0) * (t < 1)")
    gen = _CustomGeneration(spec, mask)
    shape = (E.size, n)
    a, b, c = (f(np.asarray(E, dtype=float), 0.3, shape) for f in (gen._on_grid, gen._per_bin, gen._per_cell))
    assert np.allclose(a, b, rtol=1e-15, atol=0) and np.allclose(b, c, rtol=1e-15, atol=0)

    # the reference's messages (solver.py:889-902, :944-947)
    with pytest.raises(ValueError, match="produced negative values"):
        evaluate_external_generation(ExternalGenerationSpec(mode="custom", custom_body="-1.0"), E, n, 0.0, mask)
    with pytest.raises(ValueError, match="produced non-finite values"):
        evaluate_external_generation(ExternalGenerationSpec(mode="custom", custom_body="np.log(x - 1)"), E, n, 0.0, mask)
    with pytest.raises(Exception):      # 3 values per bin: the vectorised attempts refuse, the per-cell loop cannot convert
        evaluate_external_generation(ExternalGenerationSpec(mode="custom", custom_body="np.arange(3) * 1.0"), E, n, 0.0, mask)
