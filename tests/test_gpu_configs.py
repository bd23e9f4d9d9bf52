"""BASELINE.json configurations at their full sizes, checked through what the domain offers where the oracle cannot run the
whole problem: sampled-pixel oracle checks (the collision update is pixel-local), member-by-member equality of batched
ensembles, and an extended-precision restatement that prices the tolerances of the phonon update."""
from __future__ import annotations

import numpy as np
import pytest

from golden_utils import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import qp_oracle
    return qp_oracle


def _c3_tables(T, ne=12, fmax=3.0):
    gap = 180.0
    E, dE = T.build_energy_grid(gap, 1.0, fmax, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = T.dynes_density_of_states(E, gap, 0.0)
    kr = T.recombination_kernel_base(E, gap, 440.0, 1.2)
    ks = T.scattering_kernel_base(E, gap, 440.0, 1.2)
    return E, dE, om, idx_d, idx_s, sg, rho, kr, ks


def test_config3_full_size_collision_beyond_4gib_matches_oracle_on_sampled_pixels(O):
    """BASELINE configs[2]: 4096 x 4096, NE = 12, Nw = 35, dynamic phonons - 47 planes of 134 MB (6.3 GB, plane offsets far
    beyond 32 bits).  One collision call on the whole grid; ~4000 random pixels plus the first and last cells of the planes
    are gathered and compared with the oracle's per-pixel update of the same inputs."""
    import torch
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    N, ne = 4096, 12
    E, dE, om, idx_d, idx_s, sg, rho, kr, ks = _c3_tables(T, ne)
    assert om.size == 35
    mask = np.ones((1, N * N), dtype=bool)
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    tab = eng.make_collision_tables(kr[None], ks[None], rho[None], idx_d, idx_s, sg)
    assert tab["kernel"] == "register"
    dev = eng.device
    g = torch.Generator(device=dev).manual_seed(11)
    ncell = N * N
    occ = torch.rand((1, ncell), dtype=torch.float64, device=dev, generator=g)
    # occupations from 1e-6 to 0.9 of the density of states, different in every bin
    state = torch.as_tensor(rho, device=dev)[:, None] * (1e-6 + 0.9 * occ * torch.rand((ne, ncell), dtype=torch.float64,
                                                                                        device=dev, generator=g))
    ph = torch.as_tensor(T.thermal_phonon_occupation(om, 0.25), device=dev)[:, None] * (
        0.5 + torch.rand((om.size, ncell), dtype=torch.float64, device=dev, generator=g))
    assert (state.numel() + ph.numel()) * 8 > 4 * 2**30
    rng = np.random.default_rng(5)
    sel = np.unique(np.concatenate([rng.integers(0, ncell, size=4000), np.arange(0, 130), np.arange(ncell - 130, ncell),
                                    np.arange(2**23 - 65, 2**23 + 65)]))
    sel_d = torch.as_tensor(sel, device=dev)
    s_in, p_in = state[:, sel_d].cpu().numpy(), ph[:, sel_d].cpu().numpy()
    out = torch.empty_like(state)
    dt = 0.05
    eng.collide(tab, state, out, ph, dE, dt, True, True, True)
    s_out, p_out = out[:, sel_d].cpu().numpy(), ph[:, sel_d].cpu().numpy()
    tables = {"rho": rho[None], "Kr0": kr[None], "Ks0": ks[None], "cls": np.zeros(sel.size, dtype=int),
              "idx_diff": idx_d, "idx_sum": idx_s, "sign": sg, "dE": dE}
    O.collision_step(s_in, p_in, tables, dt, en_r=True, en_s=True, update_phonons=True)
    assert rel_err(s_out, s_in) < 2e-11 and rel_err(p_out, p_in) < 2e-11
    # per-pixel bound as well (a wrong plane offset would hit single pixels, not the global norm)
    assert np.max(np.abs(s_out - s_in) / np.maximum(np.abs(s_in), 1e-300)) < 1e-9
    assert float(out.min()) >= 0.0 and bool(torch.isfinite(out).all()) and bool(torch.isfinite(ph).all())
    del state, ph, out
    torch.cuda.empty_cache()


def test_config4_batched_ensemble_equals_members_run_one_at_a_time():
    """BASELINE configs[3] (ensemble of independent 256 x 256 MKID pixels): 8 members batched as extra planes
    ([bin][member][cell]), 3 coupled steps (collision half-steps, ADI, guard) - bit-equal to each member run alone."""
    import torch
    from qpsim_amd.bench_workloads import CoupledWorkload
    members, N, steps = 8, 256, 3
    dev = torch.device("cuda", torch.cuda.current_device())
    batch = CoupledWorkload(N, dev, members=members)
    ne, ncell = batch.ne, N * N
    assert batch.state.shape == (ne, members * ncell)
    init_s = batch.state.clone()
    init_p = batch.phonon.clone()
    # members differ (their own seeds) and every member sees its own phonon field
    init_p *= 1.0 + 0.1 * torch.rand(init_p.shape, dtype=torch.float64, device=dev,
                                     generator=torch.Generator(device=dev).manual_seed(2))
    batch.phonon.copy_(init_p)
    assert not torch.equal(init_s[:, :ncell], init_s[:, ncell:2 * ncell])
    batch.run(steps)
    torch.cuda.synchronize()
    single = CoupledWorkload(N, dev, members=1)
    for m in range(members):
        sl = slice(m * ncell, (m + 1) * ncell)
        single.state.copy_(init_s[:, sl])
        single.phonon.copy_(init_p[:, sl])
        single.run(steps)
        torch.cuda.synchronize()
        assert torch.equal(single.state, batch.state[:, sl]), m
        assert torch.equal(single.phonon, batch.phonon[:, sl]), m
    assert batch.max_occ > 0.0


@pytest.mark.parametrize("ne,fmax", [(12, 3.0), (24, 4.0), (50, 10.0)])
def test_phonon_tolerance_is_the_conditioning_of_the_reference_formula_not_a_kernel_error(O, ne, fmax):
    """The reference forms (e^{b dt} - 1)/b and (1 - e^{-mu dt})/mu without expm1 (solver.py:661,697).  The same algorithm
    evaluated in 80-bit extended precision (the oracle's routine on longdouble inputs) prices that: the fp64 oracle and the
    HIP kernel must sit equally close to it - if the kernel were wrong it would be farther from the extended-precision
    value than the fp64 restatement of the reference is."""
    if np.finfo(np.longdouble).eps > 2e-19:
        pytest.skip("no 80-bit long double on this host")
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    rng = np.random.default_rng(ne)
    n = 512
    mask = np.ones((1, n), dtype=bool)
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    gap, gamma = 180.0, 0.1
    E, dE = T.build_energy_grid(gap, 1.0, fmax, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = T.dynes_density_of_states(E, gap, gamma)
    kr, ks = T.recombination_kernel_base(E, gap, 500.0, 1.2), T.scattering_kernel_base(E, gap, 400.0, 1.2)
    state = rng.random((ne, n)) * rho[:, None] * rng.choice([1e-5, 1e-2, 0.5, 0.95], size=n)[None, :]
    ph = T.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, n)))
    dt = 0.37
    tab = eng.make_collision_tables(kr[None], ks[None], rho[None], idx_d, idx_s, sg)
    assert tab["kernel"] == "register"
    s_in, p_dev = eng.upload_packed(state), eng.upload_packed(ph)
    s_out = eng.empty(ne, eng.ncell)
    eng.collide(tab, s_in, s_out, p_dev, dE, dt, True, True, True)
    s_hip, p_hip = eng.download_packed(s_out), eng.download_packed(p_dev)
    s64, p64 = O.collision_pixels(state, ph, kr, ks, rho, idx_d, idx_s, sg, dE, dt, en_r=True, en_s=True)
    L = np.longdouble
    sL, pL = O.collision_pixels(state.astype(L), ph.astype(L), kr.astype(L), ks.astype(L), rho.astype(L), idx_d, idx_s, sg,
                                L(dE), L(dt), en_r=True, en_s=True)
    assert sL.dtype == L and pL.dtype == L

    def err(a, b):
        return float(np.max(np.abs(a.astype(L) - b)) / np.max(np.abs(b)))

    e_hip_s, e_ref_s = err(s_hip, sL), err(s64, sL)
    e_hip_p, e_ref_p = err(p_hip, pL), err(p64, pL)
    print(f"NE={ne}: state  |hip - x87| = {e_hip_s:.2e}  |fp64 oracle - x87| = {e_ref_s:.2e};  "
          f"phonons  |hip - x87| = {e_hip_p:.2e}  |fp64 oracle - x87| = {e_ref_p:.2e}")
    # the kernel is as close to the extended-precision value as the fp64 restatement of the reference (factor 4 covers the
    # different summation orders; 1e-14 the plain rounding floor)
    assert e_hip_s <= 4.0 * e_ref_s + 1e-14
    assert e_hip_p <= 4.0 * e_ref_p + 1e-14
    # and the distance between the two fp64 results is bounded by their distances to the extended-precision value
    assert rel_err(p_hip, p64) <= 1.01 * (e_hip_p + e_ref_p) + 1e-16


def _rect_problem(ny, nx):
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    mask = np.ones((ny, nx), dtype=bool)
    edges = extract_edge_segments(mask)
    side_bc = {"left": BoundaryCondition("dirichlet", 0.7), "right": BoundaryCondition("robin", 0.4, 0.2),
               "up": BoundaryCondition("neumann", -0.3), "down": BoundaryCondition("absorbing")}
    return mask, edges, {e.edge_id: side_bc[e.normal] for e in edges}


@pytest.mark.parametrize("ny,nx,D", [(96, 80, 150.0), (70, 130, 40.0), (40, 1, 300.0)])
def test_exact_cn_on_stiff_steps_converges_by_chebyshev_or_raises(O, ny, nx, D):
    """r D ~ 10 (dt = 1, D = 6, dx = 0.5 gives 12): plain Richardson contracts by ~0.97 per iteration and cannot
    reach 1e-13 in a few hundred iterations; the Chebyshev semi-iteration on the same kernels does, and a budget that is too
    small raises instead of returning a partial solve (the reference's SuperLU solve is exact for any dt)."""
    from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry
    mask, edges, bcs = _rect_problem(ny, nx)
    dx, dt = 0.9, 0.11
    eng = Engine(compile_geometry(mask, edges, bcs, dx))
    op = DiffusionOperator(eng, 2, dt, dcoef=[D, 0.3 * D])
    assert op.r * D > 2.5
    u0 = np.random.default_rng(ny).random((2, ny * nx))
    v = eng.upload_packed(u0)
    eng.scratch("cn_d", 2 * ny * nx).fill_(float("nan"))      # the direction buffer starts uninitialised: must not matter
    its = eng.cn_exact_step(op, v)
    ops = O.build_grid_ops(mask, edges, bcs, dx)
    got = eng.download_packed(v)
    for k, d in enumerate([D, 0.3 * D]):
        assert rel_err(got[k], O.CNStepper(ops, d, dt).step(u0[k])) < 1e-11
    if min(ny, nx) > 1:
        rho = eng.cn_contraction_bound(op)
        plain = np.log(1e-13) / np.log(rho)
        assert rho > 0.5 and its < 0.5 * plain and its < 300, (its, plain)
        # second call: the iteration count of the first runs blind, the result is the same
        v2 = eng.upload_packed(u0)
        eng.cn_exact_step(op, v2)
        assert rel_err(eng.download_packed(v2), got) < 1e-12
        with pytest.raises(RuntimeError, match="did not reach rtol"):
            op2 = DiffusionOperator(eng, 2, dt, dcoef=[D, 0.3 * D])
            eng.cn_exact_step(op2, eng.upload_packed(u0), max_iter=3)
    else:
        assert its < 300         # a strip with absorbing / Robin side walls: L_x is diagonal but not zero, M != A


def test_pauli_reduction_propagates_nan_like_argmax():
    """A diverged state (NaN everywhere, or a single NaN) must not produce an out-of-range index: np.argmax treats NaN as
    the maximum and returns the first one (solver.py:993); the reference then reports max_occ = NaN and goes on."""
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    mask = np.ones((3, 5), dtype=bool)
    mask[1, 2] = False
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    n = int(mask.sum())
    rho = np.array([[1.0, 2.0, 4.0]])
    idx = np.zeros((3, 3), dtype=np.int32)
    tab = eng.make_collision_tables(None, None, rho, idx, idx, idx.astype(np.int8))
    state = np.full((3, n), np.nan)
    mx, top, forb = eng.pauli_stats(eng.upload_packed(state), tab, 1e-18)
    assert np.isnan(mx) and top == (0, 0) and forb is None
    state = np.random.default_rng(0).random((3, n))
    state[1, 7] = np.nan
    state[2, 3] = np.nan
    mx, top, forb = eng.pauli_stats(eng.upload_packed(state), tab, 1e-18)
    cell_of_px = np.flatnonzero(mask.reshape(-1))
    assert np.isnan(mx) and top == (1, cell_of_px[7])
    state = -np.ones((3, n))          # all occupations negative: the maximum is the least negative one, index in range
    state[2, 5] = -0.5
    mx, top, forb = eng.pauli_stats(eng.upload_packed(state), tab, 1e-18)
    f = state / rho.T
    assert mx == f.max() and top == (int(np.argmax(f) // n), cell_of_px[int(np.argmax(f) % n)])


def test_step_api_with_asymmetric_tables_runs_the_general_kernel(O):
    """Caller-supplied K_r0 / K_s0 / bin maps of the public step API need not be symmetric: such tables are
    routed to the generic kernel, which reads (i, j) and (j, i) separately - checked against the oracle's restatement of
    solver.py:703-791, which makes no symmetry assumption either."""
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    from qpsim_amd.solver import apply_collision_step_fischer_catelani_uniform
    rng = np.random.default_rng(21)
    ne, n = 7, 40
    E, dE = T.build_energy_grid(180.0, 1.0, 3.0, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = T.dynes_density_of_states(E, 180.0, 0.1)
    kr = T.recombination_kernel_base(E, 180.0, 500.0, 1.2) * (1.0 + 0.3 * rng.random((ne, ne)))
    ks = T.scattering_kernel_base(E, 180.0, 400.0, 1.2) * (1.0 + 0.3 * rng.random((ne, ne)))
    idx_d2 = idx_d.copy()
    idx_d2[1, 4] = idx_d[2, 5] + 1            # an asymmetric bin map as well
    assert not np.array_equal(kr, kr.T) and not np.array_equal(idx_d2, idx_d2.T)
    mask = np.ones((1, n), dtype=bool)
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    tab = eng.make_collision_tables(kr[None], ks[None], rho[None], idx_d2, idx_s, sg)
    assert tab["kernel"] == "generic" and not tab["symmetric"]
    assert eng.make_collision_tables(kr[None] + kr.T[None], None, rho[None], idx_d, idx_s, sg)["kernel"] != "generic"
    state = rng.random((ne, n)) * rho[:, None] * 0.5
    ph = T.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, n)))
    s, p = state.copy(), ph.copy()
    apply_collision_step_fischer_catelani_uniform(s, p, kr, ks, rho, idx_d2, idx_s, sg, dE, 0.2,
                                                  enable_recombination=True, enable_scattering=True)
    s_ref, p_ref = O.collision_pixels(state, ph, kr, ks, rho, idx_d2, idx_s, sg, dE, 0.2, en_r=True, en_s=True)
    assert rel_err(s, s_ref) < 1e-12 and rel_err(p, p_ref) < 1e-11


@pytest.mark.parametrize("energy", [False, True])
def test_adi_runs_batch_the_steps_between_store_points(O, energy):
    """`diffusion_scheme="adi"`: the steps between two store points are one library call (carried right-hand side, 32 B per
    cell-update) in scalar mode and in diffusion-only energy-resolved runs whose guard cannot fire; remainder step, store
    cadence, callback times and results must equal the oracle's step-by-step ADI."""
    from qpsim_amd.solver import run_2d_crank_nicolson
    mask, edges, bcs = _rect_problem(70, 96)
    init = np.random.default_rng(3).random(mask.shape) * 1e-4
    kw = dict(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
              total_time=1.25, dx=1.0, store_every=4)
    if energy:
        kw.update(energy_gap=180.0, energy_max_factor=3.0, num_energy_bins=5, pauli_warn_threshold=None,
                  pauli_error_threshold=None)
    seen = []
    got = run_2d_crank_nicolson(**kw, diffusion_scheme="adi", progress_callback=lambda t, f: seen.append(t))
    quiet = run_2d_crank_nicolson(**kw, diffusion_scheme="adi")
    ref = O.run(**kw, scheme="adi")
    assert got[0] == pytest.approx(ref[0]) and seen == got[0] and len(got[0]) == 5      # t = 0, 0.4, 0.8, 1.2, 1.25
    assert rel_err(np.stack(got[1]), np.stack(ref[1])) < 1e-12
    assert np.array_equal(np.stack(got[1]), np.stack(quiet[1]), equal_nan=True) and got[2] == quiet[2]
    assert np.allclose(got[2], ref[2], rtol=1e-12)
    if energy:
        assert rel_err(np.stack([np.stack(t) for t in got[4]]), np.stack([np.stack(t) for t in ref[4]])) < 1e-12


def test_store_points_download_asynchronously_and_in_order():
    """Every step stored (store_every = 1), NE + Nw + 2 planes per store through the three-slot pinned ring: the frames come
    back in order and equal a run that stores only the last step (same state, different download cadence)."""
    from qpsim_amd.solver import run_2d_crank_nicolson
    mask, edges, bcs = _rect_problem(48, 64)
    init = np.random.default_rng(8).random(mask.shape) * 1e-4
    kw = dict(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
              total_time=0.9, dx=1.0, energy_gap=180.0, energy_max_factor=3.0, num_energy_bins=6,
              enable_recombination=True, enable_scattering=True, diffusion_scheme="adi")
    ph_all, ph_last = {}, {}
    every = run_2d_crank_nicolson(**kw, store_every=1, phonon_history_out=ph_all)
    last = run_2d_crank_nicolson(**kw, store_every=9, phonon_history_out=ph_last)
    assert len(every[1]) == 10 and len(last[1]) == 2 and all(f is not None for f in every[1])
    assert np.array_equal(every[1][-1], last[1][-1], equal_nan=True) and every[2][-1] == last[2][-1]
    assert np.array_equal(np.stack(every[4][-1]), np.stack(last[4][-1]), equal_nan=True)
    assert np.array_equal(np.stack(ph_all["phonon_energy_frames"][-1]), np.stack(ph_last["phonon_energy_frames"][-1]),
                          equal_nan=True)
    assert len(ph_all["phonon_frames"]) == 10 and all(np.isfinite(m) for m in every[2])


@pytest.mark.parametrize("ne,nclass,mode", [(6, 1, "plain"), (12, 1, "ties"), (12, 1, "forbidden"), (16, 3, "plain"),
                                            (24, 1, "plain"), (12, 2, "forbidden"), (50, 1, "plain"), (33, 1, "plain")])
def test_fused_pauli_guard_equals_the_separate_reduction(ne, nclass, mode):
    """qp_collision_step_guarded (statistics reduced inside the register kernels, one partial per wave) must return exactly
    what qp_pauli_stats finds in the new state: value, argmax in np.argmax order (ties!), first forbidden index.  Sizes
    without the fused epilogue (NE >= 32 split kernels, 33 = wave kernel) take the separate pass inside the same call."""
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    rng = np.random.default_rng(ne * 3 + nclass)
    mask = rng.random((37, 53)) > 0.25                 # 1961 cells: neither a multiple of 64 nor of 128; ~25 % inactive
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    n = int(mask.sum())
    gaps = np.array([180.0, 171.0, 165.5])[:nclass]
    E, dE = T.build_energy_grid(180.0, 1.0, 3.0 if ne <= 24 else 10.0, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = np.stack([T.dynes_density_of_states(E, g, 0.1) for g in gaps])
    kr = np.stack([T.recombination_kernel_base(E, g, 500.0, 1.2) for g in gaps])
    ks = np.stack([T.scattering_kernel_base(E, g, 400.0, 1.2) for g in gaps])
    cls = rng.integers(0, nclass, size=n)
    state = rng.random((ne, n)) * rho[cls].T * rng.choice([1e-5, 1e-2, 0.5, 0.9], size=n)[None, :]
    if mode == "forbidden":                            # a bin without states that nevertheless holds density
        rho[:, 2] = 0.0
        state[2, :] = 0.0
        state[2, [n // 3, n // 2]] = 1e-6
    if mode == "ties":                                 # identical pixels: the same occupation twice, the first index wins
        state[:, 700] = state[:, 40]
        state[:, 41] = state[:, 40]
        state[:, 40] *= 0.97 / np.max(state[:, 40] / rho[0])   # near-full occupation: they stay the maximum after the update
        state[:, 41] = state[:, 40]
        state[:, 700] = state[:, 40]
    ph = T.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, n)))
    if mode == "ties":
        ph[:, 41] = ph[:, 40]
        ph[:, 700] = ph[:, 40]
    tab = eng.make_collision_tables(kr, ks, rho, idx_d, idx_s, sg, cls if nclass > 1 else None,
                                    gap_params=dict(E=E, gaps=gaps, tau_r=500.0, tau_s=400.0, T_c=1.2))
    if ne in (6, 12, 16, 24):
        assert tab["kernel"] == "register"
    res = {}
    for fused in (False, True):
        s_in, p_dev = eng.upload_packed(state), eng.upload_packed(ph)
        s_out = eng.empty(ne, eng.ncell)
        if fused:
            ticket = eng.collide_guarded(tab, s_in, s_out, p_dev, dE, 0.2, True, True, True, 1e-18)
            stats = eng.pauli_stats_result(ticket)
        else:
            eng.collide(tab, s_in, s_out, p_dev, dE, 0.2, True, True, True)
            stats = eng.pauli_stats(s_out, tab, 1e-18)
        res[fused] = (stats, eng.download_packed(s_out), eng.download_packed(p_dev))
    assert np.array_equal(res[True][1], res[False][1]) and np.array_equal(res[True][2], res[False][2])
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])
    mx, top, forb = res[True][0]
    out = res[True][1]
    f = np.where(rho[cls].T > 1e-30, out / np.maximum(rho[cls].T, 1e-30), 0.0)
    cell_of_px = np.flatnonzero(mask.reshape(-1))
    k = int(np.argmax(f))
    assert mx == f.reshape(-1)[k] and top == (k // n, cell_of_px[k % n])
    if mode == "forbidden":
        assert forb is not None and forb[0] == 2 and forb[1] == cell_of_px[n // 3]
    else:
        assert forb is None
    if mode == "ties":         # the three identical pixels give identical results; among them the first index must win
        assert np.array_equal(out[:, 40], out[:, 41]) and np.array_equal(out[:, 40], out[:, 700])
        assert top[1] == cell_of_px[40], "the identical pixels carry the maximum by construction"


@pytest.mark.parametrize("ny,nx", [(70, 130), (1, 100), (64, 1), (3, 5)])
def test_rect_combine_equals_the_general_stencil(ny, nx):
    """qp_adi_rect_combine (operator from the plan's four side terms, norm fused) vs qp_stencil_combine (per-cell geometry
    arrays) for every coefficient set the exact-CN iteration uses, plus the fused max-norm."""
    import torch
    from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry
    mask, edges, bcs = _rect_problem(ny, nx)
    eng = Engine(compile_geometry(mask, edges, bcs, 0.9))
    rect = DiffusionOperator(eng, 3, 0.11, dcoef=[6.0, 0.35, 0.0])
    gen = DiffusionOperator(eng, 3, 0.11, dcoef=[6.0, 0.35, 0.0], allow_fast=False)
    assert rect.rect is not None and gen.rect is None
    g = torch.Generator(device=eng.device).manual_seed(ny + nx)
    u = torch.rand((3, ny * nx), dtype=torch.float64, device=eng.device, generator=g)
    rin = torch.rand((3, ny * nx), dtype=torch.float64, device=eng.device, generator=g)
    for coef, use_rin in (((1.0, 1.0, 1.0, 2.0, 0.0), False), ((-1.0, 1.0, 1.0, 0.0, 1.0), True),
                          ((1.0, 0.0, 1.0, 1.0, 0.0), False), ((0.5, -0.25, 2.0, 1.0, -3.0), True)):
        a, b = eng.empty(3, ny * nx), eng.empty(3, ny * nx)
        norm = eng.empty(1)
        eng.stencil(rect, u, a, *coef[:4], rin=rin if use_rin else None, cr=coef[4], norm_out=norm)
        eng.stencil(gen, u, b, *coef[:4], rin=rin if use_rin else None, cr=coef[4])
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 4e-15 * scale
        assert float(norm.item()) == float(a.abs().max())


def test_fused_guard_workspace_holds_every_partial_the_kernels_write():
    """160 x 213 = 34080 cells (> 32768, and 34080 % 128 = 32 lies in [1, 64]): the register kernels launch
    ceil(ncell / 128) blocks of two waves and every wave writes a partial - one more than ceil(ncell / 64).  The workspace
    is allocated at EXACTLY `qp_collision_guard_workspace_bytes` with a canary behind it; the fused result must equal the
    separate reduction and the canary must survive."""
    import ctypes as C
    import torch
    from qpsim_amd import _hip
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    ny, nx, ne = 160, 213, 12
    ncell = ny * nx
    assert ncell > 32768 and 1 <= ncell % 128 <= 64
    mask = np.ones((ny, nx), dtype=bool)
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    lib = eng.lib
    nbytes = int(lib.qp_collision_guard_workspace_bytes(ncell))
    assert nbytes >= 24 * (2 * ((ncell + 127) // 128) + 512)
    E, dE = T.build_energy_grid(180.0, 1.0, 3.0, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = T.dynes_density_of_states(E, 180.0, 0.1)
    kr, ks = T.recombination_kernel_base(E, 180.0, 500.0, 1.2), T.scattering_kernel_base(E, 180.0, 400.0, 1.2)
    tab = eng.make_collision_tables(kr[None], ks[None], rho[None], idx_d, idx_s, sg)
    assert tab["kernel"] == "register"
    rng = np.random.default_rng(4)
    state = rng.random((ne, ncell)) * rho[:, None] * 0.5
    state[:, -1] = 0.93 * rho                      # the maximum sits in the very last cell: the partial of the LAST wave
    ph = T.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, ncell)))
    s_in, p_dev = eng.upload_packed(state), eng.upload_packed(ph)
    s_out = eng.empty(ne, ncell)
    pad = 4096
    raw = torch.full((nbytes + pad,), 0xA5, dtype=torch.uint8, device=eng.device)
    vals = torch.zeros(2, dtype=torch.float64, device=eng.device)
    idx = torch.zeros(2, dtype=torch.int64, device=eng.device)
    _hip.check(lib.qp_collision_step_guarded(C.byref(tab["struct"]), int(eng.d_flags.data_ptr()), ncell,
                                             int(s_in.data_ptr()), int(s_out.data_ptr()), int(p_dev.data_ptr()), 0,
                                             float(dE), 0.2, 1, 1, 1, 1e-18, int(raw.data_ptr()), int(vals.data_ptr()),
                                             int(idx.data_ptr()), eng.stream), "qp_collision_step_guarded")
    torch.cuda.synchronize()
    assert bool((raw[nbytes:] == 0xA5).all()), "the fused guard wrote past its workspace"
    mx, top, forb = eng.pauli_stats(s_out, tab, 1e-18)
    assert float(vals[0]) == mx and int(idx[0]) == top[0] * ncell + top[1] and int(idx[1]) == -1 and forb is None
    assert top[1] == ncell - 1


@pytest.mark.parametrize("combo", [(True, True, True), (True, True, False), (True, False, True), (False, True, True)])
def test_one_pass_ne50_kernel_equals_the_split_kernels_on_a_ragged_masked_grid(monkeypatch, combo):
    """NE = 50 (the reference's default `num_energy_bins`): the one-launch kernel (256-thread blocks, tables staged in LDS,
    lanes past the end of the grid and cells outside the mask inside a block) against the three-launch split path on the
    same inputs - quasiparticle planes to rounding (q = max(rho - n, 0) on both), phonon planes within the conditioning
    bound of the reference's (e^x - 1)/x (the 80-bit test above prices it), masked cells passed through untouched."""
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    en_r, en_s, upd = combo
    ne = 50
    rng = np.random.default_rng(17)
    mask = rng.random((23, 37)) > 0.2                     # 851 cells: 4 blocks of 256 threads, the last one ragged
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    assert eng.ncell % 256 != 0 and bool(eng.lib.qp_collision_onepass_available(ne))
    n = int(mask.sum())
    E, dE = T.build_energy_grid(180.0, 1.0, 10.0, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = T.dynes_density_of_states(E, 180.0, 0.1)
    kr, ks = T.recombination_kernel_base(E, 180.0, 500.0, 1.2), T.scattering_kernel_base(E, 180.0, 400.0, 1.2)
    state = rng.random((ne, n)) * rho[:, None] * rng.choice([1e-5, 1e-2, 0.5, 0.95], size=n)[None, :]
    ph = T.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, n)))
    tab = eng.make_collision_tables(kr[None], ks[None], rho[None], idx_d, idx_s, sg)
    assert tab["kernel"] == "register" and tab["ks0_diag"] is not None and tab["kr0_anti2"] is not None
    outs = {}
    for onepass in ("1", "0"):
        monkeypatch.setenv("QPSIM_COLL_ONEPASS", onepass)
        s_in, p_dev = eng.upload_packed(state), eng.upload_packed(ph)
        s_out = eng.empty(ne, eng.ncell)
        s_out.fill_(-7.0)
        eng.collide(tab, s_in, s_out, p_dev, dE, 0.37, en_r, en_s, upd)
        outs[onepass] = (eng.download_packed(s_out), eng.download_packed(p_dev), s_out.cpu().numpy(), p_dev.cpu().numpy())
    assert rel_err(outs["1"][0], outs["0"][0]) < 1e-13
    assert rel_err(outs["1"][1], outs["0"][1]) < (1e-10 if upd else 1e-300)
    hole = ~mask.reshape(-1)
    assert np.all(outs["1"][2][:, hole] == 0.0) and np.all(outs["1"][3][:, hole] == 0.0)
    if not upd:
        assert np.array_equal(outs["1"][1], ph)


@pytest.mark.parametrize("combo", [(True, True, True), (True, True, False), (True, False, True), (False, True, True)])
def test_one_pass_ne50_gap_class_kernel_equals_the_split_kernels(monkeypatch, combo):
    """Non-uniform gap at NE = 50 (per-pixel tables of solver.py:1203-1232 as 5 gap classes, one of them unused): the
    one-launch kernel forms K per lane from the amplitude tables staged in LDS (compact (anti)diagonal order in the phonon
    phase, per-class rho rows read per lane); against the three-launch gap-class path on a ragged masked grid whose blocks
    mix classes lane by lane."""
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    en_r, en_s, upd = combo
    ne = 50
    rng = np.random.default_rng(23)
    mask = rng.random((23, 37)) > 0.2
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    n = int(mask.sum())
    gaps = np.array([180.0, 171.0, 165.5, 176.25, 150.0])
    E, dE = T.build_energy_grid(180.0, 1.0, 10.0, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = np.stack([T.dynes_density_of_states(E, g, 0.1) for g in gaps])
    kr = np.stack([T.recombination_kernel_base(E, g, 500.0, 1.2) for g in gaps])
    ks = np.stack([T.scattering_kernel_base(E, g, 400.0, 1.2) for g in gaps])
    cls = rng.integers(0, 4, size=n)
    state = rng.random((ne, n)) * rho[cls].T * rng.choice([1e-5, 1e-2, 0.5, 0.95], size=n)[None, :]
    ph = T.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, n)))
    tab = eng.make_collision_tables(kr, ks, rho, idx_d, idx_s, sg, cls,
                                    gap_params=dict(E=E, gaps=gaps, tau_r=500.0, tau_s=400.0, T_c=1.2))
    assert tab["kernel"] == "register"
    outs = {}
    for onepass in ("1", "0"):
        monkeypatch.setenv("QPSIM_COLL_ONEPASS", onepass)
        s_in, p_dev = eng.upload_packed(state), eng.upload_packed(ph)
        s_out = eng.empty(ne, eng.ncell)
        s_out.fill_(-7.0)
        eng.collide(tab, s_in, s_out, p_dev, dE, 0.37, en_r, en_s, upd)
        outs[onepass] = (eng.download_packed(s_out), eng.download_packed(p_dev), s_out.cpu().numpy(), p_dev.cpu().numpy())
    assert rel_err(outs["1"][0], outs["0"][0]) < 1e-13
    assert rel_err(outs["1"][1], outs["0"][1]) < (1e-10 if upd else 1e-300)
    hole = ~mask.reshape(-1)
    assert np.all(outs["1"][2][:, hole] == 0.0) and np.all(outs["1"][3][:, hole] == 0.0)
    if not upd:
        assert np.array_equal(outs["1"][1], ph)


@pytest.mark.parametrize("ne,combo", [(12, (True, True, True)), (12, (True, False, False)), (12, (False, True, True)),
                                      (8, (True, True, True)), (8, (True, True, False)), (16, (True, True, True)), (5, (False, True, True))])
def test_double_half_step_kernel_equals_two_calls_bit_for_bit(ne, combo):
    """qp_collision_double_step_guarded = guarded half-step + generation term + half-step with the intermediate state kept
    in registers: quasiparticle planes, phonon planes and the guard statistics of the INTERMEDIATE state must equal the
    two-call sequence exactly (same arithmetic, operation for operation), on a masked grid with a ragged last block."""
    from qpsim_amd import tables as T
    from qpsim_amd.engine import CompiledGeometry, Engine, link_flags
    en_r, en_s, upd = combo
    rng = np.random.default_rng(ne + 5)
    mask = rng.random((19, 41)) > 0.2
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z))
    n = int(mask.sum())
    E, dE = T.build_energy_grid(180.0, 1.0, 3.0, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = T.dynes_density_of_states(E, 180.0, 0.1)
    kr, ks = T.recombination_kernel_base(E, 180.0, 500.0, 1.2), T.scattering_kernel_base(E, 180.0, 400.0, 1.2)
    state = rng.random((ne, n)) * rho[:, None] * rng.choice([1e-5, 1e-2, 0.5, 0.9], size=n)[None, :]
    ph = T.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, n)))
    tab = eng.make_collision_tables(kr[None], ks[None], rho[None], idx_d, idx_s, sg)
    assert tab["kernel"] == "register" and tab["pair"]
    dt_a, dt_b, gen = 0.05, 0.03, 2.5e-7
    # two calls
    s0, p_two = eng.upload_packed(state), eng.upload_packed(ph)
    s1, s2 = eng.empty(ne, eng.ncell), eng.empty(ne, eng.ncell)
    stats_two = eng.pauli_stats_result(eng.collide_guarded(tab, s0, s1, p_two, dE, dt_a, en_r, en_s, upd, 1e-18))
    eng.add_constant(s1, gen)
    eng.collide(tab, s1, s2, p_two, dE, dt_b, en_r, en_s, upd)
    # one pass
    t0, p_one = eng.upload_packed(state), eng.upload_packed(ph)
    t2 = eng.empty(ne, eng.ncell)
    t2.fill_(-3.0)
    stats_one = eng.pauli_stats_result(eng.collide_pair_guarded(tab, t0, t2, p_one, dE, dt_a, dt_b, gen, en_r, en_s, upd, 1e-18))
    assert stats_one == stats_two
    assert np.array_equal(t2.cpu().numpy(), s2.cpu().numpy())
    assert np.array_equal(p_one.cpu().numpy(), p_two.cpu().numpy())
    if not upd:
        assert np.array_equal(eng.download_packed(p_one), ph)


@pytest.mark.parametrize("gen", ["none", "constant", "pulse"])
def test_run_with_fused_half_steps_equals_the_run_without(monkeypatch, gen):
    """`run_2d_crank_nicolson`, NE = 12 full physics, stores every 3rd step, a short remainder step, generation that
    switches on and off inside the run: with the double half-step passes between store points (default) and with every
    half-step as its own call (QPSIM_COLL_PAIR=0) the outputs are identical - frames, energy frames, masses, phonon
    history, and the warning the guard raises."""
    import warnings
    from qpsim_amd.models import ExternalGenerationSpec
    from qpsim_amd.solver import run_2d_crank_nicolson
    mask, edges, bcs = _rect_problem(20, 28)
    init = 1e-3 * (1.0 + np.random.default_rng(2).random(mask.shape))
    spec = {"none": None, "constant": ExternalGenerationSpec(mode="constant", rate=3e-6),
            "pulse": ExternalGenerationSpec(mode="pulse", pulse_start=0.25, pulse_duration=0.3, pulse_rate=4e-5)}[gen]
    kw = dict(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
              total_time=1.05, dx=1.0, store_every=3, energy_gap=180.0, energy_min_factor=1.0, energy_max_factor=3.0,
              num_energy_bins=12, enable_diffusion=True, enable_recombination=True, enable_scattering=True,
              external_generation=spec, diffusion_scheme="adi", pauli_warn_threshold=None)
    outs = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("QPSIM_COLL_PAIR", pair)
        hist = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = run_2d_crank_nicolson(**kw, phonon_history_out=hist)
        outs[pair] = (res, hist)
    a, b = outs["1"], outs["0"]
    assert a[0][0] == b[0][0] and a[0][2] == b[0][2] and len(a[0][0]) == 5     # t = 0, 0.3, 0.6, 0.9, 1.05
    for fa, fb in zip(a[0][1], b[0][1]):
        assert np.array_equal(fa, fb, equal_nan=True)
    for ea, eb in zip(a[0][4], b[0][4]):
        assert np.array_equal(np.stack(ea), np.stack(eb), equal_nan=True)
    for pa, pb in zip(a[1]["phonon_energy_frames"], b[1]["phonon_energy_frames"]):
        assert np.array_equal(np.stack(pa), np.stack(pb), equal_nan=True)
