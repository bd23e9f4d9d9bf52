"""Fine tiles of the rectangle ADI path (32-cell chunks, `csrc/qp_adi_fine.inc`) against the oracle and the 64 x 64 tiles.

The reference path is `qpsim/solver.py:1545-1566` (the scalar diffusion loop) with the operators of `solver.py:174-236`;
the oracle restates the ADI step (`oracle/qp_oracle.py: ADIStepper`) and the exact CN step (SuperLU).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


@pytest.fixture(scope="module")
def O():
    from oracle import qp_oracle
    return qp_oracle


def _problem(ny, nx, dx=0.9):
    from qpsim_amd.engine import Engine, compile_geometry
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    mask = np.ones((ny, nx), dtype=bool)
    edges = extract_edge_segments(mask)
    side_bc = {"left": BoundaryCondition("dirichlet", 0.7), "right": BoundaryCondition("robin", 0.4, 0.2),
               "up": BoundaryCondition("neumann", -0.3), "down": BoundaryCondition("absorbing")}
    bcs = {e.edge_id: side_bc[e.normal] for e in edges}
    return mask, edges, bcs, Engine(compile_geometry(mask, edges, bcs, dx))


@pytest.mark.parametrize("ny,nx", [(64, 64), (128, 192), (256, 128), (64, 320), (192, 64)])
def test_fine_tiles_match_oracle_adi_and_coarse_tiles(O, monkeypatch, ny, nx):
    """Every pass of the fine kernels (entry, x, carry, exit; first / interior / last chunks in both directions, lines of
    2 ... 10 chunks) vs the oracle ADI step and vs the 64 x 64 tiles on the same plan parameters."""
    from qpsim_amd.engine import DiffusionOperator
    dx, dt = 0.9, 0.11                      # r = 0.0679
    mask, edges, bcs, eng = _problem(ny, nx, dx)
    ops = O.build_grid_ops(mask, edges, bcs, dx)
    Dc = [4.4, 0.35, 0.0, 2.0]              # a = r D up to 0.30: chunks of 32 cells still decouple (far coupling 1.7e-23)
    rng = np.random.default_rng(ny * 7 + nx)
    u0 = rng.random((len(Dc), ny * nx))
    monkeypatch.setenv("QPSIM_FINE_TILES", "1")
    fine = DiffusionOperator(eng, len(Dc), dt, dcoef=Dc)
    monkeypatch.setenv("QPSIM_FINE_TILES", "0")
    coarse = DiffusionOperator(eng, len(Dc), dt, dcoef=Dc)
    assert fine.rect is not None and fine.rect.fine and not coarse.rect.fine
    for nsteps in (1, 2, 5):
        a, b = eng.upload_packed(u0), eng.upload_packed(u0)
        eng.adi_steps(fine, a, nsteps)
        eng.adi_steps(coarse, b, nsteps)
        ha, hb = eng.download_packed(a), eng.download_packed(b)
        assert rel_err(ha, hb) < 5e-14, nsteps
        for k, D in enumerate(Dc):
            st = O.ADIStepper(ops, D, dt)
            want = u0[k].copy()
            for _ in range(nsteps):
                want = st.step(want)
            assert rel_err(ha[k], want) < 2e-13, (k, nsteps)


def test_fine_tiles_are_refused_where_32_cell_chunks_do_not_decouple(monkeypatch):
    """r D = 0.41: the far coupling of a 32-cell chunk (6.5e-21) is above the 1e-22 drop threshold -> 64 x 64 tiles;
    extents that are not multiples of 64 and decomposed blocks never get fine tiles."""
    from qpsim_amd.engine import DiffusionOperator
    monkeypatch.setenv("QPSIM_FINE_TILES", "1")
    _, _, _, eng = _problem(128, 128)
    assert not DiffusionOperator(eng, 2, 0.11, dcoef=[1.0, 6.0]).rect.fine
    assert DiffusionOperator(eng, 2, 0.11, dcoef=[1.0, 4.0]).rect.fine
    _, _, _, eng2 = _problem(128, 96)
    assert not DiffusionOperator(eng2, 1, 0.11, dcoef=[1.0]).rect.fine


@pytest.mark.parametrize("ny,nx,D", [(128, 192, 4.0), (64, 64, 1.0)])
@pytest.mark.parametrize("pr", ["1", "0"])
def test_exact_cn_step_on_fine_tiles_matches_superlu(O, monkeypatch, ny, nx, D, pr):
    """Default (unsplit CN) step vs the oracle's SuperLU solve (`solver.py:231,1155-1161`), Dirichlet / Robin / Neumann /
    absorbing sides: pr = 1 the Peaceman-Rachford cycle (`qp_adi_rect_pr_iteration`, source-plane passes of the fine
    kernels), pr = 0 the ADI-preconditioned Chebyshev iteration whose preconditioner runs `qp_adi_rect_solve` on fine tiles
    (reduce pass, plain x-solve, exit pass)."""
    from qpsim_amd.engine import DiffusionOperator, _pr_cycle
    dx, dt = 0.9, 0.11
    mask, edges, bcs, eng = _problem(ny, nx, dx)
    ops = O.build_grid_ops(mask, edges, bcs, dx)
    monkeypatch.setenv("QPSIM_FINE_TILES", "1")
    monkeypatch.setenv("QPSIM_CN_PR", pr)
    nf = 3
    Dc = [D, 0.3 * D, 0.0]
    op = DiffusionOperator(eng, nf, dt, dcoef=Dc)
    assert op.rect.fine
    cycle = _pr_cycle(op, 8)
    assert (cycle is not None) == (pr == "1")
    rng = np.random.default_rng(5)
    u0 = rng.random((nf, ny * nx))
    a = eng.upload_packed(u0)
    its = eng.cn_exact_step(op, a)      # rough data + boundary sources: the first cycle may need polishing (count returned)
    assert its >= 1
    got = eng.download_packed(a)
    for k, Dk in enumerate(Dc):
        want = O.CNStepper(ops, Dk, dt).step(u0[k])
        assert rel_err(got[k], want) < 1e-11, k
    # several steps: the cycle length adapts to the (now smoother) data and the result stays on the reference
    counts = [eng.cn_exact_step(op, a) for _ in range(4)]
    if pr == "1":       # the cycle length has adapted: the last steps are served by one cycle alone
        assert 4 <= counts[-1] <= 12 and abs(counts[-1] - op._pr_J) <= 1, counts
    want = u0[0].copy()
    st = O.CNStepper(ops, Dc[0], dt)
    for _ in range(5):
        want = st.step(want)
    assert rel_err(eng.download_packed(a)[0], want) < 1e-11


@pytest.mark.parametrize("ny,nx,D", [(70, 130, 2.0), (128, 192, 40.0), (1, 100, 6.0), (129, 257, 6.0), (64, 63, 0.5)])
def test_peaceman_rachford_cycle_on_64_tiles_matches_superlu(O, ny, nx, D):
    """The cycle on the 64 x 64 kernels (`rect_*_kernel<..., SRC>`): ragged extents, remainder chunks, the banded reduced
    solve of stiff steps (r D = 2.7), one-cell-thick grids - against the oracle's SuperLU solve, three steps."""
    from qpsim_amd.engine import DiffusionOperator, _pr_cycle
    dx, dt = 0.9, 0.11
    mask, edges, bcs, eng = _problem(ny, nx, dx)
    ops = O.build_grid_ops(mask, edges, bcs, dx)
    op = DiffusionOperator(eng, 2, dt, dcoef=[D, 0.4 * D])
    cycle = _pr_cycle(op, 8)
    assert cycle is not None and not all(p.fine for p in cycle)      # (stiff: the large-p plans of the cycle do qualify)
    rng = np.random.default_rng(ny + nx)
    u0 = rng.random((2, ny * nx))
    a = eng.upload_packed(u0)
    for _ in range(3):
        eng.cn_exact_step(op, a)
    got = eng.download_packed(a)
    for k, Dk in enumerate([D, 0.4 * D]):
        st = O.CNStepper(ops, Dk, dt)
        want = u0[k].copy()
        for _ in range(3):
            want = st.step(want)
        assert rel_err(got[k], want) < 2e-11, k


def test_peaceman_rachford_cycle_is_refused_where_it_does_not_apply():
    """Masked grids (Lx, Ly do not commute) and per-cell diffusivities have no cycle: the preconditioned iteration runs."""
    from qpsim_amd.engine import DiffusionOperator, Engine, _pr_cycle, compile_geometry
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    _, _, _, eng = _problem(128, 128)
    assert _pr_cycle(DiffusionOperator(eng, 1, 0.11, dcoef=[1.0]), 8) is not None
    assert _pr_cycle(DiffusionOperator(eng, 1, 0.11, dfield=np.full((1, 128 * 128), 1.0)), 8) is None
    mask = np.ones((96, 96), dtype=bool)
    mask[30:50, 40:70] = False
    edges = extract_edge_segments(mask)
    eng2 = Engine(compile_geometry(mask, edges, {e.edge_id: BoundaryCondition("reflective") for e in edges}, 1.0))
    assert _pr_cycle(DiffusionOperator(eng2, 1, 0.11, dcoef=[1.0]), 8) is None


def test_fine_tiles_large_grid_roundtrip_properties(monkeypatch):
    """2048^2 (65536 fine tiles, both stream-mode-0 kernels): fine vs 64 x 64 tiles on the same field, and conservation of
    the total under reflective walls."""
    import torch
    from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    N = 2048
    mask = np.ones((N, N), dtype=bool)
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("neumann", 0.0) for e in edges}
    eng = Engine(compile_geometry(mask, edges, bcs, 1.0))
    monkeypatch.setenv("QPSIM_FINE_TILES", "1")
    fine = DiffusionOperator(eng, 1, 0.6, dcoef=[1.0])
    monkeypatch.setenv("QPSIM_FINE_TILES", "0")
    coarse = DiffusionOperator(eng, 1, 0.6, dcoef=[1.0])
    assert fine.rect.fine and not coarse.rect.fine
    g = torch.Generator(device="cpu").manual_seed(3)
    u0 = torch.rand(1, N * N, generator=g, dtype=torch.float64).cuda()
    a, b = u0.clone(), u0.clone()
    eng.adi_steps(fine, a, 7)
    eng.adi_steps(coarse, b, 7)
    assert float((a - b).abs().max() / b.abs().max()) < 5e-14
    assert abs(float(a.sum() / u0.sum()) - 1.0) < 1e-12


@pytest.mark.parametrize("seed", range(8))
def test_fine_tiles_fuzz_against_coarse_tiles_and_general_kernels(monkeypatch, seed):
    """Random extents (multiples of 64), boundary kinds / values per side, diffusivities and step counts: the fine tiles,
    the 64 x 64 tiles and the per-line general kernels (a different algorithm: one Thomas solve per grid line) agree."""
    from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    rng = np.random.default_rng(1000 + seed)
    ny, nx = (int(64 * rng.integers(1, 6)) for _ in range(2))
    mask = np.ones((ny, nx), dtype=bool)
    edges = extract_edge_segments(mask)

    def bc():
        kind = ["dirichlet", "neumann", "robin", "absorbing", "reflective"][int(rng.integers(0, 5))]
        if kind == "robin":
            return BoundaryCondition("robin", float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.05, 1.0)))
        if kind in ("dirichlet", "neumann"):
            return BoundaryCondition(kind, float(rng.uniform(-0.5, 0.9)))
        return BoundaryCondition(kind)

    side_bc = {side: bc() for side in ("left", "right", "up", "down")}
    bcs = {e.edge_id: side_bc[e.normal] for e in edges}
    dx, dt = float(rng.uniform(0.7, 1.3)), float(rng.uniform(0.05, 0.15))
    r = 0.5 * dt / dx ** 2
    nf = int(rng.integers(1, 5))
    Dc = [float(v) for v in rng.uniform(0.0, 0.31 / r, nf)]          # r D < 0.31: fine tiles qualify
    if seed % 3 == 0:
        Dc[0] = 0.0
    eng = Engine(compile_geometry(mask, edges, bcs, dx))
    monkeypatch.setenv("QPSIM_FINE_TILES", "1")
    fine = DiffusionOperator(eng, nf, dt, dcoef=Dc)
    monkeypatch.setenv("QPSIM_FINE_TILES", "0")
    coarse = DiffusionOperator(eng, nf, dt, dcoef=Dc)
    slow = DiffusionOperator(eng, nf, dt, dcoef=Dc, allow_fast=False)
    assert fine.rect.fine and not coarse.rect.fine and slow.rect is None
    u0 = rng.random((nf, ny * nx))
    nsteps = int(rng.integers(1, 7))
    a, b, c = (eng.upload_packed(u0) for _ in range(3))
    eng.adi_steps(fine, a, nsteps)
    eng.adi_steps(coarse, b, nsteps)
    eng.adi_steps(slow, c, nsteps)
    ha, hb, hc = (eng.download_packed(t) for t in (a, b, c))
    assert rel_err(ha, hb) < 5e-14, (ny, nx, side_bc, Dc, nsteps)
    assert rel_err(ha, hc) < 5e-13, (ny, nx, side_bc, Dc, nsteps)


def test_carried_and_three_pass_peaceman_rachford_cycles_agree(monkeypatch):
    """`qp_adi_rect_pr_cycle`: the carried form (the y-pass of iteration j leaves the right-hand side of iteration j + 1,
    `fine_y_next_kernel`) against one three-pass `qp_adi_rect_pr_iteration` per parameter, same plans, same data."""
    import ctypes as C
    from qpsim_amd import _hip
    from qpsim_amd.engine import DiffusionOperator, _pr_cycle, _ptr
    _, _, _, eng = _problem(192, 256)
    op = DiffusionOperator(eng, 3, 0.11, dcoef=[4.0, 1.0, 0.0])
    cycle = _pr_cycle(op, 6)
    assert cycle is not None and all(p.fine for p in cycle) and len(cycle) >= 4
    rng = np.random.default_rng(11)
    u0, b0 = rng.random((3, 192 * 256)), rng.random((3, 192 * 256))
    handles = (C.POINTER(_hip.RectPlan) * len(cycle))(*[p.handle for p in cycle])
    out = []
    for carried in ("1", "0"):
        monkeypatch.setenv("QPSIM_PR_CARRIED", carried)
        u, b = eng.upload_packed(u0), eng.upload_packed(b0)
        _hip.check(eng.lib.qp_adi_rect_pr_cycle(handles, len(cycle), _ptr(u), _ptr(b), eng.stream), "qp_adi_rect_pr_cycle")
        out.append(eng.download_packed(u))
    assert rel_err(out[0], out[1]) < 1e-13
    # and the cycle does what it promises: A u = b to the cycle's reduction (A = I - r D L, checked with the general stencil)
    u = eng.upload_packed(out[0])
    res = eng.upload_packed(np.zeros_like(u0))
    eng.stencil(op, u, res, 1.0, -1.0, -1.0, 0.0)      # res = u - a Lx u - a Ly u = A u (sources are part of b)
    assert rel_err(eng.download_packed(res), b0) < 1e-7


def test_default_scheme_at_full_size_solves_the_unsplit_system():
    """4096^2 (BASELINE's headline grid), default unsplit-CN step through the carried Peaceman-Rachford cycle: the result
    satisfies (I - rL) u' = (I + rL) u to 1e-12 as evaluated by the GENERAL stencil kernel (per-cell geometry arrays - code
    that shares nothing with the cycle or with `qp_adi_rect_combine`), conserves the total under reflective walls and obeys
    the maximum principle.  Size-independent properties: no oracle reaches this size."""
    import torch
    from qpsim_amd.engine import DiffusionOperator, Engine, _pr_cycle, compile_geometry
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    N = 4096
    mask = np.ones((N, N), dtype=bool)
    edges = extract_edge_segments(mask)
    eng = Engine(compile_geometry(mask, edges, {e.edge_id: BoundaryCondition("reflective") for e in edges}, 1.0))
    op = DiffusionOperator(eng, 1, 0.1, dcoef=[6.0])
    gen = DiffusionOperator(eng, 1, 0.1, dcoef=[6.0], allow_fast=False)
    assert op.rect.fine and gen.rect is None and gen.tile is None
    cycle = _pr_cycle(op, 8)
    assert cycle is not None and all(p.fine for p in cycle)
    g = torch.Generator(device="cpu").manual_seed(7)
    u0 = (1e-4 * (1.0 + torch.rand(1, N * N, generator=g, dtype=torch.float64))).cuda()
    u = u0.clone()
    eng.cn_exact_step(op, u)
    lhs, rhs = torch.empty_like(u), torch.empty_like(u)
    eng.stencil(gen, u, lhs, 1.0, -1.0, -1.0, 0.0)       # (I - rL) u'
    eng.stencil(gen, u0, rhs, 1.0, 1.0, 1.0, 2.0)        # (I + rL) u + 2 r S
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) < 1e-12
    assert abs(float(u.sum() / u0.sum()) - 1.0) < 1e-12
    assert float(u.min()) >= float(u0.min()) and float(u.max()) <= float(u0.max())
