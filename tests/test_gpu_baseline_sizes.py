"""The BASELINE.json configurations AT THEIR SIZES against the oracle (the CPU restatement of the reference, pinned by the
recorded reference runs - `tests/test_oracle_parity.py`), not through size-independent properties only:

  configs[*] headline   4096 x 4096, ONE field, default plan = the benchmarked `fine_*_kernel<..., 0, false>` instantiation
  configs[1]            1024 x 1024, NE = 12, recombination, frozen phonons: one full Strang step C(dt/2) D(dt) C(dt/2)
  configs[3]            64 members x 256 x 256 batched (the per-GPU share of the 512-member ensemble)
  configs[4]            8192 x 8192 cut 2 x 4 into overlapped-halo blocks of 4160 x 2176 (8 virtual ranks on the one GPU)

Reference path of the diffusion step: `/root/reference/qpsim/solver.py:1545-1555` (scalar loop) and `:1428-1452` (per
bin); Strang order `:1469-1475`; collision update `:703-791`.  The oracle's NumPy ADI needs ~1-10 s per 4096^2 step."""
from __future__ import annotations

import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIDE_KINDS = {
    "reflective": None,
    "mixed": {"left": ("dirichlet", 2e-4, None), "right": ("robin", 0.4, 1e-4), "up": ("neumann", -3e-5, None),
              "down": ("absorbing", None, None)},
}


@pytest.fixture(scope="module")
def O():
    from oracle import qp_oracle
    return qp_oracle


def _rect(ny, nx, sides):
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    mask = np.ones((ny, nx), dtype=bool)
    edges = extract_edge_segments(mask)
    if sides is None:
        bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    else:
        bcs = {e.edge_id: BoundaryCondition(*[v for v in sides[e.normal] if v is not None]) for e in edges}
    return mask, edges, bcs


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


@pytest.mark.parametrize("sides", ["reflective", "mixed"])
def test_headline_4096_single_field_fine_tiles_match_oracle_adi(O, monkeypatch, sides):
    """The exact instantiation `bench.py` times (4096^2, 1 field, D = 6, dt = 0.1, dx = 1: fine tiles, cached stream mode,
    compact tables; k steps = entry pass + carried x / y sweeps + exit pass) against the oracle's ADI on the whole grid."""
    import torch
    from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry
    for knob in ("QPSIM_FINE_TILES", "QPSIM_STREAM_MODE", "QPSIM_COMPACT_TABLES"):
        monkeypatch.delenv(knob, raising=False)
    N, steps = 4096, 2
    mask, edges, bcs = _rect(N, N, SIDE_KINDS[sides])
    eng = Engine(compile_geometry(mask, edges, bcs, 1.0))
    op = DiffusionOperator(eng, 1, 0.1, dcoef=[6.0])
    assert op.rect is not None and op.rect.fine and op.rect.decoupled == (True, True)
    u0 = 1e-4 * (1.0 + np.random.default_rng(0).random((N, N)))          # bench.py's synthetic field
    u = torch.as_tensor(u0.reshape(1, -1), device=eng.device)
    eng.adi_steps(op, u, steps)
    got = u.cpu().numpy().reshape(N, N)
    t0 = time.perf_counter()
    st = O.ADIStepper(O.build_grid_ops(mask, edges, bcs, 1.0), 6.0, 0.1)
    want = u0
    for _ in range(steps):
        want = st.step_grid(want)
    print(f"oracle ADI {N}^2 x {steps} steps: {time.perf_counter() - t0:.1f} s")
    assert _rel(got, want) < 2e-13
    # per-cell as well: a wrong tile (one of 8192 waves) must not hide in the global norm
    assert float(np.max(np.abs(got - want) / np.abs(want))) < 1e-12


def test_config1_full_strang_step_1024_ne12_matches_oracle(O):
    """BASELINE configs[1] (`bench.py --workload c2`): 1024^2, NE = 12, recombination on, scattering off, phonons frozen -
    ONE whole step C(dt/2) D(dt) C(dt/2) + guard exactly as the workload (and `run_2d_crank_nicolson` with
    diffusion_scheme="adi") issues it, against the oracle: per-pixel update on every pixel, NumPy ADI on every bin."""
    import torch
    from qpsim_amd import bench_workloads as W
    from qpsim_amd import tables as T
    dev = torch.device("cuda", torch.cuda.current_device())
    wl = W.build("c2", dev)
    N, ne = 1024, 12
    assert wl.tab["kernel"] == "register" and wl.op.rect is not None and wl.state.shape == (ne, N * N)
    s0, p0 = wl.state.cpu().numpy().copy(), wl.phonon.cpu().numpy().copy()
    wl.run(1)
    torch.cuda.synchronize()
    got, got_p = wl.state.cpu().numpy(), wl.phonon.cpu().numpy()
    assert np.array_equal(got_p, p0)                                         # frozen phonons stay untouched
    gap = 180.0
    E, dE = T.build_energy_grid(gap, 1.0, 3.0, ne)
    om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
    rho = T.dynes_density_of_states(E, gap, 0.0)
    kr = T.recombination_kernel_base(E, gap, 440.0, 1.2)
    tables = {"rho": rho[None], "Kr0": kr[None], "Ks0": None, "cls": np.zeros(N * N, dtype=int), "idx_diff": idx_d,
              "idx_sum": idx_s, "sign": sg, "dE": dE}
    t0 = time.perf_counter()
    want, ph = s0.copy(), p0.copy()
    O.collision_step(want, ph, tables, 0.05, en_r=True, en_s=False, update_phonons=False)
    mask, edges, bcs = _rect(N, N, None)
    ops = O.build_grid_ops(mask, edges, bcs, 1.0)
    for i, D in enumerate(T.diffusion_coefficients(E, gap, 6.0)):
        want[i] = O.ADIStepper(ops, float(D), 0.1).step_grid(want[i].reshape(N, N)).reshape(-1)
    O.collision_step(want, ph, tables, 0.05, en_r=True, en_s=False, update_phonons=False)
    print(f"oracle coupled step {N}^2 NE={ne}: {time.perf_counter() - t0:.1f} s")
    assert _rel(got, want) < 1e-12
    assert float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300))) < 1e-10
    assert 0.0 < wl.max_occ < 1.0


def test_config3_sixty_four_batched_members_equal_members_run_alone(monkeypatch):
    """BASELINE configs[3] at the per-GPU share the benchmark runs (`--workload c4`: 64 members x 256^2, NE = 12, full
    physics): 2 coupled steps of the whole batch against members 0, 21, 42, 63 run one at a time (first, last and two
    inside: plane offsets members x ncell and the member-major tile order of the ADI plan).  The batch carries 768 planes
    (403 MB: streamed regime, 64 x 64 tiles); a member alone would get the fine tiles, which round differently (5e-14) -
    so the lone member is run once on the batch's tile family, where the results must be BIT-equal, and once on its own
    default plan, where they must agree to rounding."""
    import torch
    from qpsim_amd import bench_workloads as W
    from qpsim_amd.bench_workloads import CoupledWorkload
    dev = torch.device("cuda", torch.cuda.current_device())
    batch = W.build("c4", dev)
    members, N, steps = 64, 256, 2
    ncell = N * N
    assert batch.members == members and batch.state.shape == (12, members * ncell)
    init_s, init_p = batch.state.clone(), batch.phonon.clone()
    init_p *= 1.0 + 0.1 * torch.rand(init_p.shape, dtype=torch.float64, device=dev,
                                     generator=torch.Generator(device=dev).manual_seed(3))
    batch.phonon.copy_(init_p)
    batch.run(steps)
    torch.cuda.synchronize()
    own_family = CoupledWorkload(N, dev, members=1)
    monkeypatch.setenv("QPSIM_FINE_TILES", "1" if batch.op.rect.fine else "0")
    single = CoupledWorkload(N, dev, members=1)
    assert single.op.rect.fine == batch.op.rect.fine
    for m in (0, 21, 42, 63):
        sl = slice(m * ncell, (m + 1) * ncell)
        assert not torch.equal(init_s[:, sl], init_s[:, :ncell]) or m == 0
        for wl in (single, own_family):
            wl.state.copy_(init_s[:, sl])
            wl.phonon.copy_(init_p[:, sl])
            wl.run(steps)
        torch.cuda.synchronize()
        assert torch.equal(single.state, batch.state[:, sl]), m
        assert torch.equal(single.phonon, batch.phonon[:, sl]), m
        for a, b in ((own_family.state, batch.state[:, sl]), (own_family.phonon, batch.phonon[:, sl])):
            assert float((a - b).abs().max() / b.abs().max()) < 2e-13, m


def test_config4_8192_cut_2x4_overlapped_halo_blocks_match_oracle_adi(O):
    """BASELINE configs[4]: 8192^2 on 2 x 4 ranks.  The eight extended blocks (4160 x 2176: the plans the RCCL run builds)
    live on the one GPU as virtual ranks in lock-step; halos refreshed every 3 steps, 4 steps - so one refresh (sides and
    corners, one round) lies inside - and the own cells of every rank are compared with the oracle's ADI on the WHOLE grid.
    Then the production cadence: fresh blocks with the bound's own S (19 steps) run S + 1 steps against the undecomposed
    8192^2 plan on the same GPU (which the first part has just tied to the oracle)."""
    import torch
    from qpsim_amd.distributed import BlockTopology, HipOverlapBlock, lockstep_overlap_steps
    from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry, rect_side_terms
    N, py, px = 8192, 2, 4
    mask, edges, bcs = _rect(N, N, SIDE_KINDS["mixed"])
    geom = compile_geometry(mask, edges, bcs, 1.0)
    bc_diag, bc_src = rect_side_terms(geom)
    u0 = 1e-4 * (1.0 + np.random.default_rng(8).random((1, N, N)))
    topos = [BlockTopology(N, N, py, px, r) for r in range(py * px)]

    def gather(blocks):
        out = np.empty((N, N))
        for b, t in zip(blocks, topos):
            j0, i0, ny, nx = t.block
            out[j0:j0 + ny, i0:i0 + nx] = b.get_field()[0]
        return out

    blocks = [HipOverlapBlock(t, 1.0, 0.1, [6.0], bc_diag, bc_src, steps_per_exchange=3) for t in topos]
    assert {(b.ey, b.ex) for b in blocks} == {(4160, 2112), (4160, 2176)} and all(b.plan.fine for b in blocks)
    assert len(blocks[1].windows()) == 5 and len(blocks[0].windows()) == 3
    for b in blocks:
        b.set_field(u0)
    lockstep_overlap_steps(blocks, 4)
    got = gather(blocks)
    t0 = time.perf_counter()
    st = O.ADIStepper(O.build_grid_ops(mask, edges, bcs, 1.0), 6.0, 0.1)
    want = u0[0]
    for _ in range(4):
        want = st.step_grid(want)
    print(f"oracle ADI {N}^2 x 4 steps: {time.perf_counter() - t0:.1f} s")
    assert _rel(got, want) < 2e-13
    del st, blocks
    # the undecomposed plan on the same grid, same 4 steps (64 x 64 tiles, non-temporal stream mode: another kernel family)
    eng = Engine(geom)
    op = DiffusionOperator(eng, 1, 0.1, dcoef=[6.0])
    whole = torch.as_tensor(u0.reshape(1, -1), device=eng.device)
    eng.adi_steps(op, whole, 4)
    assert _rel(whole.cpu().numpy().reshape(N, N), want) < 2e-13
    # production cadence: S = 19, S + 1 steps -> exactly one refresh, against the undecomposed run
    blocks = [HipOverlapBlock(t, 1.0, 0.1, [6.0], bc_diag, bc_src) for t in topos]
    S = blocks[0].steps_per_exchange
    assert S == 19
    for b in blocks:
        b.set_field(u0)
    lockstep_overlap_steps(blocks, S + 1)
    whole = torch.as_tensor(u0.reshape(1, -1), device=eng.device)
    eng.adi_steps(op, whole, S + 1)
    assert _rel(gather(blocks), whole.cpu().numpy().reshape(N, N)) < 2e-13
