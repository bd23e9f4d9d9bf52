"""Domain-decomposed ADI on the GPU: several virtual ranks on one device (lock-step), and two real processes sharing
the device with gloo as the transport (the RCCL transport differs only in who moves the rows)."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _setup(gny, gnx):
    from test_distributed_cpu import _problem
    return _problem(gny, gnx)


def _oracle(mask, edges, bcs, dx, dt, D, u0, nsteps):
    from test_distributed_cpu import _oracle_steps
    return _oracle_steps(mask, edges, bcs, dx, dt, D, u0, nsteps)


@pytest.mark.parametrize("py,px,gny,gnx", [(1, 2, 64, 192), (2, 1, 192, 70), (2, 2, 128, 240), (2, 4, 256, 512), (1, 3, 5, 256)])
def test_hip_blocks_in_lockstep_match_global_adi(py, px, gny, gnx):
    from qpsim_amd.distributed import BlockTopology, HipBlockBackend, lockstep_adi_steps
    mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(gny, gnx)
    topos = [BlockTopology(gny, gnx, py, px, r) for r in range(py * px)]
    blocks = [HipBlockBackend(t, dx, dt, D, bc_diag, bc_src) for t in topos]
    for nsteps in (1, 3):
        for b in blocks:
            b.set_field(u0)
        lockstep_adi_steps(blocks, topos, nsteps)
        got = np.zeros_like(u0)
        for b, t in zip(blocks, topos):
            j0, i0, ny, nx = t.block
            got[:, j0:j0 + ny, i0:i0 + nx] = b.get_field()
        want = _oracle(mask, edges, bcs, dx, dt, D, u0, nsteps)
        assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-13


def test_decomposed_plan_rejects_stiff_coefficients_and_unaligned_blocks():
    from qpsim_amd import _hip
    from qpsim_amd.distributed import BlockTopology, HipBlockBackend
    mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(128, 128)
    with pytest.raises(_hip.QPHipError, match="too large for a decomposed grid"):
        HipBlockBackend(BlockTopology(128, 128, 1, 2, 0), dx, dt, [400.0], bc_diag, bc_src)
    # 136 rows cut in two leaves an 8-row block: the coupling through it does not underflow
    with pytest.raises(_hip.QPHipError, match="too large for a decomposed grid"):
        HipBlockBackend(BlockTopology(136, 128, 2, 1, 0), dx, dt, [6.0], bc_diag, bc_src)


def _worker(rank, world, port, py, px, gny, gnx, nsteps, out_dir):
    for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from qpsim_amd.distributed import BlockTopology, HipBlockBackend, TorchDistTransport, block_adi_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(gny, gnx)
        topo = BlockTopology(gny, gnx, py, px, rank)
        be = HipBlockBackend(topo, dx, dt, D, bc_diag, bc_src, device="cuda:0")
        be.set_field(u0)
        block_adi_steps(be, topo, TorchDistTransport(), nsteps)
        np.save(os.path.join(out_dir, f"block_{rank}.npy"), be.get_field())
    finally:
        dist.destroy_process_group()


def test_two_processes_share_the_gpu_and_exchange_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    from qpsim_amd.distributed import BlockTopology
    from test_distributed_cpu import _free_port
    py, px, gny, gnx, nsteps = 1, 2, 192, 256, 3
    mp.spawn(_worker, args=(2, _free_port(), py, px, gny, gnx, nsteps, str(tmp_path)), nprocs=2, join=True)
    mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(gny, gnx)
    got = np.zeros_like(u0)
    for r in range(2):
        j0, i0, ny, nx = BlockTopology(gny, gnx, py, px, r).block
        got[:, j0:j0 + ny, i0:i0 + nx] = np.load(tmp_path / f"block_{r}.npy")
    want = _oracle(mask, edges, bcs, dx, dt, D, u0, nsteps)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-13


@pytest.mark.parametrize("py,px,gny,gnx", [(1, 2, 64, 192), (2, 1, 192, 70), (2, 2, 192, 256), (2, 4, 256, 512)])
def test_hip_overlapped_halo_blocks_match_global_adi(py, px, gny, gnx):
    """Overlapped-halo decomposition on the GPU (virtual ranks in lock-step): ordinary single-GPU plans on block + 64-cell
    halos, halos refreshed every 3 steps, 8 steps in two calls - own cells equal the global ADI solution to rounding."""
    from qpsim_amd.distributed import BlockTopology, HipOverlapBlock, lockstep_overlap_steps
    mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(gny, gnx)
    topos = [BlockTopology(gny, gnx, py, px, r) for r in range(py * px)]
    blocks = [HipOverlapBlock(t, dx, dt, D, bc_diag, bc_src, steps_per_exchange=3) for t in topos]
    assert all(b.steps_per_exchange == 3 for b in blocks)
    for b in blocks:
        b.set_field(u0)
    total = 0
    for nsteps in (2, 6):
        lockstep_overlap_steps(blocks, nsteps)
        total += nsteps
        got = np.zeros_like(u0)
        for b, t in zip(blocks, topos):
            j0, i0, ny, nx = t.block
            got[:, j0:j0 + ny, i0:i0 + nx] = b.get_field()
        want = _oracle(mask, edges, bcs, dx, dt, D, u0, total)
        assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-13, total


def test_overlapped_halo_scheme_refuses_stiff_steps():
    from qpsim_amd.distributed import BlockTopology, HipOverlapBlock
    mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(128, 128)
    with pytest.raises(ValueError, match="too stiff"):
        HipOverlapBlock(BlockTopology(128, 128, 1, 2, 0), dx, dt, [400.0], bc_diag, bc_src)
    with pytest.raises(ValueError, match="smaller than the halo"):
        HipOverlapBlock(BlockTopology(96, 128, 2, 1, 1), dx, dt, [6.0], bc_diag, bc_src)


def _overlap_worker(rank, world, port, py, px, gny, gnx, nsteps, out_dir):
    for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from qpsim_amd.distributed import BlockTopology, HipOverlapBlock, TorchDistTransport, overlap_adi_steps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(gny, gnx)
        topo = BlockTopology(gny, gnx, py, px, rank)
        blk = HipOverlapBlock(topo, dx, dt, D, bc_diag, bc_src, device="cuda:0", steps_per_exchange=2)
        blk.set_field(u0)
        overlap_adi_steps(blk, TorchDistTransport(), nsteps)
        np.save(os.path.join(out_dir, f"block_{rank}.npy"), blk.get_field())
    finally:
        dist.destroy_process_group()


def test_two_processes_refresh_halos_over_gloo(tmp_path):
    """Two real processes on the one GPU, gloo as the transport (device strips staged through the host): the call sequence
    every rank of the RCCL run executes."""
    import torch.multiprocessing as mp
    from qpsim_amd.distributed import BlockTopology
    from test_distributed_cpu import _free_port
    py, px, gny, gnx, nsteps = 2, 1, 192, 160, 5
    mp.spawn(_overlap_worker, args=(2, _free_port(), py, px, gny, gnx, nsteps, str(tmp_path)), nprocs=2, join=True)
    mask, edges, bcs, dx, dt, D, bc_diag, bc_src, u0 = _setup(gny, gnx)
    got = np.zeros_like(u0)
    for r in range(2):
        j0, i0, ny, nx = BlockTopology(gny, gnx, py, px, r).block
        got[:, j0:j0 + ny, i0:i0 + nx] = np.load(tmp_path / f"block_{r}.npy")
    want = _oracle(mask, edges, bcs, dx, dt, D, u0, nsteps)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-13


def test_coupled_step_on_a_decomposed_grid_matches_the_single_domain_run():
    """north_star's decomposed time-stepper includes the source term: collision half-steps + ADI on 2 x 2 overlapped-halo
    blocks (virtual ranks, halos of the quasiparticle planes refreshed every 2 steps) against the same coupled steps on the
    undecomposed 256 x 256 grid."""
    import torch
    from qpsim_amd.bench_workloads import CoupledWorkload, OverlapDecomposedWorkload, _global_field
    from qpsim_amd.distributed import BlockTopology, lockstep_overlap_steps
    N, steps = 256, 5
    dev = torch.device("cuda", torch.cuda.current_device())
    ranks = [OverlapDecomposedWorkload(N, dev, coupled=True, steps_per_exchange=4, topo=BlockTopology(N, N, 2, 2, r))
             for r in range(4)]
    assert all(w.block.steps_per_exchange == 2 for w in ranks) and ranks[0].block.ey == 192
    whole = CoupledWorkload(N, dev, init_occupation=_global_field(torch, 0, 0, N, N, dev).reshape(-1))
    lockstep_overlap_steps([w.block for w in ranks], steps)
    whole.run(steps)
    torch.cuda.synchronize()
    ref = whole.state.view(whole.ne, N, N)
    scale = float(ref.abs().max())
    for w in ranks:
        j0, i0, ny, nx = w.topo.block
        got = w.block.own
        err = float((got - ref[:, j0:j0 + ny, i0:i0 + nx]).abs().max()) / scale
        assert err < 1e-13, (w.topo.rank, err)
    assert ranks[0].inner.max_occ > 0.0


@pytest.mark.parametrize("nfield,ey,ex", [(1, 192, 256), (3, 130, 71), (2, 64, 64)])
def test_halo_pack_kernel_moves_every_window_both_ways(nfield, ey, ex):
    """`qp_halo_pack` (all windows of a refresh in one launch; 16-byte path for even geometry, 8-byte path otherwise) against
    the slice copies of the device-agnostic `OverlapBlock.pack / unpack`."""
    import ctypes as C
    import torch
    from qpsim_amd import _hip
    lib = _hip.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(ey + ex)
    rects = [(0, 0, 7, 9), (ey - 5, ex - 11, 5, 11), (3, 2, ey - 6, 4), (1, ex - 8, 2, 8), (ey // 2, 0, 1, ex),
             (0, ex // 2, ey, 1), (8, 8, 16, 16), (ey - 1, 0, 1, 1)]
    u = torch.as_tensor(rng.random((nfield, ey, ex)), device=dev)
    total = sum(nfield * r[2] * r[3] for r in rects)
    buf = torch.full((total,), -1.0, dtype=torch.float64, device=dev)
    arr = (C.c_int32 * (4 * len(rects)))(*[v for r in rects for v in r])
    stream = int(torch.cuda.current_stream(dev).cuda_stream)
    _hip.check(lib.qp_halo_pack(int(u.data_ptr()), nfield, ey, ex, arr, len(rects), 0, int(buf.data_ptr()), stream), "pack")
    want = torch.cat([u[:, r:r + nr, c:c + nc].reshape(-1) for r, c, nr, nc in rects])
    assert torch.equal(buf, want)
    fresh = torch.as_tensor(rng.random(total), device=dev)
    target = u.clone()
    _hip.check(lib.qp_halo_pack(int(target.data_ptr()), nfield, ey, ex, arr, len(rects), 1, int(fresh.data_ptr()), stream),
               "unpack")
    ref, off = u.clone(), 0
    for r, c, nr, nc in rects:              # later windows overwrite earlier ones where they overlap, in window order
        n = nfield * nr * nc
        ref[:, r:r + nr, c:c + nc] = fresh[off:off + n].view(nfield, nr, nc)
        off += n
    # overlapping windows are written by different blocks of one launch: compare where exactly one window covers a cell
    cover = torch.zeros((ey, ex), dtype=torch.int32, device=dev)
    for r, c, nr, nc in rects:
        cover[r:r + nr, c:c + nc] += 1
    once = (cover <= 1)[None].expand_as(ref)
    assert torch.equal(target[once], ref[once])
    with pytest.raises(_hip.QPHipError, match="window outside the block"):
        bad = (C.c_int32 * 4)(0, 0, ey + 1, 1)
        _hip.check(lib.qp_halo_pack(int(u.data_ptr()), nfield, ey, ex, bad, 1, 0, int(buf.data_ptr()), stream), "pack")


def _coupled_2x2_worker(rank, world, port, N, steps, out_dir):
    for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from qpsim_amd.bench_workloads import OverlapDecomposedWorkload
    from qpsim_amd.distributed import measure_refresh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        w = OverlapDecomposedWorkload(N, dev, coupled=True, steps_per_exchange=4)
        assert (w.topo.py, w.topo.px) == (2, 2) and w.block.steps_per_exchange == 2 and len(w.block.windows()) == 3
        w.run(steps)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, f"own_{rank}.npy"), w.block.own.cpu().numpy())
        st = measure_refresh(w.block, w.transport, reps=2, sync=torch.cuda.synchronize)
        assert st["refresh_us"] > 0
    finally:
        dist.destroy_process_group()


def test_four_processes_2x2_coupled_step_over_gloo(tmp_path):
    """Four real processes on the one GPU (gloo transport, strips staged through the host), 2 x 2 blocks, the COUPLED step
    (collision half-steps + ADI, NE = 12) with the one-round refresh (two side strips + one corner block per rank, packed
    by `qp_halo_pack`) - own cells equal to the undecomposed run of the same steps."""
    import torch
    import torch.multiprocessing as mp
    from qpsim_amd.bench_workloads import CoupledWorkload, _global_field
    from qpsim_amd.distributed import BlockTopology
    from test_distributed_cpu import _free_port
    N, steps = 256, 5
    mp.spawn(_coupled_2x2_worker, args=(4, _free_port(), N, steps, str(tmp_path)), nprocs=4, join=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    whole = CoupledWorkload(N, dev, init_occupation=_global_field(torch, 0, 0, N, N, dev).reshape(-1))
    whole.run(steps)
    torch.cuda.synchronize()
    ref = whole.state.view(whole.ne, N, N).cpu().numpy()
    for r in range(4):
        j0, i0, ny, nx = BlockTopology(N, N, 2, 2, r).block
        got = np.load(tmp_path / f"own_{r}.npy")
        assert np.max(np.abs(got - ref[:, j0:j0 + ny, i0:i0 + nx])) / np.max(np.abs(ref)) < 1e-13, r


def test_halo_entry_points_refuse_plans_without_interface_rows():
    """`qp_adi_rect_iface_halo` / `qp_adi_rect_set_field_halo` copy nfield x nlines doubles into / out of arrays
    that only block plans of a decomposed grid own (a Peaceman-Rachford plan on fine tiles allocates one element): any other
    plan must be refused, not read out of bounds."""
    import ctypes as C
    import torch
    from qpsim_amd import _hip
    from qpsim_amd.engine import RectPlan
    lib = _hip.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    buf = torch.zeros(2 * 128, dtype=torch.float64, device=dev)
    stream = int(torch.cuda.current_stream(dev).cuda_stream)
    whole = RectPlan(lib, 128, 128, 2, 0.05, [6.0, 1.0], [0.0] * 4, [0.0] * 4)
    pr = RectPlan.peaceman_rachford(lib, 128, 128, 2, 0.05, [6.0, 1.0], [0.0] * 4, 1.0)
    for plan in (whole, pr):
        assert lib.qp_adi_rect_iface_halo(plan.handle, 0, 0, 0, int(buf.data_ptr()), stream) == -1
        assert b"decomposed grid" in lib.qp_last_error()
        assert lib.qp_adi_rect_set_field_halo(plan.handle, 0, int(buf.data_ptr()), stream) == -1
    block = RectPlan(lib, 128, 128, 2, 0.05, [6.0, 1.0], [0.0] * 4, [0.0] * 4, block=(128, 256, 0, 0))
    assert lib.qp_adi_rect_iface_halo(block.handle, 0, 1, 0, int(buf.data_ptr()), stream) == 0
    torch.cuda.synchronize()
