"""Multi-GPU execution: one process per GPU, ``torch.distributed`` (RCCL on ROCm) for the exchanges.

Two sharding modes:

* **Ensembles** -- independent problems (MKID pixels, parameter scans).  Members are dealt round-robin to ranks;
  there is no data-path communication at all (``shard_members``).
* **Domain decomposition** of one large full-rectangle grid into a ``py x px`` grid of blocks.  Collisions are
  pixel-local.  For the ADI sweeps every grid line crosses the blocks of one process row/column; in the
  decoupled-interface regime of the tiled solver (``qp_adi_rect.hip``) the coupling between neighbouring blocks is
  the same 2x2 interface system as between 64-cell chunks inside a block, so each sweep needs exactly one row of
  reduced right-hand sides from each neighbour (``nfield x nlines`` doubles, point-to-point), and a step that starts
  from a materialised field additionally one halo row of the field.  No collective sits on the time loop.

The step sequence is written once (``block_adi_steps``) against two small interfaces -- a *backend* that owns one
block (HIP: ``HipBlockBackend``) and a *transport* that moves rows between neighbours (``TorchDistTransport`` for
real ranks, ``LocalTransport`` for several virtual ranks inside one process) -- so the same orchestration is
exercised by CPU/gloo tests, by single-GPU virtual-rank tests and by the real multi-GPU run.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

TILE = 64


# ------------------------------------------------------------------------------------------------------------------ #
# partitioning
# ------------------------------------------------------------------------------------------------------------------ #
def shard_members(n_members: int, world: int, rank: int) -> list[int]:
    """Ensemble members owned by ``rank`` (round-robin, as SURVEY 8e: member k -> GPU k mod world)."""
    return list(range(rank, n_members, world))


def split_extent(n: int, parts: int) -> list[tuple[int, int]]:
    """Cut ``n`` cells into ``parts`` contiguous (offset, length) blocks whose inner boundaries are multiples of 64."""
    chunks = -(-n // TILE)
    if parts > chunks:
        raise ValueError(f"cannot cut {n} cells ({chunks} chunks of {TILE}) into {parts} blocks")
    base, extra = divmod(chunks, parts)
    out, start = [], 0
    for k in range(parts):
        c = base + (1 if k < extra else 0)
        stop = min(n, (start // TILE + c) * TILE)
        out.append((start, stop - start))
        start = stop
    return out


@dataclass(frozen=True)
class BlockTopology:
    """Position of one rank in the ``py x px`` process grid over a ``gny x gnx`` grid (row-major rank order)."""
    gny: int
    gnx: int
    py: int
    px: int
    rank: int

    @property
    def coords(self) -> tuple[int, int]:
        return divmod(self.rank, self.px)

    @property
    def block(self) -> tuple[int, int, int, int]:
        """(j0, i0, ny, nx) of the local block."""
        ry, rx = self.coords
        j0, ny = split_extent(self.gny, self.py)[ry]
        i0, nx = split_extent(self.gnx, self.px)[rx]
        return j0, i0, ny, nx

    def neighbour_at(self, drow: int, dcol: int) -> int | None:
        """Rank of the block ``drow`` block-rows down and ``dcol`` block-columns right of this one (diagonals included)."""
        ry, rx = self.coords
        ry, rx = ry + drow, rx + dcol
        if (drow or dcol) and 0 <= ry < self.py and 0 <= rx < self.px:
            return ry * self.px + rx
        return None

    def neighbour(self, direction: int, side: int) -> int | None:
        """Rank next to this block: direction 0 = along x (side 0 left, 1 right), 1 = along y (0 up, 1 down)."""
        ry, rx = self.coords
        if direction == 0:
            rx += -1 if side == 0 else 1
        else:
            ry += -1 if side == 0 else 1
        if 0 <= ry < self.py and 0 <= rx < self.px:
            return ry * self.px + rx
        return None


def choose_process_grid(world: int, gny: int, gnx: int) -> tuple[int, int]:
    """Most square ``py x px = world`` factorisation (BASELINE config 5 uses 2 x 4 at 8 GPUs), py <= px."""
    best = (1, world)
    for py in range(1, int(world ** 0.5) + 1):
        if world % py == 0:
            best = (py, world // py)
    return best


# ------------------------------------------------------------------------------------------------------------------ #
# transports
# ------------------------------------------------------------------------------------------------------------------ #
class TorchDistTransport:
    """Neighbour exchange over ``torch.distributed`` point-to-point ops (backend "nccl" = RCCL over xGMI).

    With the gloo backend device tensors are staged through host memory (gloo has no GPU send/recv); that path exists
    for tests on CPU-only or single-GPU machines.
    """

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.stage = dist.get_backend(group) == "gloo"

    def exchange(self, sends: dict[int, "object"], recvs: dict[int, "object"]) -> None:
        """``sends[peer]`` tensors go out, ``recvs[peer]`` tensors are filled; returns when both are complete."""
        dist = self.dist
        ops, staged = [], []
        for peer, t in sends.items():
            buf = t.cpu() if (self.stage and t.is_cuda) else t
            ops.append(dist.P2POp(dist.isend, buf, peer, self.group))
        for peer, t in recvs.items():
            if self.stage and t.is_cuda:
                buf = t.new_empty(t.shape, device="cpu")
                staged.append((t, buf))
            else:
                buf = t
            ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for t, buf in staged:
            t.copy_(buf)


class LocalTransport:
    """In-process mailbox for several virtual ranks driven in lock-step (single-GPU tests of the decomposition)."""

    def __init__(self):
        self.box: dict[tuple[int, int], object] = {}

    def post(self, src: int, sends: dict[int, object]) -> None:
        for peer, t in sends.items():
            self.box[(src, peer)] = t.clone()

    def collect(self, dst: int, recvs: dict[int, object]) -> None:
        for peer, t in recvs.items():
            t.copy_(self.box.pop((peer, dst)))


# ------------------------------------------------------------------------------------------------------------------ #
# the step sequence (backend- and transport-agnostic)
# ------------------------------------------------------------------------------------------------------------------ #
PH_ENTRY, PH_REDUCED_X, PH_SWEEP_X, PH_REDUCED_Y, PH_Y_CARRY, PH_Y_EXIT = range(6)


def block_adi_stages(nsteps: int):
    """The stages of ``nsteps`` consecutive ADI steps on a decomposed grid.

    Each stage is ``(phases, exchange)``: run the phases on the local block, then exchange with the neighbours along
    ``exchange`` = ("field", 1) | ("iface", 0) | ("iface", 1) | None.
    """
    yield (), ("field", 1)                               # halo rows of u for the explicit y-operator
    yield (PH_ENTRY,), ("iface", 0)                      # reduced rhs of the x-solve
    for s in range(nsteps):
        yield (PH_REDUCED_X, PH_SWEEP_X), ("iface", 1)
        if s + 1 < nsteps:
            yield (PH_REDUCED_Y, PH_Y_CARRY), ("iface", 0)
        else:
            yield (PH_REDUCED_Y, PH_Y_EXIT), None


def _stage_messages(backend, topo: BlockTopology, exchange):
    """(sends, recvs, unpack callbacks) of one exchange stage."""
    kind, direction = exchange
    sends, recvs, after = {}, {}, []
    for side in (0, 1):
        peer = topo.neighbour(direction, side)
        if peer is None:
            continue
        if kind == "field":
            sends[peer] = backend.field_boundary_rows(side)
            buf = backend.recv_buffer("field", direction, side)
            after.append((lambda s=side, b=buf: backend.set_field_halo(s, b)))
        else:
            sends[peer] = backend.pack_iface(direction, side)
            buf = backend.recv_buffer("iface", direction, side)
            after.append((lambda d=direction, s=side, b=buf: backend.unpack_iface(d, s, b)))
        recvs[peer] = buf
    return sends, recvs, after


def block_adi_steps(backend, topo: BlockTopology, transport, nsteps: int) -> None:
    """Advance the local block by ``nsteps`` ADI steps (real ranks: every process calls this collectively)."""
    for phases, exchange in block_adi_stages(nsteps):
        for ph in phases:
            backend.phase(ph)
        if exchange is not None:
            sends, recvs, after = _stage_messages(backend, topo, exchange)
            transport.exchange(sends, recvs)
            for fn in after:
                fn()


def lockstep_adi_steps(backends: list, topos: list[BlockTopology], nsteps: int) -> None:
    """Same sequence for several virtual ranks living in this process (all blocks finish a stage before the exchange)."""
    mail = LocalTransport()
    for phases, exchange in block_adi_stages(nsteps):
        for be in backends:
            for ph in phases:
                be.phase(ph)
        if exchange is None:
            continue
        pending = []
        for be, topo in zip(backends, topos):
            sends, recvs, after = _stage_messages(be, topo, exchange)
            mail.post(topo.rank, sends)
            pending.append((topo.rank, recvs, after))
        for rank, recvs, after in pending:
            mail.collect(rank, recvs)
            for fn in after:
                fn()


# ------------------------------------------------------------------------------------------------------------------ #
# HIP backend of one block
# ------------------------------------------------------------------------------------------------------------------ #
class HipBlockBackend:
    """One block of a decomposed full-rectangle grid on one GPU: tiled ADI plan + resident field planes."""

    def __init__(self, topo: BlockTopology, dx: float, dt: float, dcoef, bc_diag, bc_src, device=None):
        from . import _hip
        from .engine import RectPlan, require_gpu
        torch = require_gpu()
        self.torch = torch
        self.lib = _hip.load()
        self._hip = _hip
        self.topo = topo
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        j0, i0, ny, nx = topo.block
        self.ny, self.nx, self.nfield = ny, nx, len(dcoef)
        r = 0.5 * dt / (dx * dx)
        with torch.cuda.device(self.device):
            self.plan = RectPlan(self.lib, ny, nx, self.nfield, r, dcoef, bc_diag, bc_src,
                                 block=(topo.gny, topo.gnx, j0, i0))
        self.u = torch.zeros(self.nfield, ny * nx, dtype=torch.float64, device=self.device)
        self._bufs: dict[tuple, object] = {}

    @property
    def stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def set_field(self, global_planes: np.ndarray) -> None:
        """global_planes [nfield, gny, gnx] (host) -> local block."""
        j0, i0, ny, nx = self.topo.block
        loc = np.ascontiguousarray(global_planes[:, j0:j0 + ny, i0:i0 + nx]).reshape(self.nfield, ny * nx)
        self.u.copy_(self.torch.as_tensor(loc, device=self.device))

    def get_field(self) -> np.ndarray:
        return self.u.cpu().numpy().reshape(self.nfield, self.ny, self.nx)

    def _buf(self, key, n):
        b = self._bufs.get(key)
        if b is None:
            b = self.torch.empty(self.nfield, n, dtype=self.torch.float64, device=self.device)
            self._bufs[key] = b
        return b

    def recv_buffer(self, kind: str, direction: int, side: int):
        n = self.nx if (kind == "field" or direction == 1) else self.ny
        return self._buf(("recv", kind, direction, side), n)

    def field_boundary_rows(self, side: int):
        planes = self.u.view(self.nfield, self.ny, self.nx)
        return planes[:, 0 if side == 0 else self.ny - 1, :].contiguous()

    def set_field_halo(self, side: int, rows) -> None:
        self._hip.check(self.lib.qp_adi_rect_set_field_halo(self.plan.handle, side, int(rows.data_ptr()), self.stream),
                        "qp_adi_rect_set_field_halo")

    def pack_iface(self, direction: int, side: int):
        buf = self._buf(("send", direction, side), self.ny if direction == 0 else self.nx)
        self._hip.check(self.lib.qp_adi_rect_iface_halo(self.plan.handle, direction, side, 0, int(buf.data_ptr()),
                                                        self.stream), "qp_adi_rect_iface_halo(pack)")
        return buf

    def unpack_iface(self, direction: int, side: int, buf) -> None:
        self._hip.check(self.lib.qp_adi_rect_iface_halo(self.plan.handle, direction, side, 1, int(buf.data_ptr()),
                                                        self.stream), "qp_adi_rect_iface_halo(unpack)")

    def phase(self, ph: int) -> None:
        self._hip.check(self.lib.qp_adi_rect_phase(self.plan.handle, ph, int(self.u.data_ptr()), self.stream),
                        "qp_adi_rect_phase")


# ------------------------------------------------------------------------------------------------------------------ #
# Overlapped-halo decomposition: exchange once every S steps instead of twice per step
# ------------------------------------------------------------------------------------------------------------------ #
# The line solves of the ADI step are global, but their Green's function decays like rho^|distance| (rho ~ 0.2 for
# r D = 0.3, < 0.46 in the whole decoupled regime r D < 1.5).  A block that carries a halo of H cells of its neighbours'
# data and cuts the lines at the OUTER edge of that halo therefore computes its OWN cells with an error ~ rho^H per step
# (1e-45 for H = 64, r D = 0.3) - the same "below fp64 significance" argument by which the tiled kernels drop the far
# couplings between 64-cell chunks (qp_tile_common.h, kFarCouplingDrop = 1e-22).  The cut error creeps inwards by a few
# cells per step, so the halo stays good for S steps (``halo_steps_bound``: 19 for r D = 0.3, H = 64, 63 for H = 128;
# conservative bound 1e-20 relative) before it has to be refreshed from the neighbours.  Between refreshes a rank runs the
# ORDINARY single-GPU kernels on its extended block - S steps in one library call, 32 B per cell-update, no interface
# traffic.  A refresh is ONE round of point-to-point messages: every rank packs the H-wide strips for its side neighbours
# and the H x H corner blocks for its diagonal neighbours into one persistent send buffer (one kernel), posts all sends
# and receives of the round as one batch, and unpacks the persistent receive buffer into its halos (one kernel).  On the
# RCCL path nothing in a refresh blocks the host: the batch is ordered after the pack kernel and before the unpack kernel
# by stream dependencies.  Compared with the exact interface exchange above (``block_adi_steps``: two latency-bound
# messages per step on the critical path) this trades ~2 H / n extra cells for 2 S times fewer, larger messages; it is
# the mode the strong-scaling benchmark uses, the exact scheme remains for stiff steps (S < 1).
HALO = 64
NEIGHBOUR_OFFSETS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))     # (rows, columns) of the 8 peers


def halo_steps_bound(a: float, halo: int = HALO, limit: float = 1e-20, cap: int = 64) -> int:
    """Largest number of consecutive ADI steps for which a cut at distance ``halo`` perturbs the cells beyond it by less
    than ``limit`` (relative to the field maximum), from a worst-case recursion in absolute values: per step the error is
    mapped by |2 G - I| (G = inverse of the 1-D operator I - a L) in the cut direction, amplified by the absolute row sum
    of the same operator for the sweep in the other direction, and a new cut error 2 a |G[:, cut]| (2 |u|max) is
    injected; factor 4 for the two cut directions and the corners.  0 means: do not use the overlapped scheme."""
    a = float(a)
    if not a > 0.0:
        return cap
    n = halo + 3 * TILE
    root = (1.0 + 4.0 * a) ** 0.5
    rho = (1.0 + 2.0 * a - root) / (2.0 * a)          # decay of the Green's function of I - a L on the line
    ginf = 1.0 / root                                  # its diagonal
    idx = np.arange(n)
    K = 4.0 * ginf * rho ** np.abs(idx[:, None] - idx[None, :])      # 2 G, doubled again for the images at the walls
    K[idx, idx] = max(abs(2.0 * ginf - 1.0), abs(2.0 / (1.0 + a) - 1.0))
    cross = max(1.0, 2.0 * ginf * (1.0 + rho) / (1.0 - rho) - 2.0 * ginf + abs(2.0 * ginf - 1.0))
    inject = 2.0 * (2.0 * ginf) * rho ** idx * (2.0 * a)
    E = np.zeros(n)
    steps = 0
    for s in range(1, cap + 1):
        E = cross * (K @ E + inject)
        if 4.0 * E[halo:].max() >= limit:
            break
        steps = s
    return steps


def halo_cost_model(topo: "BlockTopology", a: float, nfield: int = 1, candidates=(64, 128), step_us_per_mcell: float = 5.6,
                    refresh_latency_us: float = 120.0, link_gbs: float = 50.0) -> dict:
    """Per-step time of the slowest rank as a function of the halo width H: the kernels run on the block + its halos every
    step, a refresh costs a fixed latency + the bytes of the largest message over one xGMI link (the messages of a round
    travel on different links at the same time) and is due every S(H) = ``halo_steps_bound(a, H)`` steps.
    ``step_us_per_mcell``: measured cost of one ADI step per million cells (4160 x 2176 block: 51 us);
    ``refresh_latency_us``: fixed cost of a refresh round (pack + one batch of point-to-point messages + unpack);
    ``link_gbs``: what one neighbour link delivers.  The bench replaces the defaults with what it measures on the
    machine.  Returns {"choice": H or None, "table": {H: {...}}}."""
    table = {}
    for H in candidates:
        S = halo_steps_bound(a, H, cap=256)
        if S < 1:
            continue
        worst = None
        for rank in range(topo.py * topo.px):
            t = BlockTopology(topo.gny, topo.gnx, topo.py, topo.px, rank)
            _, _, ny, nx = t.block
            cut_y = sum(t.neighbour(1, side) is not None for side in (0, 1))
            cut_x = sum(t.neighbour(0, side) is not None for side in (0, 1))
            if (cut_y and ny < H) or (cut_x and nx < H):
                worst = None
                break
            cells = (ny + cut_y * H) * (nx + cut_x * H)
            largest = max([H * ny] * bool(cut_x) + [H * nx] * bool(cut_y) + [0])
            if worst is None or cells > worst[0]:
                worst = (cells, largest, ny * nx)
        if worst is None:
            continue
        cells, largest, own = worst
        refresh = refresh_latency_us + 8.0 * nfield * largest / (link_gbs * 1e3)
        step = step_us_per_mcell * nfield * cells / 1e6
        table[int(H)] = {"steps_per_refresh": int(S), "cells": int(cells), "step_us": step, "refresh_us": refresh,
                         "us_per_step": step + refresh / S, "halo_cells_overhead": cells / float(own) - 1.0}
    if not table:
        return {"choice": None, "table": table}
    return {"choice": min(table, key=lambda h: table[h]["us_per_step"]), "table": table}


class OverlapBlock:
    """One rank's block of a decomposed full-rectangle grid, extended by ``halo`` cells towards every neighbour.

    ``u`` is a torch tensor [nfield, ey, ex] (device of the backend).  Subclasses implement ``advance(n)`` - n ADI steps
    on the extended block with the physical boundary condition on physical sides and a reflective wall at the outer edge
    of a halo.  Everything about windows, messages and the exchange cadence lives here, so the CPU test backend and the HIP
    backend run the same orchestration."""

    def __init__(self, topo: BlockTopology, nfield: int, a_max: float, halo: int = HALO, steps_per_exchange=None):
        self.topo, self.nfield, self.halo = topo, int(nfield), int(halo)
        j0, i0, ny, nx = topo.block
        has = lambda d, s: topo.neighbour(d, s) is not None  # noqa: E731
        self.hl, self.hr = (halo if has(0, 0) else 0), (halo if has(0, 1) else 0)
        self.hu, self.hd = (halo if has(1, 0) else 0), (halo if has(1, 1) else 0)
        if (self.hl or self.hr) and nx < halo or (self.hu or self.hd) and ny < halo:
            raise ValueError(f"blocks of {ny} x {nx} cells are smaller than the halo ({halo})")
        self.ny, self.nx = ny, nx
        self.ey, self.ex = ny + self.hu + self.hd, nx + self.hl + self.hr
        bound = halo_steps_bound(a_max, halo, cap=256)
        self.steps_per_exchange = bound if steps_per_exchange is None else min(int(steps_per_exchange), bound)
        if self.steps_per_exchange < 1 and topo.py * topo.px > 1:
            raise ValueError(f"r D = {a_max:.3g} is too stiff for the overlapped-halo scheme with a halo of {halo} cells; "
                             "use the exact interface exchange (block_adi_steps)")
        self.since_exchange = 0
        self.u = None        # set by the subclass
        self._windows = None
        self._send_buf = self._recv_buf = None

    # -- the block inside the extended array ----------------------------------------------------------------------
    @property
    def own(self):
        return self.u[:, self.hu:self.hu + self.ny, self.hl:self.hl + self.nx]

    def ext_origin(self) -> tuple[int, int]:
        j0, i0, _, _ = self.topo.block
        return j0 - self.hu, i0 - self.hl

    def side_bcs(self, bc_diag, bc_src):
        """(bc_diag[4], bc_src[4]) of the extended block: the physical condition on physical sides, reflective walls at the
        outer edge of a halo (left, right, up, down)."""
        cut = [self.hl > 0, self.hr > 0, self.hu > 0, self.hd > 0]
        return ([0.0 if c else float(v) for c, v in zip(cut, bc_diag)],
                [0.0 if c else float(v) for c, v in zip(cut, bc_src)])

    # -- windows of one refresh round -----------------------------------------------------------------------------
    def windows(self):
        """[(peer, send window, receive window)] of one refresh, one entry per existing neighbour (sides and diagonals);
        a window is (row0, col0, rows, cols) inside the extended block.  What I send towards offset (dr, dc) is the part of
        my OWN cells within ``halo`` of that side / corner; what I receive from there lands in the halo on that side."""
        if self._windows is None:
            H = self.halo

            def span(d, lead, n):            # (send start, receive start, length) along one axis for offset d
                if d < 0:
                    return lead, lead - H, H
                if d > 0:
                    return lead + n - H, lead + n, H
                return lead, lead, n

            out = []
            for dr, dc in NEIGHBOUR_OFFSETS:
                peer = self.topo.neighbour_at(dr, dc)
                if peer is None:
                    continue
                sr, rr, nr = span(dr, self.hu, self.ny)
                sc, rc, nc = span(dc, self.hl, self.nx)
                out.append((peer, (sr, sc, nr, nc), (rr, rc, nr, nc)))
            self._windows = out
        return self._windows

    def _buffers(self):
        """Persistent packed send / receive buffers of the refresh and their per-peer views (allocated once)."""
        if self._send_buf is None:
            sizes = [self.nfield * w[1][2] * w[1][3] for w in self.windows()]
            total = max(int(sum(sizes)), 1)
            self._send_buf = self.u.new_empty(total)
            self._recv_buf = self.u.new_empty(total)
            self._views, off = [], 0
            for (peer, sw, rw), n in zip(self.windows(), sizes):
                shape = (self.nfield, sw[2], sw[3])
                self._views.append((peer, self._send_buf[off:off + n].view(shape), self._recv_buf[off:off + n].view(shape)))
                off += n
        return self._views

    def pack(self) -> None:
        """Own cells near every neighbour -> the packed send buffer (device-agnostic fallback: one slice copy per window)."""
        for (peer, (r, c, nr, nc), _), (_, sview, _) in zip(self.windows(), self._buffers()):
            sview.copy_(self.u[:, r:r + nr, c:c + nc])

    def unpack(self) -> None:
        for (peer, _, (r, c, nr, nc)), (_, _, rview) in zip(self.windows(), self._buffers()):
            self.u[:, r:r + nr, c:c + nc].copy_(rview)

    def refresh_messages(self):
        """(sends, recvs) of one refresh round: per-peer views of the persistent packed buffers.  Call ``pack()`` before
        the exchange and ``unpack()`` after it."""
        views = self._buffers()
        return {peer: s for peer, s, _ in views}, {peer: r for peer, _, r in views}

    @property
    def refresh_bytes(self) -> int:
        """Bytes this rank receives per refresh."""
        return 8 * sum(self.nfield * w[2][2] * w[2][3] for w in self.windows())

    def advance(self, nsteps: int) -> None:
        raise NotImplementedError


def overlap_stages(nsteps: int, steps_per_exchange: int, since_exchange: int):
    """("steps", n) and ("refresh", None) stages of ``nsteps`` steps (a refresh is ONE round of messages); the caller tracks
    ``since_exchange`` with the same arithmetic."""
    left, since = int(nsteps), int(since_exchange)
    while left > 0:
        if since >= steps_per_exchange:
            yield ("refresh", None)
            since = 0
        n = min(left, steps_per_exchange - since)
        yield ("steps", n)
        left -= n
        since += n


def refresh_halos(block: OverlapBlock, transport) -> None:
    """One refresh of ``block``'s halos over ``transport`` (real ranks: every process calls this collectively)."""
    block.pack()
    sends, recvs = block.refresh_messages()
    transport.exchange(sends, recvs)
    block.unpack()


def overlap_adi_steps(block: OverlapBlock, transport, nsteps: int, exchange: bool = True) -> None:
    """Advance the local block by ``nsteps`` ADI steps; every rank calls this collectively with the same ``nsteps``.
    ``exchange=False`` skips the halo refreshes (timing of the exchange share only - results are then wrong)."""
    single = block.topo.py * block.topo.px == 1
    spe = max(block.steps_per_exchange, 1) if not single else max(int(nsteps), 1)
    for kind, arg in overlap_stages(nsteps, spe, block.since_exchange):
        if kind == "steps":
            block.advance(arg)
            block.since_exchange += arg
        else:
            block.since_exchange = 0
            if exchange and not single:
                refresh_halos(block, transport)


def lockstep_overlap_steps(blocks: list, nsteps: int) -> None:
    """The same sequence for several virtual ranks living in this process."""
    mail = LocalTransport()
    spe = blocks[0].steps_per_exchange
    for kind, arg in overlap_stages(nsteps, spe, blocks[0].since_exchange):
        if kind == "steps":
            for b in blocks:
                b.advance(arg)
                b.since_exchange += arg
            continue
        for b in blocks:
            b.since_exchange = 0
            b.pack()
            mail.post(b.topo.rank, b.refresh_messages()[0])
        for b in blocks:
            mail.collect(b.topo.rank, b.refresh_messages()[1])
            b.unpack()


def measure_refresh(block: OverlapBlock, transport, reps: int = 5, sync=None) -> dict:
    """Host wall-clock of the stages of a refresh (pack, exchange, unpack), each bracketed by ``sync()`` (device
    synchronisation; default: none, for CPU blocks) - microseconds, averaged over ``reps`` refreshes after one untimed.
    Timing only: the halos are simply refreshed again, results are unaffected."""
    import time
    sync = sync or (lambda: None)
    acc = {"pack_us": 0.0, "exchange_us": 0.0, "unpack_us": 0.0}
    for rep in range(reps + 1):
        sends, recvs = block.refresh_messages()
        sync()
        t0 = time.perf_counter()
        block.pack()
        sync()
        t1 = time.perf_counter()
        transport.exchange(sends, recvs)
        sync()
        t2 = time.perf_counter()
        block.unpack()
        sync()
        t3 = time.perf_counter()
        if rep:
            acc["pack_us"] += 1e6 * (t1 - t0)
            acc["exchange_us"] += 1e6 * (t2 - t1)
            acc["unpack_us"] += 1e6 * (t3 - t2)
    out = {k: v / reps for k, v in acc.items()}
    out["refresh_us"] = sum(out.values())
    out["bytes_received"] = block.refresh_bytes
    return out


class HipHaloPacking:
    """``pack`` / ``unpack`` of an ``OverlapBlock`` whose ``u`` is a contiguous device tensor [nfield, ey, ex]: all windows of
    a refresh in ONE kernel each way (``qp_halo_pack``)."""

    def _halo_pack_call(self, op: int) -> None:
        import ctypes as C
        from . import _hip
        views = self._buffers()
        if not views:
            return
        u = self.u
        if not u.is_contiguous():
            raise ValueError("the extended block must be one contiguous [nfield, ey, ex] tensor")
        which = 1 if op == 0 else 2
        rects = [v for w in self.windows() for v in w[which]]
        arr = (C.c_int32 * len(rects))(*rects)
        buf = self._send_buf if op == 0 else self._recv_buf
        torch = self.torch
        stream = int(torch.cuda.current_stream(u.device).cuda_stream)
        _hip.check(_hip.load().qp_halo_pack(int(u.data_ptr()), self.nfield, self.ey, self.ex, arr, len(self.windows()), op,
                                            int(buf.data_ptr()), stream), "qp_halo_pack")

    def pack(self) -> None:
        self._halo_pack_call(0)

    def unpack(self) -> None:
        self._halo_pack_call(1)


class HipOverlapBlock(HipHaloPacking, OverlapBlock):
    """Overlapped-halo block on one GPU: an ordinary (undecomposed) ``qp_adi_rect_plan`` on the extended block."""

    def __init__(self, topo: BlockTopology, dx: float, dt: float, dcoef, bc_diag, bc_src, halo: int = HALO, device=None,
                 steps_per_exchange=None):
        from . import _hip
        from .engine import RectPlan, require_gpu
        torch = require_gpu()
        r = 0.5 * dt / (dx * dx)
        OverlapBlock.__init__(self, topo, len(dcoef), r * float(max(dcoef)), halo, steps_per_exchange)
        self.torch, self.lib, self._hip = torch, _hip.load(), _hip
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        bd, bs = self.side_bcs(bc_diag, bc_src)
        with torch.cuda.device(self.device):
            self.plan = RectPlan(self.lib, self.ey, self.ex, self.nfield, r, dcoef, bd, bs)
        self.u = torch.zeros(self.nfield, self.ey, self.ex, dtype=torch.float64, device=self.device)

    def set_field(self, global_planes: np.ndarray) -> None:
        """global_planes [nfield, gny, gnx] (host) -> extended local block (own cells and halos)."""
        j0, i0 = self.ext_origin()
        loc = np.ascontiguousarray(global_planes[:, j0:j0 + self.ey, i0:i0 + self.ex])
        self.u.copy_(self.torch.as_tensor(loc, device=self.device))
        self.since_exchange = 0

    def get_field(self) -> np.ndarray:
        return self.own.cpu().numpy()

    def advance(self, nsteps: int) -> None:
        stream = int(self.torch.cuda.current_stream(self.device).cuda_stream)
        self._hip.check(self.lib.qp_adi_rect_steps(self.plan.handle, int(self.u.data_ptr()), int(nsteps), stream),
                        "qp_adi_rect_steps")
