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
