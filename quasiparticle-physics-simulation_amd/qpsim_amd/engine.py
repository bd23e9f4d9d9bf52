"""Device-side engine: geometry compilation, resident state and kernel sequencing.

Host code here is plumbing only (PyTorch-ROCm tensors for device memory and streams, ctypes calls
into ``libqpsim_hip.so``).  All arithmetic of the time loop runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import math
import os

import numpy as np

from . import _hip
from .models import BoundaryCondition, EdgeSegment

FLAG_XM, FLAG_XP, FLAG_YM, FLAG_YP, FLAG_ACTIVE = 1, 2, 4, 8, 16
_DIRECTIONS = ("up", "down", "left", "right")  # the reference's face walk order (solver.py:25-30)


class BoundaryAssignmentError(ValueError):
    """A boundary face or edge has no boundary condition (reference solver.py:21)."""


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("qpsim_amd needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback for the time loop.")
    return torch


# --------------------------------------------------------------------------------------------------------- #
# geometry compilation (host, vectorised): mask + edges + edge_conditions -> per-cell operator tables
# --------------------------------------------------------------------------------------------------------- #
@dataclass
class CompiledGeometry:
    mask: np.ndarray      # [ny, nx] bool
    dx: float
    flags: np.ndarray     # [ny, nx] uint8
    ex: np.ndarray        # [ny, nx] f64, BC diagonal terms of x-faces in 1/dx^2 units
    ey: np.ndarray
    sx: np.ndarray        # BC sources in 1/dx^2 units
    sy: np.ndarray

    @property
    def is_full_rectangle(self) -> bool:
        return bool(self.mask.all())


def _face_terms(bc: BoundaryCondition, dx: float) -> tuple[float, float]:
    """(diagonal, source) contribution of one boundary face in units of 1/dx^2 (solver.py:112-149)."""
    kind = bc.normalized_kind()
    if kind == "reflective":
        return 0.0, 0.0
    if kind == "absorbing":
        return 2.0, 0.0
    if kind == "dirichlet":
        return 2.0, 2.0 * float(bc.value or 0.0)
    if kind == "neumann":
        return 0.0, float(bc.value or 0.0) * dx
    if kind == "robin":
        return float(bc.value or 0.0) * dx, float(bc.aux_value or 0.0) * dx
    raise BoundaryAssignmentError(f"Unsupported boundary kind: {bc.kind}")


def link_flags(mask: np.ndarray) -> np.ndarray:
    m = np.asarray(mask, dtype=bool)
    pad = np.zeros((m.shape[0] + 2, m.shape[1] + 2), dtype=bool)
    pad[1:-1, 1:-1] = m
    f = np.zeros(m.shape, dtype=np.uint8)
    f |= (m & pad[1:-1, :-2]).astype(np.uint8) * FLAG_XM
    f |= (m & pad[1:-1, 2:]).astype(np.uint8) * FLAG_XP
    f |= (m & pad[:-2, 1:-1]).astype(np.uint8) * FLAG_YM
    f |= (m & pad[2:, 1:-1]).astype(np.uint8) * FLAG_YP
    f |= m.astype(np.uint8) * FLAG_ACTIVE
    return f


def compile_geometry(mask: np.ndarray, edges: list[EdgeSegment], edge_conditions: dict[str, BoundaryCondition],
                     dx: float) -> CompiledGeometry:
    """Per-cell link flags and boundary terms; same checks / errors as solver.py:152-212."""
    if dx <= 0:
        raise ValueError("dx must be positive.")
    mask = np.asarray(mask, dtype=bool)
    if mask.ndim != 2:
        raise ValueError("mask must be 2D.")
    if not mask.any():
        raise ValueError("Geometry mask has no interior points.")
    ny, nx = mask.shape
    # BC index per face; later edges overwrite earlier ones, edges without a BC are skipped (solver.py:37-50)
    bc_list: list[tuple[float, float]] = []
    face_bc = {d: np.full((ny, nx), -1, dtype=np.int32) for d in _DIRECTIONS}
    for edge in edges:
        bc = edge_conditions.get(edge.edge_id)
        if bc is None:
            continue
        bc.validate()
        bc_list.append(_face_terms(bc, dx))
        k = len(bc_list) - 1
        for d in _DIRECTIONS:
            rc = [(f.row, f.col) for f in edge.faces if f.direction == d]
            if rc:
                rows, cols = np.asarray(rc, dtype=np.int64).T
                face_bc[d][rows, cols] = k
    missing = [e.edge_id for e in edges if e.edge_id not in edge_conditions]
    if missing:
        raise BoundaryAssignmentError(
            f"All edges must be assigned boundary conditions before simulation. Missing: {len(missing)}")

    flags = link_flags(mask)
    link_of = {"up": FLAG_YM, "down": FLAG_YP, "left": FLAG_XM, "right": FLAG_XP}
    diag = np.asarray([t[0] for t in bc_list] + [0.0])
    src = np.asarray([t[1] for t in bc_list] + [0.0])
    ex = np.zeros((ny, nx))
    ey = np.zeros((ny, nx))
    sx = np.zeros((ny, nx))
    sy = np.zeros((ny, nx))
    unassigned = np.zeros((ny, nx), dtype=np.int8)  # 1 + index of first direction lacking a BC
    for order, d in reversed(list(enumerate(_DIRECTIONS))):
        boundary = mask & ((flags & link_of[d]) == 0)
        k = face_bc[d]
        bad = boundary & (k < 0)
        unassigned[bad] = order + 1
        kk = np.where(boundary & (k >= 0), k, len(bc_list))
        if d in ("left", "right"):
            ex += diag[kk]
            sx += src[kk]
        else:
            ey += diag[kk]
            sy += src[kk]
    if unassigned.any():
        r, c = (int(v) for v in np.argwhere(unassigned > 0)[0])
        d = _DIRECTIONS[int(unassigned[r, c]) - 1]
        raise BoundaryAssignmentError(f"Missing boundary condition for face at cell ({r}, {c}) direction '{d}'.")
    return CompiledGeometry(mask, float(dx), flags, ex, ey, sx, sy)


# --------------------------------------------------------------------------------------------------------- #
# device engine
# --------------------------------------------------------------------------------------------------------- #
def _ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


def rect_side_terms(geom: CompiledGeometry):
    """(bc_diag[4], bc_src[4]) in left, right, up, down order if the geometry is a full rectangle whose four
    sides each carry one boundary condition; None otherwise (then only the general kernels apply)."""
    if not geom.is_full_rectangle:
        return None
    ny, nx = geom.mask.shape

    def side(arr, sl):
        vals = arr[sl]
        return float(vals.flat[0]) if np.all(vals == vals.flat[0]) else None

    if nx > 1:
        xs = [side(geom.ex, (slice(None), 0)), side(geom.ex, (slice(None), -1)),
              side(geom.sx, (slice(None), 0)), side(geom.sx, (slice(None), -1))]
    else:  # one column: both x-faces sit on the same cell; only their sum enters
        xs = [side(geom.ex, (slice(None), 0)), 0.0, side(geom.sx, (slice(None), 0)), 0.0]
    if ny > 1:
        ys = [side(geom.ey, (0, slice(None))), side(geom.ey, (-1, slice(None))),
              side(geom.sy, (0, slice(None))), side(geom.sy, (-1, slice(None)))]
    else:
        ys = [side(geom.ey, (0, slice(None))), 0.0, side(geom.sy, (0, slice(None))), 0.0]
    if any(v is None for v in xs + ys):
        return None
    return [xs[0], xs[1], ys[0], ys[1]], [xs[2], xs[3], ys[2], ys[3]]


def structured_bin_maps(idx_diff, idx_sum, sign, allow_shared: bool = False):
    """(diag_bin[NE], anti_bin[2NE-1]) if idx_diff[i][j] depends only on |i-j|, idx_sum[i][j] only on i+j and sign is
    sign(i-j); None otherwise.  Unless ``allow_shared``, None is also returned when a phonon bin is fed by both a
    diagonal (k >= 1) and an anti-diagonal (merged bins, e.g. when 2 E_min / dE is an integer)."""
    idx_diff, idx_sum, sign = np.asarray(idx_diff), np.asarray(idx_sum), np.asarray(sign)
    ne = idx_diff.shape[0]
    if ne < 2:
        return None
    i, j = np.indices((ne, ne))
    diag = idx_diff[np.arange(ne), 0]            # k = i - 0
    anti = np.concatenate([idx_sum[0, :], idx_sum[1:, -1]])   # m = 0..2ne-2
    if not (np.array_equal(idx_diff, diag[np.abs(i - j)]) and np.array_equal(idx_sum, anti[i + j])
            and np.array_equal(sign, np.sign(i - j))):
        return None
    if np.unique(diag[1:]).size != ne - 1 or np.unique(anti).size != anti.size:
        return None
    used = np.concatenate([diag[1:], anti])
    if not allow_shared and np.unique(used).size != used.size:
        return None
    return diag.astype(np.int32), anti.astype(np.int32)


def diagonal_major(table: np.ndarray) -> np.ndarray:
    """[..., ne, ne] -> [..., ne, ne] with out[k, i] = table[i, i - k] for k <= i < ne, 0 elsewhere: row k holds diagonal k
    (the pairs with E_i - E_j = k dE) indexed by the higher bin - what a sweep along that diagonal reads is contiguous."""
    t = np.asarray(table, dtype=np.float64)
    ne = t.shape[-1]
    out = np.zeros_like(t)
    for k in range(ne):
        i = np.arange(k, ne)
        out[..., k, i] = t[..., i, i - k]
    return out


def antidiagonal_major(table: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """[..., ne, ne] -> [..., 2 ne - 1, ne] with out[m, i] = scale * table[i, m - i] for 0 <= m - i < ne, 0 elsewhere: row m
    holds anti-diagonal m (the pairs with E_i + E_j fixed) indexed by one of its bins."""
    t = np.asarray(table, dtype=np.float64)
    ne = t.shape[-1]
    out = np.zeros(t.shape[:-2] + (2 * ne - 1, ne))
    for m in range(2 * ne - 1):
        i = np.arange(max(0, m - ne + 1), min(ne, m + 1))
        out[..., m, i] = scale * t[..., i, m - i]
    return out


def tag_merged_bins(diag, anti):
    """(diag, anti, n_merged): entries of a phonon bin fed by both a diagonal k >= 1 and an anti-diagonal m get
    (slot + 1) << 16 added, slot numbering the merged bins (what the register collision kernels expect, qpsim_hip.h)."""
    diag, anti = np.array(diag, dtype=np.int32), np.array(anti, dtype=np.int32)
    where = {int(b): m for m, b in enumerate(anti)}
    slots = 0
    for k in range(1, diag.size):
        m = where.get(int(diag[k]))
        if m is not None:
            tag = (slots + 1) << 16
            diag[k] |= tag
            anti[m] |= tag
            slots += 1
    return diag, anti, slots


class RectPlan:
    """Owner of a ``qp_adi_rect_plan`` (device tables + work planes of the fast full-rectangle ADI path)."""

    def __init__(self, lib, ny, nx, nfield, r, dcoef, bc_diag, bc_src, force_banded: bool = False, block=None):
        """``block = (gny, gnx, j0, i0)`` makes this the plan of one block of a decomposed gny x gnx grid."""
        self._lib = lib
        self._h = C.POINTER(_hip.RectPlan)()
        dc = (C.c_double * nfield)(*[float(v) for v in dcoef])
        bd = (C.c_double * 4)(*bc_diag)
        bs = (C.c_double * 4)(*bc_src)
        gny, gnx, j0, i0 = (ny, nx, 0, 0) if block is None else block
        _hip.check(lib.qp_adi_rect_plan_create_block(ny, nx, nfield, float(r), dc, bd, bs, int(bool(force_banded)),
                                                     int(gny), int(gnx), int(j0), int(i0), C.byref(self._h)),
                   "qp_adi_rect_plan_create_block")
        self.decoupled = (bool(lib.qp_adi_rect_plan_decoupled(self._h, 0)), bool(lib.qp_adi_rect_plan_decoupled(self._h, 1)))
        self.fine = bool(lib.qp_adi_rect_plan_fine(self._h))      # 32-cell chunks (qp_adi_fine.inc) instead of 64 x 64 tiles

    @classmethod
    def peaceman_rachford(cls, lib, ny, nx, nfield, r, dcoef, bc_diag, p: float, share: "RectPlan | None" = None):
        """Plan of one Peaceman-Rachford iteration with parameter ``p`` (``qp_adi_rect_plan_create_pr``); ``share`` lends its
        work plane (keep it alive as long as the borrower)."""
        self = cls.__new__(cls)
        self._lib = lib
        self._h = C.POINTER(_hip.RectPlan)()
        self._lender = share
        dc = (C.c_double * nfield)(*[float(v) for v in dcoef])
        bd = (C.c_double * 4)(*bc_diag)
        _hip.check(lib.qp_adi_rect_plan_create_pr(ny, nx, nfield, float(r), dc, bd, float(p),
                                                  share.handle if share is not None else None, C.byref(self._h)),
                   "qp_adi_rect_plan_create_pr")
        self.decoupled = (bool(lib.qp_adi_rect_plan_decoupled(self._h, 0)), bool(lib.qp_adi_rect_plan_decoupled(self._h, 1)))
        self.fine, self.p = bool(lib.qp_adi_rect_plan_fine(self._h)), float(p)
        return self

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self._lib.qp_adi_rect_plan_destroy(self._h)
            self._h = C.POINTER(_hip.RectPlan)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _jacobi_dn(u: float, m: float) -> tuple[float, float]:
    """(dn(u | m), K(m)) by the arithmetic-geometric mean (Abramowitz & Stegun 16.4, 17.6), parameter m = k^2 < 1."""
    a, b, c = [1.0], [math.sqrt(1.0 - m)], [math.sqrt(m)]
    while abs(c[-1]) > 1e-17 and len(a) < 40:
        a.append(0.5 * (a[-1] + b[-1]))
        b.append(math.sqrt(a[-2] * b[-1]))
        c.append(0.5 * (a[-2] - b[-2]))
    n = len(a) - 1
    K = math.pi / (2.0 * a[n])
    phi = [0.0] * (n + 1)
    phi[n] = (2.0 ** n) * a[n] * u
    for k in range(n, 0, -1):
        phi[k - 1] = 0.5 * (phi[k] + math.asin(c[k] * math.sin(phi[k]) / a[k]))
    if n == 0:
        return 1.0, K
    return math.cos(phi[0]) / math.cos(phi[1] - phi[0]), K


def peaceman_rachford_cycle(alpha: float, beta: float, J: int):
    """Jordan's optimal cyclic parameters of J Peaceman-Rachford iterations for commuting H, V with spectra in
    [alpha, beta], p_j = beta dn((2j-1) K / (2J), k), k^2 = 1 - (alpha/beta)^2, and the worst-case factor
    max_{h,v} prod_j |(h - p_j)/(h + p_j)| |(v - p_j)/(v + p_j)| of the cycle (evaluated on 4000 points of the interval)."""
    alpha, beta = float(alpha), float(max(beta, alpha * (1.0 + 1e-9)))
    m = 1.0 - (alpha / beta) ** 2
    K = _jacobi_dn(0.0, m)[1]
    ps = [beta * _jacobi_dn((2 * j - 1) * K / (2.0 * J), m)[0] for j in range(1, J + 1)]
    h = np.geomspace(alpha, beta, 4000)
    f = np.ones_like(h)
    for p in ps:
        f *= np.abs((h - p) / (h + p))
    return ps, float(f.max()) ** 2


def peaceman_rachford_parameters(alpha: float, beta: float, reduction: float, jmax: int = 24):
    """The shortest cycle (``peaceman_rachford_cycle``) whose worst-case factor is <= ``reduction`` (or the cycle of
    ``jmax`` iterations when none is).  Returns (parameters, worst-case factor)."""
    best = None
    for J in range(1, jmax + 1):
        best = peaceman_rachford_cycle(alpha, beta, J)
        if best[1] <= reduction:
            break
    return best


class TilePlan:
    """Owner of a ``qp_adi_tile_plan`` (tiled ADI path for masked grids with one diffusivity per field)."""

    def __init__(self, lib, geom: CompiledGeometry, nfield, r, dcoef=None, dfield_dev=None):
        """``dcoef``: one diffusivity per field (host values) or ``dfield_dev``: device tensor [nfield, ncell]."""
        self._lib = lib
        self._h = C.POINTER(_hip.TilePlan)()
        ny, nx = geom.mask.shape
        host = [np.ascontiguousarray(geom.flags, dtype=np.uint8)] + [
            np.ascontiguousarray(a, dtype=np.float64) for a in (geom.ex, geom.ey, geom.sx, geom.sy)]
        if dfield_dev is None:
            dc = (C.c_double * nfield)(*[float(v) for v in dcoef])
            _hip.check(lib.qp_adi_tile_plan_create(ny, nx, nfield, float(r), dc, *[a.ctypes.data for a in host],
                                                   C.byref(self._h)), "qp_adi_tile_plan_create")
        else:
            _hip.check(lib.qp_adi_tile_plan_create_var(ny, nx, nfield, float(r), int(dfield_dev.data_ptr()),
                                                       *[a.ctypes.data for a in host], C.byref(self._h)),
                       "qp_adi_tile_plan_create_var")
        counts = (C.c_int32 * 3)()
        far = C.c_double()
        _hip.check(lib.qp_adi_tile_plan_info(self._h, counts, C.byref(far)), "qp_adi_tile_plan_info")
        self.tile_counts = {"empty": counts[0], "clean": counts[1], "general": counts[2]}
        self.far_coupling = far.value

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self._lib.qp_adi_tile_plan_destroy(self._h)
            self._h = C.POINTER(_hip.TilePlan)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DiffusionOperator:
    """(I - r L_x), (I - r L_y) and friends for one time step size on a batch of fields."""

    def __init__(self, engine: "Engine", nfield: int, dt: float, dcoef=None, dfield=None, allow_fast: bool = True,
                 force_banded: bool = False, allow_tile: bool = True):
        torch = engine.torch
        self.engine = engine
        self.nfield = int(nfield)
        self.dt = float(dt)
        self.r = 0.5 * self.dt / (engine.geom.dx * engine.geom.dx)
        self.dcoef = None if dcoef is None else torch.as_tensor(np.asarray(dcoef, dtype=np.float64), device=engine.device)
        self.dfield = None if dfield is None else torch.as_tensor(
            np.ascontiguousarray(dfield, dtype=np.float64), device=engine.device)
        if (self.dcoef is None) == (self.dfield is None):
            raise ValueError("give exactly one of dcoef / dfield")
        g = engine
        self.desc = _hip.GridDesc.make(g.ny, g.nx, self.nfield, _ptr(g.d_flags), _ptr(g.d_ex), _ptr(g.d_ey), _ptr(g.d_sx),
                                  _ptr(g.d_sy), _ptr(self.dcoef), _ptr(self.dfield))
        # full rectangle + one D per field + one BC per side -> tiled partition-method kernels
        self.rect = None
        sides = rect_side_terms(engine.geom) if (allow_fast and dcoef is not None) else None
        self._sides = sides
        self._pr_cycles: dict = {}
        if sides is not None and self.dt > 0.0:
            with torch.cuda.device(engine.device):
                self.rect = RectPlan(engine.lib, g.ny, g.nx, self.nfield, self.r, np.asarray(dcoef, dtype=float),
                                     sides[0], sides[1], force_banded=force_banded)
        # any other geometry with one D per field -> tiled kernels driven by per-cell codes, unless r*D is too large for
        # 64-cell chunks to decouple (then the per-line kernels below remain)
        self.tile = None
        self.tile_refused = None
        if self.rect is None and allow_fast and allow_tile and self.dt > 0.0:
            with torch.cuda.device(engine.device):
                try:
                    if dcoef is not None:
                        self.tile = TilePlan(engine.lib, engine.geom, self.nfield, self.r, np.asarray(dcoef, dtype=float))
                    else:
                        self.tile = TilePlan(engine.lib, engine.geom, self.nfield, self.r, dfield_dev=self.dfield)
                except _hip.QPHipError as exc:
                    if exc.status != -3:      # QP_ERR_UNSUPPORTED
                        raise
                    self.tile_refused = str(exc)


def _pr_bounds(op: "DiffusionOperator"):
    """(alpha, beta) holding the spectra of H = I/2 - r D Lx and V = I/2 - r D Ly of ``op``, or None when the operator has no
    Peaceman-Rachford cycle: not a full rectangle with one D per field and one BC per side (Lx, Ly would not commute), or a
    boundary diagonal term below zero (H, V no longer bounded below by 1/2)."""
    if (op.rect is None or op._sides is None or min(op._sides[0]) < 0.0 or os.environ.get("QPSIM_CN_PR", "1") == "0"):
        return None
    amax = op.r * float(op.dcoef.max().item())
    if amax <= 0.0:
        return None
    emax = max(op._sides[0])
    return 0.5, 0.5 + amax * max(4.0, 2.0 + emax, 2.0 * emax)      # Gershgorin; 2 e: a direction one cell thick


def _pr_cycle(op: "DiffusionOperator", J: int):
    """The plans of the J-iteration Peaceman-Rachford cycle of ``op`` (cached, at most four cycles), or None."""
    if J in op._pr_cycles:
        return op._pr_cycles[J]
    cycle = None
    bounds = _pr_bounds(op)
    eng = op.engine
    if bounds is not None:
        ps, _ = peaceman_rachford_cycle(bounds[0], bounds[1], int(J))
        dc = op.dcoef.cpu().numpy()
        try:
            with eng.torch.cuda.device(eng.device):
                plans = []
                for p in ps:
                    plans.append(RectPlan.peaceman_rachford(eng.lib, eng.ny, eng.nx, op.nfield, op.r, dc, op._sides[0], p,
                                                            share=plans[0] if plans else None))
            cycle = plans
        except _hip.QPHipError as exc:
            if exc.status != -3:      # QP_ERR_UNSUPPORTED
                raise
    while len(op._pr_cycles) >= 4:      # the cycle length moves one iteration at a time: a few neighbouring lengths stay built
        op._pr_cycles.pop(next(iter(op._pr_cycles)))
    op._pr_cycles[J] = cycle
    return cycle


def _pr_initial_length(op: "DiffusionOperator", reduction: float):
    bounds = _pr_bounds(op)
    if bounds is None:
        return None
    return len(peaceman_rachford_parameters(bounds[0], bounds[1], reduction)[0])


class FrameTicket:
    """Handle of one asynchronous frame download (``Engine.download_frames_async``)."""

    def __init__(self, slot, done_event, numel: int, shape):
        self._slot, self._done, self._numel, self._shape = slot, done_event, numel, shape
        self._frames = None

    def result(self) -> np.ndarray:
        if self._frames is None:
            self._done.synchronize()
            self._frames = self._slot["host"][:self._numel].numpy().reshape(self._shape).copy()
            if self._slot["ticket"] is self:
                self._slot["ticket"] = None
            self._slot = None
        return self._frames


class Engine:
    """Owns the device copies of one geometry and sequences the kernels on torch's current stream."""

    def __init__(self, geom: CompiledGeometry, device: str | int | None = None):
        torch = require_gpu()
        self.torch = torch
        self.lib = _hip.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.geom = geom
        self.ny, self.nx = geom.mask.shape
        self.ncell = self.ny * self.nx
        up = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=self.device)  # noqa: E731
        self.d_flags = up(geom.flags, np.uint8)
        self.d_ex, self.d_ey = up(geom.ex, np.float64), up(geom.ey, np.float64)
        self.d_sx, self.d_sy = up(geom.sx, np.float64), up(geom.sy, np.float64)
        self.mask_flat = geom.mask.reshape(-1)
        self._full_mask = bool(self.mask_flat.all())
        self._interior_idx = None
        self._ws = torch.empty(int(self.lib.qp_pauli_workspace_bytes()), dtype=torch.uint8, device=self.device)
        # results of the reductions: [max occupation, spare | first-max index, first forbidden index] in ONE 32-byte buffer,
        # so that a guard ticket is one device-to-host copy (a copy is ~4.7 us on the stream: 2 % of a 1024^2 NE = 12 step)
        self._red_buf = torch.zeros(4, dtype=torch.int64, device=self.device)
        self._red_vals = self._red_buf[:2].view(torch.float64)
        self._red_idx = self._red_buf[2:]
        self._scratch = {}
        self._pinned_stream = None

    # -- plumbing -------------------------------------------------------------------------------------------
    @property
    def stream(self) -> int:
        """hipStream_t of torch's current stream; `pin_stream()` caches it for the duration of a time loop (the lookup costs
        ~4 us and a step of a small problem issues dozens of launches)."""
        if self._pinned_stream is not None:
            return self._pinned_stream
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def pin_stream(self, on: bool = True) -> None:
        self._pinned_stream = int(self.torch.cuda.current_stream(self.device).cuda_stream) if on else None

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=self.device)

    def scratch(self, key: str, numel: int):
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < numel:
            buf = self.empty(int(numel))
            self._scratch[key] = buf
        return buf

    def upload_packed(self, packed: np.ndarray):
        """[nfield, n_interior] host array (reference layout) -> [nfield, ncell] device planes, holes = 0."""
        packed = self.torch.as_tensor(np.ascontiguousarray(packed, dtype=np.float64), device=self.device)
        if self._full_mask:
            return packed
        full = self.torch.zeros((packed.shape[0], self.ncell), dtype=self.torch.float64, device=self.device)
        full[:, self._interior_index()] = packed          # scatter on the device: the host only sends the packed values
        return full

    def download_packed(self, planes) -> np.ndarray:
        if self._full_mask:
            return planes.detach().cpu().numpy()
        return planes.detach()[:, self._interior_index()].cpu().numpy()

    def _interior_index(self):
        if self._interior_idx is None:
            self._interior_idx = self.torch.as_tensor(np.flatnonzero(self.mask_flat), device=self.device)
        return self._interior_idx

    def download_frames(self, planes, scale: float = 1.0, full_shape=None, offset=(0, 0)) -> np.ndarray:
        """[nfield, ncell] device planes -> host [nfield, ny, nx] frames, NaN outside the mask (reconstruct_field), times
        `scale`; the padding happens on the device, the host only receives.  With ``full_shape`` the engine grid is the
        window at ``offset`` of a larger all-NaN frame (the solver crops padded geometries onto their bounding box)."""
        return self.download_frames_async(planes, scale, full_shape, offset).result()

    def download_frames_async(self, planes, scale: float = 1.0, full_shape=None, offset=(0, 0)) -> "FrameTicket":
        """Same frames, but the call only ENQUEUES: the NaN padding runs on the compute stream into a staging buffer, the
        device-to-host copy runs on a side stream into pinned memory, and the compute stream goes on with the next time
        step.  ``ticket.result()`` waits for that copy and returns the frames as a fresh NumPy array.  A small ring of
        staging / pinned buffer pairs is reused (one ring for single planes, one for plane sets); taking a slot whose
        previous ticket is still open completes that ticket first (its copy has had a whole store interval to finish)."""
        torch = self.torch
        planes = planes.reshape(-1, self.ncell)
        n = planes.shape[0]
        shape = (n, self.ny, self.nx) if full_shape is None else (n,) + tuple(full_shape)
        numel = int(np.prod(shape))
        if not hasattr(self, "_dl_slots"):
            self._dl_stream = torch.cuda.Stream(device=self.device)
            # Two rings: a store point of a full-physics run enqueues up to four downloads (integrated frame, state, phonon
            # planes, phonon sum) - two of one plane, two of NE / Nw planes.  Each ring holds two store points' worth of its
            # size class, so a slot is recycled only after a whole store interval (and small frames do not grow to the
            # size of the state: 2 x 2 large staging pairs instead of 3 of everything).
            self._dl_slots = {cls: [{"dev": None, "host": None, "ticket": None} for _ in range(4)] for cls in ("small", "large")}
            self._dl_next = {"small": 0, "large": 0}
        ring_name = "small" if n <= 1 else "large"
        ring = self._dl_slots[ring_name]
        slot = ring[self._dl_next[ring_name]]
        self._dl_next[ring_name] = (self._dl_next[ring_name] + 1) % len(ring)
        if slot["ticket"] is not None:
            slot["ticket"].result()                    # frees the slot (keeps the frames inside the ticket)
        if slot["dev"] is None or slot["dev"].numel() < numel:
            slot["dev"] = self.empty(numel)
            slot["host"] = torch.empty(numel, dtype=torch.float64).pin_memory()
        stage = slot["dev"][:numel]
        compute = torch.cuda.current_stream(self.device)
        embed = full_shape is not None and tuple(full_shape) != (self.ny, self.nx)
        pad_out = self.scratch("nan_pad", n * self.ncell)[:n * self.ncell] if embed else stage
        _hip.check(self.lib.qp_nan_pad(_ptr(self.d_flags), self.ncell, n, _ptr(planes), float(scale), _ptr(pad_out),
                                       self.stream), "qp_nan_pad")
        if embed:
            full = stage.view(shape)
            full.fill_(float("nan"))
            full[:, offset[0]:offset[0] + self.ny, offset[1]:offset[1] + self.nx] = pad_out.view(n, self.ny, self.nx)
        ready = torch.cuda.Event()
        ready.record(compute)
        self._dl_stream.wait_event(ready)
        with torch.cuda.stream(self._dl_stream):
            slot["host"][:numel].copy_(stage, non_blocking=True)
        done = torch.cuda.Event()
        done.record(self._dl_stream)
        ticket = FrameTicket(slot, done, numel, shape)
        slot["ticket"] = ticket
        return ticket

    def masked_sum(self, plane) -> float:
        """Sum of one plane over the cells inside the mask (holes hold 0 by invariant)."""
        return float(plane.sum().item())

    # -- diffusion --------------------------------------------------------------------------------------------
    def stencil(self, op: DiffusionOperator, u, out, c0, cx, cy, cs, rin=None, cr=0.0, norm_out=None):
        """out = c0 u + cx rLx u + cy rLy u + cs rD S + cr rin; with ``norm_out`` (device scalar) also max |out|.  Full
        rectangles with one boundary condition per side use the plan's own operator (no per-cell geometry arrays); other
        geometries the general kernel (boundary terms read only at boundary cells); the norm is formed in the same pass."""
        if op.rect is not None:
            _hip.check(self.lib.qp_adi_rect_combine(op.rect.handle, _ptr(u), _ptr(rin), _ptr(out), c0, cx, cy, cs, cr,
                                                    _ptr(self._ws) if norm_out is not None else 0, _ptr(norm_out),
                                                    self.stream), "qp_adi_rect_combine")
            return
        _hip.check(self.lib.qp_stencil_combine_norm(C.byref(op.desc), op.r, _ptr(u), _ptr(rin), _ptr(out), c0, cx, cy, cs,
                                                    cr, _ptr(self._ws) if norm_out is not None else 0, _ptr(norm_out),
                                                    self.stream), "qp_stencil_combine")

    def sweep(self, op: DiffusionOperator, direction: int, rhs, x):
        scr = self.scratch("thomas", 2 * op.nfield * self.ncell)
        _hip.check(self.lib.qp_implicit_sweep(C.byref(op.desc), op.r, direction, _ptr(rhs), _ptr(x), _ptr(scr),
                                              self.stream), "qp_implicit_sweep")

    def adi_steps(self, op: DiffusionOperator, u, nsteps: int = 1):
        """`nsteps` consecutive Peaceman-Rachford steps in place (fast path carries the state between steps)."""
        if op.rect is not None:
            _hip.check(self.lib.qp_adi_rect_steps(op.rect.handle, _ptr(u), int(nsteps), self.stream), "qp_adi_rect_steps")
            return u
        if op.tile is not None:
            _hip.check(self.lib.qp_adi_tile_steps(op.tile.handle, _ptr(u), int(nsteps), self.stream), "qp_adi_tile_steps")
            return u
        for _ in range(int(nsteps)):
            self.adi_step(op, u)
        return u

    def adi_step(self, op: DiffusionOperator, u, out=None):
        """Peaceman-Rachford step: (I-rLx)u* = (I+rLy)u + rS; (I-rLy)u' = (I+rLx)u* + rS.  In place unless `out`."""
        if op.rect is not None or op.tile is not None:
            if out is not None:
                out.copy_(u)
                u = out
            return self.adi_steps(op, u, 1)
        n = op.nfield * self.ncell
        t1 = self.scratch("adi_t1", n).view(op.nfield, self.ncell)
        t2 = self.scratch("adi_t2", n).view(op.nfield, self.ncell)
        self.stencil(op, u, t1, 1.0, 0.0, 1.0, 1.0)
        self.sweep(op, 0, t1, t1)
        self.stencil(op, t1, t2, 1.0, 1.0, 0.0, 1.0)
        res = u if out is None else out
        self.sweep(op, 1, t2, res)
        return res

    def _precondition(self, op: DiffusionOperator, res) -> None:
        """res <- (I - rLy)^-1 (I - rLx)^-1 res: the ADI factorisation applied as preconditioner."""
        if op.rect is not None:
            _hip.check(self.lib.qp_adi_rect_solve(op.rect.handle, _ptr(res), self.stream), "qp_adi_rect_solve")
        elif op.tile is not None:
            _hip.check(self.lib.qp_adi_tile_solve(op.tile.handle, _ptr(res), self.stream), "qp_adi_tile_solve")
        else:
            self.sweep(op, 0, res, res)
            self.sweep(op, 1, res, res)

    def cn_contraction_bound(self, op: DiffusionOperator) -> float:
        """Upper estimate of the contraction factor of the plain ADI-preconditioned Richardson iteration,
        (a lx / (1 + a lx)) (a ly / (1 + a ly)) with a = r max D and l = the Gershgorin bound of -L_dir (4 + boundary term)."""
        cached = getattr(op, "_cn_rho", None)
        if cached is not None:
            return cached
        dmax = float(op.dcoef.max().item()) if op.dcoef is not None else float(op.dfield.max().item())
        a = op.r * dmax
        lx = a * (4.0 + float(self.geom.ex.max())) if self.nx > 1 else a * float(self.geom.ex.max())
        ly = a * (4.0 + float(self.geom.ey.max())) if self.ny > 1 else a * float(self.geom.ey.max())
        op._cn_rho = (lx / (1.0 + lx)) * (ly / (1.0 + ly))
        return op._cn_rho

    # contraction bound from which the Chebyshev semi-iteration replaces plain Richardson: at rho = 0.3 (r D = 0.3, the
    # physical step sizes) it converges like 0.09^k instead of 0.3^k - 6 instead of 12 iterations on rough data
    CHEBYSHEV_FROM = 0.02

    def cn_exact_step(self, op: DiffusionOperator, u, rtol: float = 1e-13, max_iter: int = 2000):
        """Unsplit CN step (I - rL)u' = (I + rL)u + 2rS by ADI-preconditioned iteration, in place.

        The ADI factorisation M = (I-rLx)(I-rLy) differs from A = I - rL by r^2 Lx Ly, so
        v <- v + M^-1 (R - A v) contracts with factor rho(Tx Ty) < 1 (Tx = (I-rLx)^-1 rLx).  The starting
        guess is the ADI step itself; on strips with reflective side walls it is already exact and no iteration runs.
        Whenever the contraction bound rho = (a lx / (1 + a lx)) (a ly / (1 + a ly)) exceeds CHEBYSHEV_FROM, the same
        residual / preconditioner kernels are driven by the Chebyshev semi-iteration on the interval [1 - rho, 1] that
        holds the spectrum of M^-1 A (half the iterations at r D = 0.3, a tenth at r D = 10); if the residual ever grows
        (non-commuting Lx, Ly on an exotic mask) the plain iteration takes over for that operator.
        Raises ``RuntimeError`` when ``rtol`` is not reached within ``max_iter`` iterations - the reference's SuperLU solve
        is exact for any dt, so a silent partial solve would not be a drop-in.  Returns the number of iterations.
        """
        n = op.nfield * self.ncell
        R = self.scratch("cn_R", n).view(op.nfield, self.ncell)
        res = self.scratch("cn_res", n).view(op.nfield, self.ncell)
        v = self.scratch("cn_v", n).view(op.nfield, self.ncell)
        norms = self.scratch("cn_norms", 2)           # [max |R|, max |R - A v|], both read back in one transfer
        self.stencil(op, u, R, 1.0, 1.0, 1.0, 2.0, norm_out=norms[0:1])
        rho = self.cn_contraction_bound(op)
        # Full rectangles (commuting Lx, Ly): one cycle of Peaceman-Rachford iterations with Jordan's parameters on the
        # spectrum of I/2 - r D L_dir - 8 plane transfers per iteration against 13 of the preconditioned iteration below and
        # a larger reduction per iteration (r D = 0.3: x45 against x11).  The residual is checked afterwards; whatever is
        # left (worst-case bound missed, e.g. rounding at very small rtol) is polished by the iteration below.
        # The cycle length follows the data.  It starts at the length whose worst-case factor is 100 rtol (the old field is
        # an O(r D) guess), drops by one iteration whenever a cycle ended 30 times below the tolerance (smooth physical
        # fields need 5-6 iterations where random data needs 8 at r D = 0.3) and grows by one when a cycle missed it -
        # after which shorter cycles are not tried again for 8, 16, 32 ... steps (doubling with every miss).
        J = None
        if rho > self.CHEBYSHEV_FROM:
            J = getattr(op, "_pr_J", None)
            if J is None:
                J = op._pr_J = _pr_initial_length(op, max(100.0 * rtol, 1e-15))
                op._pr_hold, op._pr_backoff = 0, 8
        cycle = _pr_cycle(op, J) if J is not None else None
        if cycle is not None:
            # in place on u (the right-hand side R is already formed): no copies when the cycle suffices
            handles = (C.POINTER(_hip.RectPlan) * len(cycle))(*[plan.handle for plan in cycle])
            _hip.check(self.lib.qp_adi_rect_pr_cycle(handles, len(cycle), _ptr(u), _ptr(R), self.stream),
                       "qp_adi_rect_pr_cycle")
            # the check needs the NORM of the residual only (its plane is formed again by the polishing loop if the check
            # fails): no output plane
            self.stencil(op, u, None, -1.0, 1.0, 1.0, 0.0, rin=R, cr=1.0, norm_out=norms[1:2])
            scale, err = (float(x) for x in norms.cpu())
            if not np.isfinite(err):
                raise FloatingPointError("exact-CN iteration diverged (non-finite residual)")
            if err <= rtol * scale:
                if op._pr_hold > 0:
                    op._pr_hold -= 1
                elif err <= rtol * scale / 30.0 and J > 2:
                    op._pr_J = J - 1
                return len(cycle)
            op._pr_J, op._pr_hold = min(J + 1, 24), op._pr_backoff
            op._pr_backoff = min(2 * op._pr_backoff, 512)
            v.copy_(u)
        else:
            v.copy_(u)
            self.adi_step(op, v)
        if rho > self.CHEBYSHEV_FROM and not getattr(op, "_cn_plain", False):
            its = self._cn_chebyshev(op, R, res, v, norms, rho, rtol, max_iter)
            if its >= 0:
                u.copy_(v)
                return its
            op._cn_plain = True       # residual grew: spectrum outside the assumed interval; v holds the best iterate
            its = -its
        else:
            its = 0
        # The iteration count hardly changes from one step to the next (same operator, smooth data), and looking at
        # the residual costs a device-to-host round trip: the count of the previous call runs blind, then every
        # iteration is checked.  (Running an iteration more than strictly needed only lowers the residual further.)
        blind = its + getattr(op, "_cn_its", 0)
        first = its
        while True:
            look = its >= blind or its >= max_iter
            self.stencil(op, v, res, -1.0, 1.0, 1.0, 0.0, rin=R, cr=1.0, norm_out=norms[1:2] if look else None)
            if look:
                scale, err = (float(x) for x in norms.cpu())
                if not np.isfinite(err):
                    raise FloatingPointError("exact-CN iteration diverged (non-finite residual)")
                if err <= rtol * scale:
                    break
                if its >= max_iter:
                    raise RuntimeError(
                        f"exact-CN iteration did not reach rtol={rtol:g} in {max_iter} iterations (residual "
                        f"{err / max(scale, 1e-300):.3g}, estimated contraction factor {rho:.4f}); the step is too stiff "
                        "(r D = dt D / (2 dx^2) very large) - reduce dt or use diffusion_scheme='adi'.")
            self._precondition(op, res)
            _hip.check(self.lib.qp_axpy(n, 1.0, _ptr(res), _ptr(v), self.stream), "qp_axpy")
            its += 1
        # next call: one iteration fewer runs blind when this one was already far below the tolerance at its first look
        done = its - first
        op._cn_its = done - 1 if (its == blind and done > 0 and err <= 0.01 * rtol * scale) else done
        u.copy_(v)
        return its

    def _cn_chebyshev(self, op: DiffusionOperator, R, res, v, norms, rho: float, rtol: float, max_iter: int) -> int:
        """Chebyshev semi-iteration on M^-1 A (spectrum in [1 - rho, 1]) from the iterate in ``v``.  Returns the iteration
        count, or minus the count when the residual grew (the caller continues with the plain iteration)."""
        n = op.nfield * self.ncell
        d = self.scratch("cn_d", n)
        lmin, lmax = 1.0 - rho, 1.0
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rk = 1.0 / sigma
        blind = getattr(op, "_cn_cheb_its", 0)
        best = float("inf")
        its = 0
        while True:
            look = its >= blind or its >= max_iter
            self.stencil(op, v, res, -1.0, 1.0, 1.0, 0.0, rin=R, cr=1.0, norm_out=norms[1:2] if look else None)
            if look:
                scale, err = (float(x) for x in norms.cpu())
                if not np.isfinite(err):
                    raise FloatingPointError("exact-CN iteration diverged (non-finite residual)")
                if err <= rtol * scale:
                    break
                if err > 10.0 * best:
                    return -max(its, 1)
                best = min(best, err)
                if its >= max_iter:
                    raise RuntimeError(
                        f"exact-CN (Chebyshev) iteration did not reach rtol={rtol:g} in {max_iter} iterations (residual "
                        f"{err / max(scale, 1e-300):.3g}, contraction bound {rho:.6f}); reduce dt or use "
                        "diffusion_scheme='adi'.")
            self._precondition(op, res)
            if its == 0:
                c1, c2 = 1.0 / theta, 0.0
            else:
                rn = 1.0 / (2.0 * sigma - rk)
                c1, c2 = 2.0 * rn / delta, rn * rk
                rk = rn
            _hip.check(self.lib.qp_cheb_update(n, c1, _ptr(res), c2, _ptr(d), _ptr(v), self.stream), "qp_cheb_update")
            its += 1
        op._cn_cheb_its = its - 1 if (its == blind and its > 0 and err <= 0.01 * rtol * scale) else its
        return its

    def _absmax_into(self, a, out) -> None:
        _hip.check(self.lib.qp_absmax(_ptr(a), a.numel(), _ptr(self._ws), _ptr(out), self.stream), "qp_absmax")

    def absmax(self, a) -> float:
        _hip.check(self.lib.qp_absmax(_ptr(a), a.numel(), _ptr(self._ws), _ptr(self._red_vals), self.stream), "qp_absmax")
        return float(self._red_vals[0].item())

    # -- collisions, generation, reductions -------------------------------------------------------------------
    def make_collision_tables(self, kr0, ks0, rho, idx_diff, idx_sum, sign, cls_packed=None, allow_fast=True,
                              kernel: str = "auto", gap_params: dict | None = None):
        """Upload per-gap-class tables ([C,NE,NE], [C,NE]) and maps; returns an opaque handle.

        ``kernel``: "auto" | "generic" | "wave" | "wave_unstructured" forces a collision kernel (tests, A/B timing).
        ``gap_params`` (gap classes only): ``dict(E=E_bins, gaps=class_gaps, tau_r=, tau_s=, T_c=)`` - lets the register
        kernel form K^r_0, K^s_0 per pixel from gap-independent tables (they are separable in the gap)."""
        torch = self.torch
        up = lambda a, dt: None if a is None else torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=self.device)  # noqa: E731
        rho = np.atleast_2d(np.asarray(rho, dtype=np.float64))
        nclass, ne = rho.shape
        h = {"kr0": up(None if kr0 is None else np.asarray(kr0).reshape(nclass, ne, ne), np.float64),
             "ks0": up(None if ks0 is None else np.asarray(ks0).reshape(nclass, ne, ne), np.float64),
             "rho": up(rho, np.float64), "idx_diff": up(idx_diff, np.int32), "idx_sum": up(idx_sum, np.int32),
             "sign": up(sign, np.int8), "cls": None, "ne": ne, "nclass": nclass}
        nw = int(max(np.max(idx_diff), np.max(idx_sum))) + 1
        if nclass > 1:
            if cls_packed is None:
                raise ValueError("cls is required for more than one gap class")
            full = np.zeros(self.ncell, dtype=np.int32)
            full[self.mask_flat] = np.asarray(cls_packed, dtype=np.int32)
            h["cls"] = up(full, np.int32)
        h["nw"] = nw
        h["diag_bin"] = h["anti_bin"] = None
        h["merged_slots"] = 0
        structure = structured_bin_maps(idx_diff, idx_sum, sign, allow_shared=True)
        shared = structure is not None and structured_bin_maps(idx_diff, idx_sum, sign) is None
        if structure is not None and kernel != "wave_unstructured":
            diag, anti = (np.array(a, dtype=np.int32) for a in structure)
            if shared:
                # a phonon bin fed by diagonal k AND anti-diagonal m: both table entries carry (slot + 1) << 16 so that the
                # register kernels park the diagonal's sums in scratch slot `slot` until the anti-diagonal finalises the bin
                diag, anti, h["merged_slots"] = tag_merged_bins(diag, anti)
            h["diag_bin"], h["anti_bin"] = up(diag, np.int32), up(anti, np.int32)
        # the register and wave kernels visit each unordered pair once (K^r_0, K^s_0 and the bin maps of the reference are
        # symmetric, sign antisymmetric: solver.py:463-490, 668-683); caller-supplied tables of the step API that are not
        # run the generic kernel, which reads (i, j) and (j, i) separately
        sym = lambda a: a is None or np.array_equal(np.asarray(a), np.swapaxes(np.asarray(a), -1, -2))  # noqa: E731
        symmetric = (sym(kr0 if kr0 is None else np.asarray(kr0).reshape(nclass, ne, ne))
                     and sym(ks0 if ks0 is None else np.asarray(ks0).reshape(nclass, ne, ne))
                     and sym(idx_diff) and sym(idx_sum)
                     and np.array_equal(np.asarray(sign), -np.asarray(sign).T))
        h["symmetric"] = bool(symmetric)
        if (not allow_fast or not symmetric) and kernel == "auto":
            kernel = "generic"
        flag_bits = {"auto": 0, "generic": 1, "wave": 2, "wave_unstructured": 2}[kernel] | (4 if shared else 0)
        if kernel == "wave_unstructured":
            kernel = "wave"
        wave_ok = ne <= 64 and nw <= 192
        h["gap_sq"] = h["kr_amp"] = h["ks_amp"] = h["pair_inv"] = None
        classes_ok = False
        if (nclass > 1 and gap_params is not None and structure is not None
                and bool(self.lib.qp_collision_register_kernel_classes(ne))):
            from .tables import KB_UEV_PER_K
            E = np.asarray(gap_params["E"], dtype=np.float64)
            kTc = KB_UEV_PER_K * float(gap_params["T_c"])
            psum, pdiff = E[:, None] + E[None, :], E[:, None] - E[None, :]
            h["pair_inv"] = up(1.0 / np.maximum(E[:, None] * E[None, :], 1e-30), np.float64)
            h["gap_sq"] = up(np.asarray(gap_params["gaps"], dtype=np.float64) ** 2, np.float64)
            if kr0 is not None:
                h["kr_amp"] = up((1.0 / float(gap_params["tau_r"])) * (psum / kTc) ** 2 / kTc, np.float64)
            if ks0 is not None:
                ksa = (1.0 / float(gap_params["tau_s"])) * pdiff ** 2 / kTc ** 3
                np.fill_diagonal(ksa, 0.0)
                h["ks_amp"] = up(ksa, np.float64)
            classes_ok = True
        h["kernel"] = ("generic" if (kernel == "generic" or not wave_ok) else
                       "register" if (kernel == "auto" and structure is not None and (nclass == 1 or classes_ok)
                                      and bool(self.lib.qp_collision_register_kernel_available(ne))) else "wave")
        h["fast"] = h["kernel"] != "generic"      # no accumulator planes needed
        # one-pass kernel (ne = 30, 32, 40, 50): the kernel tables once more in (anti)diagonal-major order (qpsim_hip.h)
        h["ks0_diag"] = h["kr0_anti2"] = None
        if (h["kernel"] == "register" and nclass == 1 and symmetric and bool(self.lib.qp_collision_onepass_available(ne))):
            if ks0 is not None:
                h["ks0_diag"] = up(diagonal_major(np.asarray(ks0).reshape(ne, ne)), np.float64)
            if kr0 is not None:
                h["kr0_anti2"] = up(antidiagonal_major(np.asarray(kr0).reshape(ne, ne), 2.0), np.float64)
        # consecutive half-steps of neighbouring Strang steps in one pass (qp_collision_double_step_guarded)
        h["pair"] = bool(h["kernel"] == "register" and nclass == 1 and structure is not None and not shared and symmetric
                         and kernel == "auto" and self.lib.qp_collision_pair_available(ne)
                         and os.environ.get("QPSIM_COLL_PAIR", "1") != "0")
        h["struct"] = _hip.CollisionTables.make(ne, nw, nclass, _ptr(h["kr0"]), _ptr(h["ks0"]), _ptr(h["rho"]),
                                           _ptr(h["idx_diff"]), _ptr(h["idx_sum"]), _ptr(h["sign"]), _ptr(h["cls"]),
                                           _ptr(h["diag_bin"]), _ptr(h["anti_bin"]), flag_bits,
                                           _ptr(h["gap_sq"]), _ptr(h["kr_amp"]), _ptr(h["ks_amp"]), _ptr(h["pair_inv"]),
                                           _ptr(h["ks0_diag"]), _ptr(h["kr0_anti2"]))
        return h

    def collide(self, tables, state, state_out, phonon, dE, dt, en_r, en_s, update_phonons):
        need_acc = update_phonons and (en_r or en_s) and not tables["fast"]
        acc = self.scratch("coll_acc", 2 * tables["nw"] * self.ncell) if need_acc else None
        if tables["kernel"] == "register" and tables["merged_slots"] and update_phonons and en_r and en_s:
            acc = self.scratch("coll_acc", 2 * tables["merged_slots"] * self.ncell)      # merged-bin stash
        _hip.check(self.lib.qp_collision_step(C.byref(tables["struct"]), _ptr(self.d_flags), self.ncell, _ptr(state),
                                              _ptr(state_out), _ptr(phonon), _ptr(acc), float(dE), float(dt),
                                              int(bool(en_r)), int(bool(en_s)), int(bool(update_phonons)), self.stream),
                   "qp_collision_step")

    def collide_guarded(self, tables, state, state_out, phonon, dE, dt, en_r, en_s, update_phonons, floor: float,
                        ncell: int | None = None, flags=None):
        """``collide`` + the Pauli-guard reduction of ``state_out`` in one library call (``qp_collision_step_guarded``): the
        single-pass register kernels reduce the statistics of the new densities while they are still in registers.
        Returns a ticket for ``pauli_stats_result`` (asynchronous read-back, as ``pauli_stats_launch``)."""
        nc = self.ncell if ncell is None else int(ncell)
        need_acc = update_phonons and (en_r or en_s) and not tables["fast"]
        acc = self.scratch("coll_acc", 2 * tables["nw"] * nc) if need_acc else None
        if tables["kernel"] == "register" and tables["merged_slots"] and update_phonons and en_r and en_s:
            acc = self.scratch("coll_acc", 2 * tables["merged_slots"] * nc)
        nbytes = int(self.lib.qp_collision_guard_workspace_bytes(nc))
        ws = getattr(self, "_guard_ws", None)
        if ws is None or ws.numel() < nbytes:
            ws = self._guard_ws = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        hv, hi, ev = self._guard_slot()
        self._guard_ne = tables["ne"]
        _hip.check(self.lib.qp_collision_step_guarded(
            C.byref(tables["struct"]), _ptr(self.d_flags if flags is None else flags), nc, _ptr(state), _ptr(state_out),
            _ptr(phonon), _ptr(acc), float(dE), float(dt), int(bool(en_r)), int(bool(en_s)), int(bool(update_phonons)),
            float(floor), _ptr(ws), _ptr(self._red_vals), _ptr(self._red_idx), self.stream), "qp_collision_step_guarded")
        self._guard_copy(hv)
        ev.record(self.torch.cuda.current_stream(self.device))
        return hv, hi, ev, nc

    def collide_pair_guarded(self, tables, state, state_out, phonon, dE, dt_first, dt_second, gen_amount, en_r, en_s,
                             update_phonons, floor: float, ncell: int | None = None, flags=None):
        """The closing collision half-step of one Strang step, its Pauli guard, the constant / pulse generation term of the
        next step and that step's opening half-step as ONE pass over the state (``qp_collision_double_step_guarded``; only
        call when ``tables["pair"]``).  Returns the guard ticket of the step that closes (as ``collide_guarded``)."""
        nc = self.ncell if ncell is None else int(ncell)
        nbytes = int(self.lib.qp_collision_guard_workspace_bytes(nc))
        ws = getattr(self, "_guard_ws", None)
        if ws is None or ws.numel() < nbytes:
            ws = self._guard_ws = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        hv, hi, ev = self._guard_slot()
        self._guard_ne = tables["ne"]
        _hip.check(self.lib.qp_collision_double_step_guarded(
            C.byref(tables["struct"]), _ptr(self.d_flags if flags is None else flags), nc, _ptr(state), _ptr(state_out),
            _ptr(phonon), float(dE), float(dt_first), float(dt_second), float(gen_amount), int(bool(en_r)), int(bool(en_s)),
            int(bool(update_phonons)), float(floor), _ptr(ws), _ptr(self._red_vals), _ptr(self._red_idx), self.stream),
            "qp_collision_double_step_guarded")
        self._guard_copy(hv)
        ev.record(self.torch.cuda.current_stream(self.device))
        return hv, hi, ev, nc

    GUARD_LAG = 3       # guard tickets a time loop may keep outstanding (GUARD_LAG + 1 pinned read-back slots)

    def _guard_slot(self):
        """Next pinned read-back slot of the guard.  A time loop looks at the guard of step k only after step k + GUARD_LAG
        has been enqueued, so the device always has queued work while the host reads (measured: lags 1, 3 and 6 give the
        same 0.067 / 0.084 / 0.455 ms per coupled NE = 12 step at 64^2 / 256^2 / 1024^2 - the loop is not host-bound)."""
        torch = self.torch
        if not hasattr(self, "_guard_slots"):
            self._guard_slots, self._guard_host = [], {}
            for _ in range(self.GUARD_LAG + 1):
                buf = torch.empty(4, dtype=torch.int64).pin_memory()          # host image of _red_buf
                hv, hi = buf[:2].view(torch.float64), buf[2:]
                self._guard_host[hv.data_ptr()] = buf
                self._guard_slots.append((hv, hi, torch.cuda.Event()))
            self._guard_next = 0
        slot = self._guard_slots[self._guard_next]
        self._guard_next = (self._guard_next + 1) % len(self._guard_slots)
        return slot

    def _guard_copy(self, hv) -> None:
        """Asynchronous read-back of the reduction results into the pinned slot whose value view is ``hv`` (one copy)."""
        self._guard_host[hv.data_ptr()].copy_(self._red_buf, non_blocking=True)

    def add_constant(self, state, amount: float):
        _hip.check(self.lib.qp_add_constant(_ptr(self.d_flags), self.ncell, state.shape[0], _ptr(state), float(amount),
                                            self.stream), "qp_add_constant")

    def add_scaled(self, state, g, scale: float):
        _hip.check(self.lib.qp_add_scaled(state.numel(), _ptr(state), _ptr(g), float(scale), self.stream), "qp_add_scaled")

    def pauli_stats(self, state, tables, floor: float):
        """(max occupation, (energy index, cell index), forbidden (energy, cell) or None)."""
        return self.pauli_stats_result(self.pauli_stats_launch(state, tables, floor))

    def pauli_stats_launch(self, state, tables, floor: float, ncell: int | None = None, flags=None):
        """Enqueue the Pauli-guard reduction and an asynchronous read-back into pinned memory; returns a ticket for
        ``pauli_stats_result``.  GUARD_LAG + 1 tickets may be outstanding, so a time loop can enqueue the next steps
        before it looks at an earlier step's guard and the GPU never waits for the host round trip."""
        torch = self.torch
        hv, hi, ev = self._guard_slot()
        nc = self.ncell if ncell is None else int(ncell)
        self._guard_ne = tables["ne"]
        _hip.check(self.lib.qp_pauli_stats(_ptr(state), _ptr(tables["rho"]), _ptr(tables["cls"]),
                                           _ptr(self.d_flags if flags is None else flags), tables["ne"], tables["nclass"],
                                           nc, float(floor), _ptr(self._ws), _ptr(self._red_vals), _ptr(self._red_idx),
                                           self.stream), "qp_pauli_stats")
        self._guard_copy(hv)
        ev.record(torch.cuda.current_stream(self.device))
        return hv, hi, ev, nc

    def pauli_stats_result(self, ticket):
        hv, hi, ev, nc = ticket
        ev.synchronize()
        mx = float(hv[0])
        i0, i1 = int(hi[0]), int(hi[1])
        i0 = min(max(i0, 0), nc * int(self._guard_ne) - 1)
        top = (i0 // nc, i0 % nc)
        forb = None if i1 < 0 else (i1 // nc, i1 % nc)
        return mx, top, forb

    def energy_integral(self, state, dE: float):
        out = self.empty(self.ncell)
        _hip.check(self.lib.qp_energy_integrate(_ptr(state), state.shape[0], self.ncell, float(dE), _ptr(out),
                                                self.stream), "qp_energy_integrate")
        return out

    def weighted_sum(self, planes, weights):
        out = self.empty(self.ncell)
        w = self.torch.as_tensor(np.asarray(weights, dtype=np.float64), device=self.device)
        _hip.check(self.lib.qp_weighted_sum(_ptr(planes), _ptr(w), planes.shape[0], self.ncell, _ptr(out), self.stream),
                   "qp_weighted_sum")
        return out
