"""NumPy backend of one block for the decomposition orchestration tests (CPU, test-only).

Same phase / halo interface as ``qpsim_amd.distributed.HipBlockBackend`` but every block is ONE chunk per line: local
tridiagonal solves with dense-free Thomas sweeps, first/last columns of the local inverse for the interface systems.
Mathematically the exact partition (Schur) solve of the global Peaceman-Rachford step when the far corner of the block
inverses underflows (blocks >= 64 cells, a <~ 1), which is the regime the distributed path supports.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle.qp_oracle import thomas_batched
from qpsim_amd.distributed import (PH_ENTRY, PH_REDUCED_X, PH_REDUCED_Y, PH_SWEEP_X, PH_Y_CARRY, PH_Y_EXIT,
                                   BlockTopology)


class NumpyBlockBackend:
    def __init__(self, topo: BlockTopology, dx, dt, dcoef, bc_diag, bc_src):
        self.topo = topo
        j0, i0, ny, nx = topo.block
        self.j0, self.i0, self.ny, self.nx = j0, i0, ny, nx
        self.nfield = len(dcoef)
        self.a = (0.5 * dt / (dx * dx)) * np.asarray(dcoef, dtype=float)          # [nfield]
        self.bc_diag, self.bc_src = bc_diag, bc_src                               # left, right, up, down
        self.u = np.zeros((self.nfield, ny, nx))
        self.work = np.zeros_like(self.u)
        self.halo_u = [np.zeros((self.nfield, nx)), np.zeros((self.nfield, nx))]
        # per direction: tridiagonal pieces of the local operator, g/h columns, neighbour coupling flags
        self.dirs = [self._direction(topo.gnx, i0, nx, bc_diag[0], bc_diag[1], bc_src[0], bc_src[1]),
                     self._direction(topo.gny, j0, ny, bc_diag[2], bc_diag[3], bc_src[2], bc_src[3])]
        self.iface = [np.zeros((self.nfield, 2, ny)), np.zeros((self.nfield, 2, nx))]     # own (y[0], y[last])
        self.halo = [[np.zeros((self.nfield, ny)), np.zeros((self.nfield, ny))],
                     [np.zeros((self.nfield, nx)), np.zeros((self.nfield, nx))]]
        self._recv = {}

    def _direction(self, gn, off, n, e_lo, e_hi, s_lo, s_hi):
        gk = off + np.arange(n)
        links = (gk > 0).astype(float) + (gk < gn - 1)
        e = np.where(gk == 0, e_lo, 0.0) + np.where(gk == gn - 1, e_hi, 0.0)
        src = np.where(gk == 0, s_lo, 0.0) + np.where(gk == gn - 1, s_hi, 0.0)
        a = self.a[:, None]
        diag = 1.0 + a * (links + e)[None, :]
        off_in = -a * np.ones((1, n))                     # couplings inside the block
        lo = off_in.copy(); lo[:, 0] = 0.0
        hi = off_in.copy(); hi[:, -1] = 0.0
        g = thomas_batched(lo, diag, hi, np.eye(n)[0][None, :] * np.ones((self.nfield, 1)))
        h = thomas_batched(lo, diag, hi, np.eye(n)[-1][None, :] * np.ones((self.nfield, 1)))
        return dict(lo=lo, diag=diag, hi=hi, g=g, h=h, links=links, e=e, src=src, has_lo=off > 0, has_hi=off + n < gn,
                    cm=(gk > 0), cp=(gk < gn - 1))

    # ---- field / halo interface ---------------------------------------------------------------------------------
    def set_field(self, global_planes):
        self.u[...] = global_planes[:, self.j0:self.j0 + self.ny, self.i0:self.i0 + self.nx]

    def get_field(self):
        return self.u.copy()

    def recv_buffer(self, kind, direction, side):
        n = self.nx if (kind == "field" or direction == 1) else self.ny
        return self._recv.setdefault((kind, direction, side), torch.zeros(self.nfield, n, dtype=torch.float64))

    def field_boundary_rows(self, side):
        return torch.from_numpy(self.u[:, 0 if side == 0 else -1, :].copy())

    def set_field_halo(self, side, rows):
        self.halo_u[side][...] = rows.numpy()

    def pack_iface(self, direction, side):
        return torch.from_numpy(self.iface[direction][:, 0 if side == 0 else 1, :].copy())

    def unpack_iface(self, direction, side, buf):
        self.halo[direction][side][...] = buf.numpy()

    # ---- numerics -----------------------------------------------------------------------------------------------
    def _ghosts(self, direction):
        """Solved values just outside the block on both sides, [nfield, nlines] each."""
        d = self.dirs[direction]
        own = self.iface[direction]
        a = self.a[:, None]
        gl = np.zeros_like(own[:, 0])
        gr = np.zeros_like(gl)
        # interface to the previous block: E - s F = yl_prev, F - t E = yf_own; by symmetry of the equal-size
        # neighbour blocks the neighbour's h[last] equals this block's mirrored g[0] only for identical blocks, so the
        # coefficients are taken from the closed form of the GLOBAL interior operator (blocks are >= 64 cells long)
        s_t = a * d["g"][:, :1]          # a g[0]   (same value as the neighbour's a h[last] up to ~rho^n)
        if d["has_lo"]:
            yl, yf = self.halo[direction][0], own[:, 0]
            gl = (yl + s_t * yf) / (1.0 - s_t * s_t)
        s_h = a * d["h"][:, -1:]
        if d["has_hi"]:
            yl, yf = own[:, 1], self.halo[direction][1]
            gr = (yf + s_h * yl) / (1.0 - s_h * s_h)
        return gl, gr

    def _explicit(self, v, direction, gl, gr, other_src):
        """(I + a L_dir) v + a (s_dir + s_other) on the block; v is [nfield, nlines, n] with the direction last."""
        d = self.dirs[direction]
        a = self.a[:, None, None]
        prev = np.concatenate([gl[:, :, None], v[:, :, :-1]], axis=2)
        nxt = np.concatenate([v[:, :, 1:], gr[:, :, None]], axis=2)
        lap = d["cm"][None, None, :] * prev + d["cp"][None, None, :] * nxt - (d["links"] + d["e"])[None, None, :] * v
        return v + a * lap + a * d["src"][None, None, :] + a * other_src[None, :, None]

    def _other_src(self, direction):
        """Sources of the faces normal to the OTHER direction, per line of `direction`."""
        if direction == 0:   # lines = rows; y-face sources sit on global rows 0 and gny-1
            gj = self.j0 + np.arange(self.ny)
            return np.where(gj == 0, self.bc_src[2], 0.0) + np.where(gj == self.topo.gny - 1, self.bc_src[3], 0.0)
        gi = self.i0 + np.arange(self.nx)
        return np.where(gi == 0, self.bc_src[0], 0.0) + np.where(gi == self.topo.gnx - 1, self.bc_src[1], 0.0)

    def _solve(self, rhs, direction, gl, gr):
        d = self.dirs[direction]
        a = self.a[:, None]
        r = rhs.copy()
        r[:, :, 0] += a * gl
        r[:, :, -1] += a * gr
        return thomas_batched(d["lo"][:, None, :], d["diag"][:, None, :], d["hi"][:, None, :], r)

    def _reduce(self, rhs, direction):
        d = self.dirs[direction]
        self.iface[direction][:, 0, :] = np.einsum("fk,flk->fl", d["g"], rhs)
        self.iface[direction][:, 1, :] = np.einsum("fk,flk->fl", d["h"], rhs)

    def phase(self, ph):
        T = lambda x: np.swapaxes(x, 1, 2)  # noqa: E731  ([f, rows, cols] <-> [f, cols, rows])
        if ph == PH_ENTRY:
            gu = self.halo_u[0] if self.dirs[1]["has_lo"] else np.zeros((self.nfield, self.nx))
            gd = self.halo_u[1] if self.dirs[1]["has_hi"] else np.zeros((self.nfield, self.nx))
            self.work = T(self._explicit(T(self.u), 1, gu, gd, self._other_src(1)))
            self._reduce(self.work, 0)
        elif ph == PH_SWEEP_X:
            gl, gr = self._ghosts(0)
            ustar = self._solve(self.work, 0, gl, gr)
            self.work = self._explicit(ustar, 0, gl, gr, self._other_src(0))
            self._reduce(T(self.work), 1)
        elif ph in (PH_Y_CARRY, PH_Y_EXIT):
            gu, gd = self._ghosts(1)
            unew = self._solve(T(self.work), 1, gu, gd)
            if ph == PH_Y_EXIT:
                self.u = T(unew)
            else:
                self.work = T(self._explicit(unew, 1, gu, gd, self._other_src(1)))
                self._reduce(self.work, 0)
        elif ph in (PH_REDUCED_X, PH_REDUCED_Y):
            pass
        else:
            raise ValueError(ph)


class NumpyOverlapBlock:
    """CPU twin of ``qpsim_amd.distributed.HipOverlapBlock``: the oracle's ADI stepper on the extended block (physical
    boundary condition on physical sides, reflective wall at the outer edge of a halo)."""

    def __new__(cls, topo, dx, dt, dcoef, side_bc, halo=64, steps_per_exchange=None):
        from oracle import qp_oracle as O
        from qpsim_amd.distributed import OverlapBlock
        from qpsim_amd.geometry import extract_edge_segments
        from qpsim_amd.models import BoundaryCondition

        class _Impl(OverlapBlock):
            def __init__(self):
                r = 0.5 * dt / (dx * dx)
                super().__init__(topo, len(dcoef), r * max(dcoef), halo, steps_per_exchange)
                mask = np.ones((self.ey, self.ex), dtype=bool)
                edges = extract_edge_segments(mask)
                cut = {"left": self.hl > 0, "right": self.hr > 0, "up": self.hu > 0, "down": self.hd > 0}
                bcs = {e.edge_id: (BoundaryCondition("reflective") if cut[e.normal] else side_bc[e.normal]) for e in edges}
                ops = O.build_grid_ops(mask, edges, bcs, dx)
                self.steppers = [O.ADIStepper(ops, d, dt) for d in dcoef]
                self.u = torch.zeros(self.nfield, self.ey, self.ex, dtype=torch.float64)

            def set_field(self, global_planes):
                j0, i0 = self.ext_origin()
                self.u.copy_(torch.from_numpy(np.ascontiguousarray(global_planes[:, j0:j0 + self.ey, i0:i0 + self.ex])))
                self.since_exchange = 0

            def get_field(self):
                return self.own.numpy().copy()

            def advance(self, nsteps):
                a = self.u.numpy()
                for k, st in enumerate(self.steppers):
                    g = a[k].copy()
                    for _ in range(nsteps):
                        g = st.step_grid(g)
                    a[k] = g

        return _Impl()
