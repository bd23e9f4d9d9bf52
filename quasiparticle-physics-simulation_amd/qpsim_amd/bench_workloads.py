"""Synthetic benchmark workloads of the hot path (used by the repo-level ``bench.py``).

Every workload exposes ``step()`` (one time step of the hot path on resident device state), the number of
cell-updates and algorithmic HBM bytes one step stands for, and ``roofline(n)`` which times the dominant kernel
with HIP events on the launch stream.  Inputs follow SURVEY.md 8(d): full mask, reflective walls, dx = 1,
D0 = 6, dt = 0.1, ``1e-4 (1 + default_rng(0).random)`` initial field.
"""
from __future__ import annotations

import re

import numpy as np

from . import tables as T
from .engine import DiffusionOperator, Engine, compile_geometry
from .geometry import extract_edge_segments
from .models import BoundaryCondition

HBM_PEAK_GBS = 8000.0


def _rect_engine(N: int, device):
    mask = np.ones((N, N), dtype=bool)
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    return Engine(compile_geometry(mask, edges, bcs, 1.0), device=device)


class ADIWorkload:
    """N x N scalar CN-ADI diffusion; `chunk` consecutive steps per library call share the carried state."""

    def __init__(self, N: int, device, nfield: int = 1):
        self.N, self.nfield = N, nfield
        self.eng = _rect_engine(N, device)
        torch = self.eng.torch
        rng = np.random.default_rng(0)
        init = 1e-4 * (1.0 + rng.random((nfield, N * N)))
        self.u = torch.as_tensor(init, device=self.eng.device)
        D = [6.0] * nfield
        self.op = DiffusionOperator(self.eng, nfield, 0.1, dcoef=D)
        self.grid = [N, N]
        self.cell_updates_per_step = float(N) * N * nfield
        self.bytes_per_step = 32.0 * self.cell_updates_per_step     # 2 sweeps x (8 B read + 8 B write)
        self.path = "rect-tiled partition ADI" if self.op.rect is not None else "general per-line Thomas"
        self.description = (f"{N}x{N} fp64 CN-ADI step (Peaceman-Rachford, both sweeps), {nfield} field(s), "
                            "full rectangle, reflective walls, D=6 dt=0.1 dx=1")
        self._pending = 0

    def run(self, k: int):
        """Advance exactly k diffusion steps (one library call; the field is materialised again at the end)."""
        self.eng.adi_steps(self.op, self.u, k)

    run_steps = run

    def roofline(self, nrep: int) -> dict:
        """Average duration of one sweep kernel launch (HIP events on the launch stream) vs algorithmic bytes."""
        torch = self.eng.torch
        k = 10
        self.run_steps(2)
        torch.cuda.synchronize(self.eng.device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(self.eng.device))
        for _ in range(nrep):
            self.run_steps(k)
        ev1.record(torch.cuda.current_stream(self.eng.device))
        torch.cuda.synchronize(self.eng.device)
        ms = ev0.elapsed_time(ev1)
        # one multi-step call = 1 entry pass + k x (x-sweep + y-sweep) tile kernels (+ two tiny reduced solves per step)
        sweeps = nrep * (2 * k + 1)
        per_sweep_s = ms * 1e-3 / sweeps
        bytes_per_launch = 16.0 * self.cell_updates_per_step         # 8 B read + 8 B write per cell per sweep
        achieved = bytes_per_launch / per_sweep_s / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "rect_x_kernel / rect_y_kernel (one tile sweep)" if self.op.rect is not None else "thomas_lines_kernel",
                "bytes_per_launch": bytes_per_launch, "avg_launch_us": per_sweep_s * 1e6,
                "note": "launch time = (event time of k-step calls) / (2k+1 sweep launches); includes the reduced-system kernels"}


def build(name: str, device):
    m = re.fullmatch(r"adi(\d+)", name)
    if m:
        return ADIWorkload(int(m.group(1)), device)
    raise ValueError(f"unknown workload '{name}'")
