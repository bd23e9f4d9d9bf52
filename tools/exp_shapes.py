#!/usr/bin/env python3
"""Timing sweep of the rectangle ADI path over grid shapes (per-sweep launch time, fraction of the 8 TB/s HBM peak).

usage: python tools/exp_shapes.py 1024x1024 2048x2048 4160x2176x1 ...      (NYxNX[xNFIELD])
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry  # noqa: E402
from qpsim_amd.geometry import extract_edge_segments  # noqa: E402
from qpsim_amd.models import BoundaryCondition  # noqa: E402


def one(ny, nx, nf, k=10, nrep=20):
    mask = np.ones((ny, nx), dtype=bool)
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    eng = Engine(compile_geometry(mask, edges, bcs, 1.0), device="cuda:0")
    u = torch.as_tensor(1e-4 * (1.0 + np.random.default_rng(0).random((nf, ny * nx))), device=eng.device)
    op = DiffusionOperator(eng, nf, 0.1, dcoef=[6.0] * nf)
    eng.adi_steps(op, u, 3)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(nrep):
        eng.adi_steps(op, u, k)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    sweeps = nrep * (2 * k + 1)
    us = ms * 1e3 / sweeps
    gbs = 16.0 * ny * nx * nf / (us * 1e-6) / 1e9
    tiles = nf * -(-ny // 64) * -(-nx // 64)
    print(f"{ny:6d} x {nx:6d} x {nf:3d}  tiles {tiles:7d}  sweep {us:8.2f} us  {gbs:8.1f} GB/s  frac {gbs / 8000:.3f}  "
          f"cell-upd/s {ny * nx * nf / (2 * us * 1e-6):.3e}", flush=True)
    del op, eng, u
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        parts = [int(v) for v in spec.split("x")]
        one(parts[0], parts[1], parts[2] if len(parts) > 2 else 1)
