#!/usr/bin/env python3
"""Is a small-grid ADI loop bound by the GPU or by the CPU's launch rate?  Per sweep: CPU time to enqueue vs GPU time.
usage: python tools/exp_launch.py N [k]   (QPSIM_FINE_TILES=0/1 selects the tile family)"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch
from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry
from qpsim_amd.geometry import extract_edge_segments
from qpsim_amd.models import BoundaryCondition

N = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mask = np.ones((N, N), dtype=bool)
edges = extract_edge_segments(mask)
bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
eng = Engine(compile_geometry(mask, edges, bcs, 1.0), device="cuda:0")
u = torch.as_tensor(1e-4 * (1.0 + np.random.default_rng(0).random((1, N * N))), device=eng.device)
op = DiffusionOperator(eng, 1, 0.1, dcoef=[6.0])
eng.adi_steps(op, u, 50)
torch.cuda.synchronize()
for rep in range(3):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    eng.adi_steps(op, u, k)
    ev1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = 2 * k + 1
    print(f"N={N} fine={op.rect.fine} k={k}: enqueue {1e6 * (t1 - t0) / n:.2f} us/launch, GPU {1e3 * ev0.elapsed_time(ev1) / n:.2f} us/launch, "
          f"wall {1e6 * (t2 - t0) / n:.2f} us/launch")
