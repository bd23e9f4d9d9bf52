#!/usr/bin/env python3
"""Timing of the exact-CN (default scheme) step on N x N rectangles or rings:  python tools/exp_cn.py 1024 4096 ring2048"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from qpsim_amd.bench_workloads import _rect_engine, ring_mask  # noqa: E402
from qpsim_amd.engine import DiffusionOperator, Engine, compile_geometry  # noqa: E402
from qpsim_amd.geometry import extract_edge_segments  # noqa: E402
from qpsim_amd.models import BoundaryCondition  # noqa: E402

for arg in sys.argv[1:]:
    ring = arg.startswith("ring")
    N = int(arg[4:] if ring else arg)
    if ring:
        mask = ring_mask(N)
        edges = extract_edge_segments(mask)
        eng = Engine(compile_geometry(mask, edges, {e.edge_id: BoundaryCondition("reflective") for e in edges}, 1.0),
                     device="cuda:0")
        u = eng.upload_packed(1e-4 * (1.0 + np.random.default_rng(0).random((1, int(mask.sum())))))
    else:
        eng = _rect_engine(N, "cuda:0")
        u = torch.as_tensor(1e-4 * (1.0 + np.random.default_rng(0).random((1, N * N))), device=eng.device)
    op = DiffusionOperator(eng, 1, 0.1, dcoef=[6.0])
    its = [eng.cn_exact_step(op, u) for _ in range(3)]
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        its.append(eng.cn_exact_step(op, u))
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    print(f"{arg}: exact-CN step {ms:.3f} ms, iterations {its}, rho bound {eng.cn_contraction_bound(op):.3f}; "
          f"{N * N / ms * 1e3:.3e} cell-updates/s", flush=True)
