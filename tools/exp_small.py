#!/usr/bin/env python3
"""Wall time per step of the drop-in call on small grids (the reference's everyday sizes):  python tools/exp_small.py"""
import sys
import time
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd")):
    sys.path.insert(0, p)
from qpsim_amd.geometry import extract_edge_segments  # noqa: E402
from qpsim_amd.models import BoundaryCondition  # noqa: E402
from qpsim_amd.solver import run_2d_crank_nicolson  # noqa: E402

warnings.simplefilter("ignore")
for N, kw, label in [(64, {}, "scalar"), (256, {}, "scalar"),
                     (64, dict(energy_gap=180.0, energy_max_factor=3.0, num_energy_bins=12, enable_recombination=True,
                               enable_scattering=True), "NE=12 full physics"),
                     (256, dict(energy_gap=180.0, energy_max_factor=3.0, num_energy_bins=12, enable_recombination=True,
                                enable_scattering=True), "NE=12 full physics")]:
    mask = np.ones((N, N), dtype=bool)
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    init = 1e-4 * (1.0 + np.random.default_rng(0).random((N, N)))
    for scheme in ("cn_exact", "adi"):
        steps = 200
        args = dict(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
                    total_time=0.1 * steps, dx=1.0, store_every=steps, diffusion_scheme=scheme, **kw)
        run_2d_crank_nicolson(**{**args, "total_time": 0.5, "store_every": 5})      # warm-up (plan creation, first launches)
        t0 = time.perf_counter()
        run_2d_crank_nicolson(**args)
        el = time.perf_counter() - t0
        print(f"{N}x{N} {label:20s} {scheme:9s} {1e3 * el / steps:8.3f} ms/step (incl. setup {el:.2f} s for {steps} steps)", flush=True)
