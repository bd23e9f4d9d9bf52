#!/usr/bin/env python3
"""Which cycles the default scheme builds and uses during a drop-in run (debugging aid of the adaptive cycle length)."""
import sys, time, warnings
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd")):
    sys.path.insert(0, p)
from qpsim_amd import engine as E
from qpsim_amd.geometry import extract_edge_segments
from qpsim_amd.models import BoundaryCondition
from qpsim_amd.solver import run_2d_crank_nicolson

warnings.simplefilter("ignore")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
log = []
orig_cycle, orig_step, orig_create = E._pr_cycle, E.Engine.cn_exact_step, E.RectPlan.peaceman_rachford.__func__

def cyc(op, red):
    t0 = time.perf_counter(); had = red in op._pr_cycles
    c = orig_cycle(op, red)
    if not had: log.append(("build", red, None if c is None else len(c), round(1e3 * (time.perf_counter() - t0), 2)))
    return c
def step(self, op, u, *a, **k):
    t0 = time.perf_counter()
    its = orig_step(self, op, u, *a, **k)
    log.append(("step", its, getattr(op, "_pr_target", None), round(1e3 * (time.perf_counter() - t0), 3)))
    return its
E._pr_cycle = cyc
E.Engine.cn_exact_step = step
mask = np.ones((N, N), dtype=bool)
edges = extract_edge_segments(mask)
bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
init = 1e-4 * (1.0 + np.random.default_rng(0).random((N, N)))
args = dict(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
            total_time=0.1 * steps, dx=1.0, store_every=steps, energy_gap=180.0, energy_max_factor=3.0, num_energy_bins=12,
            enable_recombination=True, enable_scattering=True)
t0 = time.perf_counter()
run_2d_crank_nicolson(**args)
print(f"total {time.perf_counter() - t0:.3f} s")
builds = [l for l in log if l[0] == "build"]
print("builds:", builds)
st = [l for l in log if l[0] == "step"]
print("first steps:", st[:12])
print("last steps:", st[-4:])
print("mean step ms:", np.mean([l[3] for l in st]))
