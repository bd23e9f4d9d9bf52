#!/usr/bin/env python3
"""Wall time per coupled step (NE = 12, full physics, guard every step) of a single small problem - the host-latency regime:
python tools/exp_guardlag.py 64 256 1024"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from qpsim_amd.bench_workloads import CoupledWorkload  # noqa: E402

for N in [int(a) for a in sys.argv[1:]]:
    wl = CoupledWorkload(N, torch.device("cuda", 0))
    wl.run(20)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        wl.run(200)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200)
    print(f"N={N}: {1e3 * best:.4f} ms/step  (guard lag {wl.eng.GUARD_LAG})", flush=True)
