#!/usr/bin/env python3
"""Benchmark of the hot path: cell-updates/s of the N x N fp64 CN-ADI step (+ optional collision workloads).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches one rank
per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

Workloads (``--workload``):
  adi4096 (default)  4096 x 4096 scalar field, all-reflective full rectangle, D=6, dt=0.1, dx=1: one step =
                     one Peaceman-Rachford CN-ADI step (both sweeps).  This is the configuration the metric
                     and the >=40 %-of-HBM-roofline target of BASELINE.json are quoted on.
  adi<N>             same at N x N (e.g. adi8192 for the cache-cold point, adi1024).
  cn<N>              the DEFAULT scheme of run_2d_crank_nicolson on the same N x N problem: one unsplit Crank-Nicolson step
                     (what the reference's SuperLU solve computes) = right-hand side + Peaceman-Rachford cycle + residual
                     check; `roofline` is the whole step against a transfer model, not one launch.
  c2                 1024 x 1024, NE=12, recombination on, phonons frozen: Strang C(dt/2) D(dt) C(dt/2) per step
                     (BASELINE configs[1]); cell-updates count NE diffusion updates per pixel per step.
  c3                 4096 x 4096, NE=12, recombination + scattering with dynamic phonons (BASELINE configs[2]).
  c4                 64 independent 256 x 256 MKID pixels per GPU, NE=12, full physics (BASELINE configs[3] is 512
                     members over 8 GPUs; members never communicate, so --gpus N runs 64 N members).
  ring<N>[x<F>]      annulus mask inscribed in N x N (the reference's donut geometry), F fields: masked tiled path;
                     cell-updates count the cells inside the mask only.
  dd<N>[c]           one N x N scalar field (c: NE = 12 coupled step) domain-decomposed over the ranks (BASELINE configs[4] =
                     dd8192 at 8 GPUs): overlapped-halo decomposition, halo refresh over RCCL every S steps; "scaling":
                     "strong".  ddx<N>: the exact interface exchange after every sweep (the scheme for stiff steps).

Multi-GPU (the driver's `--gpus N`, default workload): the headline value stays the same workload per GPU as at N = 1
(independent 4096^2 fields, ensemble sharding, no data-path collective -> "scaling": "weak"), and the SAME JSON line
carries two sub-records measured in the same run:
  "strong"    dd8192 over the N ranks (north_star: >= 6x at 8 GPUs): value, ms/step, the 1-rank time of the same 8192^2
              problem measured on rank 0 in this run, the speed-up against it, and the share of the halo exchanges;
  "ensemble"  BASELINE configs[3]: 64 N independent 256^2 MKID members, NE = 12 full physics, 64 per GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "quasiparticle-physics-simulation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half of the 157.3 TF fp32 vector peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="adi4096")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=512, help="grid edge of the bounded CPU-baseline sample")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank")
    ap.add_argument("--no-subrecords", action="store_true", help="N > 1: skip the strong-scaling / ensemble sub-records")
    ap.add_argument("--force-subrecords", action="store_true",
                    help="emit the sub-records with one rank too (with --force-dist: rehearsal of the N > 1 code path)")
    ap.add_argument("--strong-size", type=int, default=8192, help="grid edge of the strong-scaling sub-record")
    ap.add_argument("--subrecord-timeout", type=float, default=420.0,
                    help="seconds after which unfinished N > 1 sub-records are reported as errors, the line printed and "
                         "every rank exits with status 3")
    ap.add_argument("--sustained-seconds", type=float, default=0.5,
                    help="pre-spin and window length of the `sustained` sub-record (0 disables it)")
    return ap.parse_args()


def full_rectangle_problem(N: int):
    """SURVEY 8(d) synthetic inputs: full mask, reflective walls, dx=1, D0=6, dt=0.1, seeded initial field."""
    from qpsim_amd.geometry import extract_edge_segments
    from qpsim_amd.models import BoundaryCondition
    mask = np.ones((N, N), dtype=bool)
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    init = 1e-4 * (1.0 + np.random.default_rng(0).random((N, N)))
    return mask, edges, bcs, init


def host_info() -> dict:
    """Host cores of the box and the thread settings the CPU baseline ran under (north_star: core count stated)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"host_cores": os.cpu_count(), "host_cores_usable": usable,
            "omp_num_threads": os.environ.get("OMP_NUM_THREADS", "unset"),
            "threads_note": "cores = threads the timed code used: SuperLU solve and the NumPy sweeps are single-threaded"}


def cpu_baseline(args, workload: str) -> dict:
    """Reference algorithm on host cores, timed on a bounded sample of the same synthetic workload.

    ADI workloads: unsplit CN, SuperLU factor once + solve per step (solver.py:1545-1555) on a cpu_size^2 grid
    (factorisation at 4096^2 is infeasible: fill-in grows super-linearly).
    Coupled workloads (c2/c3/c4): the oracle's whole step C(dt/2) D(dt) C(dt/2) with NE = 12 on a 192^2 grid (the
    oracle vectorises the per-pixel update over pixels; the reference itself loops over pixels in Python).
    """
    from oracle import qp_oracle as O
    coupled = workload in ("c2", "c3", "c4") or workload.startswith("coupled")
    N = 192 if coupled else args.cpu_size
    mask, edges, bcs, init = full_rectangle_problem(N)
    if not coupled:
        t0 = time.perf_counter()
        ops = O.build_grid_ops(mask, edges, bcs, 1.0)
        st = O.CNStepper(ops, 6.0, 0.1)
        t_setup = time.perf_counter() - t0
        u = init[mask].astype(float)
        steps = 0
        t0 = time.perf_counter()
        while True:
            u = st.step(u)
            steps += 1
            el = time.perf_counter() - t0
            if el > 8.0 or steps >= 200:
                break
        # second CPU figure of BASELINE.md section 4: the same Peaceman-Rachford ADI step in NumPy (vectorised Thomas sweeps)
        adi = O.ADIStepper(ops, 6.0, 0.1)
        g = init.astype(float).copy()
        asteps = 0
        t0 = time.perf_counter()
        while True:
            g = adi.step_grid(g)
            asteps += 1
            ael = time.perf_counter() - t0
            if ael > 6.0 or asteps >= 200:
                break
        return {
            "value": N * N * steps / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": (f"oracle unsplit-CN (SuperLU factor once, solve per step) on {N}x{N} scalar field, {steps} steps "
                       f"in {el:.2f}s after {t_setup:.1f}s assembly+factorisation; single-threaded like the reference"),
            "numpy_adi": {"value": N * N * asteps / ael, "unit": "cell-updates/s",
                          "sample": f"oracle NumPy ADI (batched Thomas sweeps) on {N}x{N}, {asteps} steps in {ael:.2f}s"},
            **host_info(),
        }
    ne = 12
    frozen = workload == "c2"
    steps = 8
    t0 = time.perf_counter()
    O.run(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
          total_time=0.1 * steps, dx=1.0, store_every=steps, energy_gap=180.0, energy_min_factor=1.0,
          energy_max_factor=3.0, num_energy_bins=ne, enable_diffusion=True, enable_recombination=True,
          enable_scattering=workload != "c2", tau_0=440.0, T_c=1.2, bath_temperature=0.1,
          freeze_phonon_dynamics=frozen, scheme="cn")
    el = time.perf_counter() - t0
    return {
        **host_info(),
        "value": N * N * ne * steps / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
        "sample": (f"oracle full step (collision half-steps vectorised over pixels + unsplit CN/SuperLU per bin) on {N}x{N}, "
                   f"NE={ne}, {steps} steps incl. operator setup in {el:.2f}s; NumPy, one process"),
    }


def timed_steps(wl, steps: int, dev, use_dist: bool, **run_kw) -> float:
    """Wall time of exactly `steps` steps of `wl`, barrier + device sync on both sides, MAX over ranks."""
    import torch
    import torch.distributed as dist

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    sync()
    t0 = time.perf_counter()
    wl.run(steps, **run_kw)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sync()
    return elapsed


def sustained_record(args, wl, dev, use_dist: bool, world: int, ms_per_step_hint: float) -> dict:
    """The same steps in the sustained-clock regime: >= `--sustained-seconds` of them as pre-spin, then a timed window of
    the same length.  The headline window (`--steps` steps right after warm-up) lasts a few milliseconds on the ADI
    workloads and starts from an idle chip; this record is the rate a long time loop sees."""
    target = max(args.sustained_seconds, 0.0)
    n = int(min(max(args.steps, np.ceil(1e3 * target / max(ms_per_step_hint, 1e-6))), 50000))
    t0 = time.perf_counter()
    spun = 0
    while time.perf_counter() - t0 < target:       # pre-spin: whole calls of n steps until the time is up
        wl.run(n)
        import torch
        torch.cuda.synchronize(dev)
        spun += n
    elapsed = timed_steps(wl, n, dev, use_dist)
    rec = {"steps": n, "prespin_steps": spun, "prespin_s": time.perf_counter() - t0 - elapsed, "window_s": elapsed,
           "ms_per_step": 1e3 * elapsed / n, "value": wl.cell_updates_per_step * world * n / elapsed,
           "unit": "cell-updates/s",
           "hbm_frac_of_step": (wl.bytes_per_step * n / elapsed / 1e9) / HBM_PEAK_GBS}
    if hasattr(wl, "sweep_launches"):
        sweeps = wl.sweep_launches(n)
        rec["sweep_us"] = 1e6 * elapsed / sweeps
        rec["sweep_hbm_frac"] = (16.0 * wl.cell_updates_per_step / (elapsed / sweeps) / 1e9) / HBM_PEAK_GBS
        rec["note"] = "sweep_us = window / (2 steps + 1) tile-sweep launches of the one library call"
    return rec


def strong_scaling_record(args, dev, world: int, rank: int, stage: dict | None = None) -> dict:
    """north_star's strong-scaling figure, measured in this run: the SAME N x N problem (default 8192^2, BASELINE configs[4])
    first on rank 0 alone (the other ranks wait at the barrier), then decomposed over all ranks."""
    import torch
    import torch.distributed as dist
    from qpsim_amd import bench_workloads as W
    N = args.strong_size
    stage = {} if stage is None else stage

    def at(name: str) -> None:
        stage["name"] = f"strong: {name}"

    if os.environ.get("QPSIM_BENCH_HANG_RANK") == str(rank):      # rehearsal of the watchdog only: this rank never arrives
        at("rehearsal: this rank hangs on purpose (QPSIM_BENCH_HANG_RANK)")
        time.sleep(1e6)
    at("1-rank reference run on rank 0")
    t1 = torch.zeros(1, dtype=torch.float64, device=dev)
    if rank == 0:
        single = W.ADIWorkload(N, dev)
        single.run(max(args.warmup, 1))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        single.run(args.steps)
        torch.cuda.synchronize(dev)
        t1[0] = time.perf_counter() - t0
        del single
        torch.cuda.empty_cache()
    at("all_reduce(1-rank time)")
    dist.all_reduce(t1, op=dist.ReduceOp.MAX)        # everybody learns the 1-rank time (and waits for it)
    one_rank = float(t1.item())
    # Halo width: wider halos cost cells every step and save refreshes (S = 19 steps at 64 cells, 64 at 128 for r D = 0.3).
    # Both candidates are built and MEASURED here - step time with the refreshes skipped, and the stages of a refresh -
    # and the one with the smaller time per step runs the timed region (QPSIM_DD_HALO forces one).
    from qpsim_amd.distributed import measure_refresh
    forced = os.environ.get("QPSIM_DD_HALO")
    table, built = {}, {}
    for H in ([int(forced)] if forced else [64, 128]):
        at(f"halo {H}: plan")
        cand, why = None, ""
        try:
            cand = W.OverlapDecomposedWorkload(N, dev, halo=H)
        except ValueError as exc:          # blocks smaller than the halo, or the step too stiff for it
            why = str(exc)
        fits = torch.tensor([1.0 if cand is not None else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(fits, op=dist.ReduceOp.MIN)          # a width is a candidate only if it fits on EVERY rank
        if float(fits.item()) < 1.0:
            table[H] = {"error": why or "does not fit on another rank"}
            del cand
            continue
        spe = cand.block.steps_per_exchange
        at(f"halo {H}: warm-up incl. first halo refresh (send/recv connections)")
        cand.run(max(args.warmup, spe + 1))
        k = max(5, min(args.steps, 20))
        at(f"halo {H}: timed steps without refreshes")
        step_us = 1e6 * timed_steps(cand, k, dev, True, exchange=False) / k
        at(f"halo {H}: refresh stages (pack, isend/irecv batch, unpack)")
        if world > 1:
            st = measure_refresh(cand.block, cand.transport, reps=5, sync=lambda: torch.cuda.synchronize(dev))
        else:
            st = {"pack_us": 0.0, "exchange_us": 0.0, "unpack_us": 0.0, "refresh_us": 0.0, "bytes_received": 0}
        vec = torch.tensor([st["pack_us"], st["exchange_us"], st["unpack_us"], st["refresh_us"]], dtype=torch.float64,
                           device=dev)
        dist.all_reduce(vec, op=dist.ReduceOp.MAX)          # every rank sees the same numbers and makes the same choice
        pack_us, exchange_us, unpack_us, refresh_us = (float(v) for v in vec.tolist())
        table[H] = {"steps_per_refresh": spe, "step_us": step_us, "pack_us": pack_us, "exchange_us": exchange_us,
                    "unpack_us": unpack_us, "refresh_us": refresh_us, "us_per_step_model": step_us + refresh_us / max(spe, 1),
                    "bytes_received_per_refresh": st["bytes_received"], "halo_cells_overhead": cand.halo_overhead}
        built[H] = cand
    if not built:
        raise RuntimeError(f"no halo width fits: {table}")
    choice = min(built, key=lambda h: table[h]["us_per_step_model"])
    wl = built[choice]
    for h in list(built):
        if h != choice:
            del built[h]
    torch.cuda.empty_cache()
    spe = wl.block.steps_per_exchange
    at("timed steps with halo refreshes (isend/irecv + barrier)")
    elapsed = timed_steps(wl, args.steps, dev, True)
    at("timed steps without halo refreshes")
    no_xchg = timed_steps(wl, args.steps, dev, True, exchange=False)      # same kernels, refreshes skipped: timing only
    return {
        "workload": wl.description, "path": wl.path, "scaling": "strong", "rccl_ranks": dist.get_world_size(),
        "backend": dist.get_backend(),
        "value": float(N) * N * args.steps / elapsed, "unit": "cell-updates/s", "ms_per_step": 1e3 * elapsed / args.steps,
        "one_rank_ms_per_step": 1e3 * one_rank / args.steps, "speedup_vs_one_rank": one_rank / elapsed,
        "exchange_share_of_time": max(0.0, 1.0 - no_xchg / elapsed), "steps_per_halo_refresh": spe,
        "refresh_rounds": 1, "halo": choice, "halo_candidates": table,
        "pack_us": table[choice]["pack_us"], "exchange_us": table[choice]["exchange_us"],
        "unpack_us": table[choice]["unpack_us"], "refresh_us": table[choice]["refresh_us"],
        "halo_cells_overhead": wl.halo_overhead,
        "hbm_frac_of_step_per_gpu": (32.0 * N * N / world * args.steps / elapsed / 1e9) / HBM_PEAK_GBS,
        "note": "one refresh = pack kernel + ONE batch of point-to-point messages (sides and corners) + unpack kernel; "
                "the stage times are host wall-clock with a device synchronisation after each stage (max over ranks)",
    }


def ensemble_record(args, dev, world: int, stage: dict | None = None) -> dict:
    """BASELINE configs[3]: ensemble of independent 256^2 MKID pixels, NE = 12 full physics, 64 members per GPU
    (512 at 8 GPUs), members never communicate."""
    from qpsim_amd import bench_workloads as W
    stage = {} if stage is None else stage
    stage["name"] = "ensemble: setup + warm-up"
    wl = W.build("c4", dev)
    wl.run(max(1, min(args.warmup, 5)))
    steps = max(1, min(args.steps, 50))
    stage["name"] = "ensemble: timed steps (barrier + all_reduce of the time)"
    elapsed = timed_steps(wl, steps, dev, True)
    return {"workload": wl.description, "scaling": "weak", "members_total": 64 * world,
            "value": wl.cell_updates_per_step * world * steps / elapsed, "unit": "cell-updates/s", "steps": steps,
            "ms_per_step": 1e3 * elapsed / steps, "pixel_steps_per_s": wl.npix * world * steps / elapsed}


_RESULT_FD = None


def _claim_stdout() -> None:
    """Keep stdout for the ONE JSON line: file descriptor 1 is pointed at stderr for everything else.  RCCL prints a
    version banner ("HIP version : ...", "Librccl path : ...") to stdout when a communicator is created, gloo its
    connection summary - with N ranks that is 4 N lines in front of the result."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def _emit(result: dict) -> None:
    line = (json.dumps(result) + "\n").encode()
    sys.stdout.flush()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, line)


def main():
    args = parse_args()
    _claim_stdout()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs (not used by the driver): QPSIM_BENCH_BACKEND=gloo and QPSIM_BENCH_DEVICE=0 let several ranks share one
    # GPU with gloo as the transport, which exercises the whole N > 1 code path where RCCL (one rank per GPU) cannot run.
    backend = os.environ.get("QPSIM_BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("QPSIM_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", device_index)

    from qpsim_amd import _hip
    from qpsim_amd import bench_workloads as W

    build = _hip.build_info()
    if build.get("qp_abl", 0) != 0:
        sys.stderr.write(f"bench.py: {build['library']} is a timing-only ablation build (QP_ABL={build['qp_abl']}); its "
                         "results are wrong by construction - refusing to measure it\n")
        sys.exit(4)

    wl = W.build(args.workload, dev)
    if args.warmup > 0:
        wl.run(args.warmup)
    elapsed = timed_steps(wl, args.steps, dev, use_dist)     # exactly `steps` steps of the hot path, enqueued back to back

    # per-kernel timing of the dominant kernel with HIP events on the launch stream (same inputs, same loop)
    roof = wl.roofline(max(5, min(args.steps, 20)))
    strong = getattr(wl, "scaling", "weak") == "strong"
    value = wl.cell_updates_per_step * world * args.steps / elapsed
    result = {
        "metric": "cell-updates/sec on N×N CN ADI step; achieved HBM GB/s vs roofline",
        "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl.description, "grid": wl.grid, "fields_per_gpu": wl.nfield,
                   "parallelism": ("single GPU" if world == 1 else
                                   f"domain decomposition over {world} GPUs (RCCL point-to-point halo refresh)" if strong
                                   else f"independent problems x{world} (ensemble sharding, no collective)"),
                   "path": wl.path},
        "roofline": roof,
        "hbm_frac_of_step": (wl.bytes_per_step * args.steps / elapsed / 1e9) / HBM_PEAK_GBS,
        "build": build,
    }
    if args.sustained_seconds > 0:
        result["sustained"] = sustained_record(args, wl, dev, use_dist, world, 1e3 * elapsed / args.steps)
    if hasattr(wl, "coll_bytes_per_call"):
        result["pixel_steps_per_s"] = wl.npix * world * args.steps / elapsed
    del wl
    torch.cuda.empty_cache()
    if (world > 1 or (args.force_subrecords and use_dist)) and not args.no_subrecords and args.workload == "adi4096":
        # north_star's two multi-GPU figures, measured in the same run and reported next to the weak-scaling headline
        # The headline above is complete at this point; the sub-records must not be able to lose it.  An exception in
        # one becomes its "error" field, and if a rank gets stuck in them (a peer died, a collective that never returns)
        # every rank's watchdog ends its process after `--subrecord-timeout` seconds, rank 0 printing the line first.
        printed = threading.Event()
        stage = {"name": "start"}

        def watchdog():
            # A hung process that has touched the GPU must not be reported as success: the headline line is printed (it is
            # complete), the pending stage goes to stderr, and every rank leaves with status 3.
            sys.stderr.write(f"bench.py rank {rank}: sub-records not finished within {args.subrecord_timeout:.0f} s, "
                             f"pending stage: {stage['name']}; exiting with status 3\n")
            sys.stderr.flush()
            if rank == 0 and not printed.is_set():
                printed.set()
                for key in ("strong", "ensemble"):
                    result.setdefault(key, {"error": f"not finished within {args.subrecord_timeout:.0f} s "
                                                     f"(pending stage: {stage['name']})"})
                _emit(result)
            os._exit(3)

        timer = threading.Timer(args.subrecord_timeout + (0.0 if rank == 0 else 5.0), watchdog)
        timer.daemon = True
        timer.start()
        failed = False
        for key, fn in (("strong", lambda: strong_scaling_record(args, dev, world, rank, stage)),
                        ("ensemble", lambda: ensemble_record(args, dev, world, stage))):
            if failed:        # a rank that left a record early is out of step with its peers' collectives: stop here
                result[key] = {"error": "skipped: an earlier sub-record failed on some rank"}
                continue
            stage["name"] = f"{key}: setup"
            try:
                result[key] = fn()
                ok = 1.0
            except Exception as exc:      # noqa: BLE001 - reported in the line, the headline stays valid
                result[key] = {"error": f"{type(exc).__name__}: {exc}"}
                ok = 0.0
            # every rank learns whether ALL ranks finished this record; a failure anywhere marks the record on rank 0 too
            stage["name"] = f"{key}: all_reduce(ok)"
            flag = torch.tensor([ok], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) < 1.0:
                failed = True
                if "error" not in result[key]:
                    result[key] = {"error": "another rank failed in this sub-record; timings discarded", **result[key]}
        stage["name"] = "done"
        timer.cancel()
        if printed.is_set():              # the watchdog fired between the last record and cancel()
            os._exit(3)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0 would hold the others up)
            result["cpu_baseline"] = cpu_baseline(args, args.workload)
        _emit(result)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
