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
FP64_VECTOR_PEAK_TFLOPS = 78.6     # half of the 157.3 TF fp32 vector peak (MI355X_MICROARCH.md, chip-level parameters)


def _rect_engine(N: int, device, nx: int | None = None):
    mask = np.ones((N, N if nx is None else nx), dtype=bool)
    edges = extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
    return Engine(compile_geometry(mask, edges, bcs, 1.0), device=device)


def ring_mask(N: int) -> np.ndarray:
    """Annulus inscribed in the N x N grid (outer radius ~N/2, inner radius 0.15 N): the reference's "donut" test
    geometry (qpsim/test_cases.py:627-780) at benchmark size."""
    y, x = np.indices((N, N))
    rr = np.hypot(y - (N - 1) / 2.0, x - (N - 1) / 2.0)
    return (rr <= 0.498 * N) & (rr >= 0.15 * N)


class ADIWorkload:
    """N x N scalar CN-ADI diffusion; `chunk` consecutive steps per library call share the carried state."""

    def __init__(self, N: int, device, nfield: int = 1, ring: bool = False):
        self.N, self.nfield, self.ring = N, nfield, ring
        if ring:
            mask = ring_mask(N)
            edges = extract_edge_segments(mask)
            bcs = {e.edge_id: BoundaryCondition("reflective") for e in edges}
            self.eng = Engine(compile_geometry(mask, edges, bcs, 1.0), device=device)
            ncell = int(mask.sum())
        else:
            self.eng = _rect_engine(N, device)
            ncell = N * N
        torch = self.eng.torch
        rng = np.random.default_rng(0)
        init = 1e-4 * (1.0 + rng.random((nfield, ncell)))
        self.u = self.eng.upload_packed(init) if ring else torch.as_tensor(init, device=self.eng.device)
        D = [6.0] * nfield
        self.op = DiffusionOperator(self.eng, nfield, 0.1, dcoef=D)
        self.grid = [N, N]
        self.cell_updates_per_step = float(ncell) * nfield          # cells inside the mask only
        self.bytes_per_step = 32.0 * self.cell_updates_per_step     # 2 sweeps x (8 B read + 8 B write)
        self.path = ("rect-tiled partition ADI, fine tiles (32-cell chunks)" if self.op.rect is not None and self.op.rect.fine else
                     "rect-tiled partition ADI" if self.op.rect is not None else
                     f"masked tiled ADI {self.op.tile.tile_counts}" if self.op.tile is not None else "general per-line Thomas")
        shape = "ring mask (donut)" if ring else "full rectangle"
        self.description = (f"{N}x{N} fp64 CN-ADI step (Peaceman-Rachford, both sweeps), {nfield} field(s), "
                            f"{shape}, reflective walls, D=6 dt=0.1 dx=1")
        self._pending = 0

    def run(self, k: int):
        """Advance exactly k diffusion steps (one library call; the field is materialised again at the end)."""
        self.eng.adi_steps(self.op, self.u, k)

    run_steps = run

    @staticmethod
    def sweep_launches(k: int) -> int:
        """Tile-sweep launches of one ``run(k)``: the entry pass + an x- and a y-sweep per step."""
        return 2 * int(k) + 1

    def _pmc_traffic(self):
        """HBM bytes per sweep from the committed PMC summaries (profiles/, separate rocprofv3 --pmc passes of this same
        command); only the configurations that were measured get a number."""
        import json
        from pathlib import Path
        prof = Path(__file__).resolve().parents[2] / "profiles"
        if self.nfield != 1 or self.N not in ((4096,) if self.ring else (1024, 2048, 4096)):
            return None
        stem = "ring4096_pmc.json" if self.ring else f"adi{self.N}_pmc.json"
        found = sorted(prof.glob(f"r0*_{stem}"))          # the latest round's passes
        if not found:
            return None
        self._traffic_file = found[-1].name
        k = json.loads(found[-1].read_text())["kernels"]
        pick = lambda frag: sum(v["hbm_bytes_per_launch"] for n_, v in k.items() if frag in n_)  # noqa: E731
        if self.op.rect is not None and self.op.rect.fine:
            val = 0.5 * (pick("fine_x_kernel<true, 0, false>") + pick("fine_y_kernel<1, 0, false>"))
        elif self.op.rect is not None:
            val = 0.5 * (pick("rect_x_kernel<true, 0, true>") + pick("rect_y_kernel<1, 0, true>"))
        elif self.op.tile is not None:      # one sweep = one merged launch (clean + general tiles); mean of x and carried y
            val = 0.5 * (pick("tile_x_merged_kernel<true>") + pick("tile_y_merged_kernel<1>"))
        else:
            return None
        return val or None

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
        traffic = self._pmc_traffic()
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": None if traffic is None else
                f"profiles/{getattr(self, '_traffic_file', '')} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                "workload, not measured in this run)",
                "kernel": ("fine_x_kernel / fine_y_kernel (one tile sweep, 32-cell chunks)"
                           if self.op.rect is not None and self.op.rect.fine else
                           "rect_x_kernel / rect_y_kernel (one tile sweep)" if self.op.rect is not None else
                           "tile_x_kernel / tile_y_kernel, clean + general launches of one sweep" if self.op.tile is not None
                           else "thomas_lines_kernel"),
                "bytes_per_launch": bytes_per_launch, "avg_launch_us": per_sweep_s * 1e6,
                "note": "launch time = (event time of k-step calls) / (2k+1 sweep launches); includes the reduced-system kernels"}


class ExactCNWorkload(ADIWorkload):
    """The DEFAULT scheme of `run_2d_crank_nicolson` (unsplit Crank-Nicolson, what the reference's SuperLU solve computes,
    `solver.py:231,1155-1161,1441`): one `Engine.cn_exact_step` per step on a full rectangle - right-hand side, one carried
    Peaceman-Rachford cycle of J iterations, residual check."""

    def __init__(self, N: int, device):
        super().__init__(N, device)
        self.description = (f"{N}x{N} fp64 unsplit Crank-Nicolson step (default scheme: Peaceman-Rachford cycle to "
                            "rtol 1e-13 of the residual), 1 field, full rectangle, reflective walls, D=6 dt=0.1 dx=1")
        self.path = "qp_adi_rect_combine + qp_adi_rect_pr_cycle (carried, fine tiles where the plans qualify) + residual check"
        self.iterations = []

    def run(self, k: int):
        for _ in range(k):
            self.iterations.append(self.eng.cn_exact_step(self.op, self.u))

    run_steps = run

    def roofline(self, nrep: int) -> dict:
        """Plane transfers of one step by the cycle's own accounting (2 for the right-hand side, 6 J + 2 for the carried
        cycle, 3 for the residual check) x 8 B x cells, over the measured step time."""
        torch = self.eng.torch
        self.run(2)
        torch.cuda.synchronize(self.eng.device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(self.eng.device))
        self.run(nrep)
        ev1.record(torch.cuda.current_stream(self.eng.device))
        torch.cuda.synchronize(self.eng.device)
        per_step = ev0.elapsed_time(ev1) * 1e-3 / nrep
        J = int(self.iterations[-1])
        transfers = 2 + 6 * J + 2 + 3
        moved = transfers * 8.0 * self.cell_updates_per_step
        achieved = moved / per_step / 1e9
        self.bytes_per_step = moved
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "traffic_source": None,
                "kernel": "one exact-CN step: rect_combine_kernel x2, fine_y_kernel<0,SRC>, J x fine_x_kernel<SRC>, "
                          "(J-1) x fine_y_next_kernel, fine_y_kernel<2>",
                "bytes_per_launch": moved, "avg_launch_us": per_step * 1e6, "iterations_per_step": J,
                "note": f"whole step, not one launch: {transfers} plane transfers of 8 B per cell (model) over the step time"}


class CoupledWorkload:
    """Energy-resolved hot loop: external generation off, Strang C(dt/2) D(dt) C(dt/2), Pauli guard every step.

    SURVEY 8(d) parameters: gap 180 ueV, factors 1-3, NE = 12, tau_0 = 440 ns, Tc = 1.2 K, Tb = 0.1 K, gamma = 0.
    c2: N = 1024, recombination only, phonons frozen.  c3: N = 4096, recombination + scattering, dynamic phonons.
    """

    def __init__(self, N, device, *, ne=12, recombination=True, scattering=True, dynamic_phonons=True, label="",
                 members=1, fmax=3.0, nx=None, init_occupation=None, gap_classes=1):
        """``members`` independent N x N problems are batched: planes are laid out [bin][member][cell], so the ADI plan
        sees NE*members fields of N x N and the collision kernel sees members*N*N pixels (no coupling between members).
        ``nx``: N x nx rectangle instead of a square; ``init_occupation``: device tensor [N*nx] replacing the seeded field;
        ``gap_classes`` > 1: non-uniform gap (one gap value per block of columns, 0.9 gap ... gap), collision tables per class."""
        self.N, self.nfield, self.ne, self.members = N, ne * members, ne, members
        NX = N if nx is None else int(nx)
        self.eng = eng = _rect_engine(N, device, NX)
        torch = eng.torch
        gap, D0, dt = 180.0, 6.0, 0.1
        self.dt = dt
        E, dE = T.build_energy_grid(gap, 1.0, fmax, ne)
        self.dE = dE
        om, idx_d, idx_s, sg = T.build_phonon_frequency_map(E)
        self.nw = om.size
        rho = T.dynes_density_of_states(E, gap, 0.0)
        if gap_classes > 1:
            gaps = np.linspace(0.9 * gap, gap, gap_classes)
            cls = np.tile((np.arange(NX) * gap_classes) // NX, N * members)
            rho_c = np.stack([T.dynes_density_of_states(E, g, 0.0) for g in gaps])
            kr = np.stack([T.recombination_kernel_base(E, g, 440.0, 1.2) for g in gaps]) if recombination else None
            ks = np.stack([T.scattering_kernel_base(E, g, 440.0, 1.2) for g in gaps]) if scattering else None
            self.tab = eng.make_collision_tables(kr, ks, rho_c, idx_d, idx_s, sg, cls,
                                                 gap_params=dict(E=E, gaps=gaps, tau_r=440.0, tau_s=440.0, T_c=1.2))
        else:
            kr = T.recombination_kernel_base(E, gap, 440.0, 1.2)[None] if recombination else None
            ks = T.scattering_kernel_base(E, gap, 440.0, 1.2)[None] if scattering else None
            self.tab = eng.make_collision_tables(kr, ks, rho[None], idx_d, idx_s, sg)
        self.en_r, self.en_s, self.upd = recombination, scattering, dynamic_phonons
        w = rho / (np.sum(rho) * dE)
        npix = members * N * NX
        self.npix = npix
        if init_occupation is None:
            init = np.concatenate([1e-4 * (1.0 + np.random.default_rng(1000 + m if members > 1 else 0).random(N * NX))
                                   for m in range(members)])
            self.state = torch.as_tensor(w[:, None] * init[None, :], device=eng.device)     # [NE][members*ncell]
        else:
            self.state = torch.as_tensor(w, device=eng.device)[:, None] * init_occupation.reshape(1, -1)
        self.alt = torch.empty_like(self.state)
        nph = T.thermal_phonon_occupation(om, 0.1)
        self.phonon = torch.as_tensor(np.repeat(nph[:, None], npix, axis=1), device=eng.device)
        self.coll_flags = torch.full((npix,), 16, dtype=torch.uint8, device=eng.device)   # every pixel interior
        self.op = DiffusionOperator(eng, ne * members, dt,
                                    dcoef=np.repeat(T.diffusion_coefficients(E, gap, D0), members))
        self.grid = [N, NX]
        self.cell_updates_per_step = float(N) * NX * ne * members
        planes_rw = (ne + self.nw) + (ne + (self.nw if dynamic_phonons else 0))
        self.coll_bytes_per_call = 8.0 * planes_rw * npix
        self.bytes_per_step = 48.0 * self.cell_updates_per_step + 2 * self.coll_bytes_per_call
        self.path = f"rect-tiled ADI + {self.tab['kernel']} collision kernel"
        ens = f"{members} independent members of " if members > 1 else ""
        self.description = (f"{label}{ens}{N}x{NX} fp64, NE={ne}, Nw={self.nw}: Strang C(dt/2) D(dt) C(dt/2) + Pauli guard per step; "
                            f"recombination={'on' if recombination else 'off'}, scattering={'on' if scattering else 'off'}, "
                            f"phonons {'dynamic' if dynamic_phonons else 'frozen'}; reflective walls, D0=6 dt=0.1 dx=1")
        self.max_occ = 0.0

    def _collide(self, dtc, guarded: bool = False):
        """One collision call over all members; ``guarded``: the Pauli guard of the new state is reduced by the same call
        (as `run_2d_crank_nicolson` does for the last collision of a step) and a read-back ticket is returned."""
        eng = self.eng
        ticket = None
        if guarded:
            ticket = eng.collide_guarded(self.tab, self.state, self.alt, self.phonon, self.dE, dtc, self.en_r, self.en_s,
                                         self.upd, 1e-18, ncell=self.npix, flags=self.coll_flags)
        else:
            import ctypes as C
            from . import _hip
            acc = None
            if self.upd and not self.tab["fast"]:
                acc = eng.scratch("coll_acc", 2 * self.tab["nw"] * self.npix)
            elif self.upd and self.tab["merged_slots"] and self.en_r and self.en_s:
                acc = eng.scratch("coll_acc", 2 * self.tab["merged_slots"] * self.npix)
            _hip.check(eng.lib.qp_collision_step(C.byref(self.tab["struct"]), int(self.coll_flags.data_ptr()), self.npix,
                                                 int(self.state.data_ptr()), int(self.alt.data_ptr()),
                                                 int(self.phonon.data_ptr()), 0 if acc is None else int(acc.data_ptr()),
                                                 float(self.dE), float(dtc), int(self.en_r), int(self.en_s), int(self.upd),
                                                 eng.stream), "qp_collision_step")
        self.state, self.alt = self.alt, self.state
        return ticket

    def _guard_launch(self):
        """Pauli guard over all members (device reduction + asynchronous read-back, as `run_2d_crank_nicolson` does)."""
        return self.eng.pauli_stats_launch(self.state, self.tab, 1e-18, ncell=self.npix, flags=self.coll_flags)

    def _guard_check(self, ticket):
        mx, _, forb = self.eng.pauli_stats_result(ticket)
        if forb is not None or mx > 1.0:
            raise ValueError("Pauli guard tripped in the benchmark state")
        self.max_occ = max(self.max_occ, mx)

    def _collide_pair(self, dt_a, dt_b):
        """Closing half-step of one step + guard + opening half-step of the next as one pass (what `run_2d_crank_nicolson`
        issues between store points when the tables allow it)."""
        ticket = self.eng.collide_pair_guarded(self.tab, self.state, self.alt, self.phonon, self.dE, dt_a, dt_b, 0.0,
                                               self.en_r, self.en_s, self.upd, 1e-18, ncell=self.npix, flags=self.coll_flags)
        self.state, self.alt = self.alt, self.state
        return ticket

    def run(self, k: int):
        """k steps C(dt/2) D(dt) C(dt/2) + guard; the guard of step s is read on the host after step s + GUARD_LAG has been
        enqueued (every step is checked, the last ones before returning) - as `run_2d_crank_nicolson` does.  Between two
        steps the closing and the opening half-step run as one pass over the state where the tables allow it
        (`tab["pair"]`: NE = 8, 12, one gap class, no merged bins) - the same 2 k half-steps, 2 k + 1 ... k + 1 launches."""
        pending = []
        pair = bool(self.tab.get("pair"))
        opened = False
        for s in range(k):
            if not opened:
                self._collide(0.5 * self.dt)
            self.eng.adi_steps(self.op, self.state, 1)
            if pair and s + 1 < k:
                pending.append(self._collide_pair(0.5 * self.dt, 0.5 * self.dt))
                opened = True
            else:
                pending.append(self._collide(0.5 * self.dt, guarded=True))
                opened = False
            while len(pending) > self.eng.GUARD_LAG:
                self._guard_check(pending.pop(0))
        for ticket in pending:
            self._guard_check(ticket)

    def _pmc_traffic(self, pair: bool):
        """HBM bytes per launch of the dominant collision kernel from the committed PMC summary of `--workload c3`
        (profiles/r03_c3_pmc.json; separate rocprofv3 --pmc passes of this same command)."""
        import json
        from pathlib import Path
        f = Path(__file__).resolve().parents[2] / "profiles" / "r03_c3_pmc.json"
        if not (self.N == 4096 and self.ne == 12 and self.members == 1 and self.upd and self.en_s and f.exists()):
            return None
        want = "collision_pair_kernel<12, true, true, true>" if pair else "collision_diag_kernel<12, true, true, true, false>"
        for name, v in json.loads(f.read_text())["kernels"].items():
            if want in name:
                return v["hbm_bytes_per_launch"]
        return None

    def roofline(self, nrep: int) -> dict:
        """Dominant kernel = the collision update: HIP-event time per launch vs the ALGORITHMIC bytes of the pixel-updates
        that launch performs (16 (NE + Nw) B per pixel-update).  Where the time loop runs the double half-step
        pass (`tab["pair"]`) one launch performs TWO pixel-updates per pixel, so its algorithmic bytes are twice a
        call's - while the bytes it actually moves are those of one (the intermediate state stays in registers): the
        `traffic` counter then reads about half of `bytes_per_launch`."""
        import os
        torch = self.eng.torch
        dev = self.eng.device
        pair = bool(self.tab.get("pair"))
        step = (lambda: self._collide_pair(0.5 * self.dt, 0.5 * self.dt)) if pair else (lambda: self._collide(0.5 * self.dt))
        updates = 2 if pair else 1
        step()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        for _ in range(nrep):
            step()
        ev1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        per_call = ev0.elapsed_time(ev1) * 1e-3 / nrep
        nbytes = updates * self.coll_bytes_per_call
        achieved = nbytes / per_call / 1e9
        pairs = self.ne * self.ne
        flops = updates * 26.0 * pairs * self.npix     # ~26 NE^2 flop per pixel-update
        tflops = flops / per_call / 1e12
        classes_onepass = (self.tab.get("gap_sq") is not None and self.ne == 50
                           and int(self.tab["struct"].nclass) <= 16)      # qp_collision_onepass.hip: gap-class form
        onepass = (self.tab.get("ks0_diag") is not None or self.tab.get("kr0_anti2") is not None or classes_onepass) \
            and os.environ.get("QPSIM_COLL_ONEPASS", "1") != "0"
        kernel = {"register": ("collision_pair_kernel (two half-steps per launch, intermediate state in registers)" if pair else
                               "collision_onepass_kernel (one launch, tables staged in LDS)" if onepass else
                               "collision_diag_kernel" if self.ne < 32 else
                               "collision_range_kernel x2 + collision_phonon_kernel (one call)"),
                  "wave": "collision_wave_kernel", "generic": "collision_generic_kernel"}[self.tab["kernel"]]
        traffic, source = self._pmc_traffic(pair), None
        if traffic is not None:
            source = "profiles/r03_c3_pmc.json (committed rocprofv3 --pmc passes of this workload, not measured in this run)"
        common = {"traffic": traffic, "traffic_source": source, "kernel": kernel, "bytes_per_launch": nbytes,
                  "pixel_updates_per_launch": updates * self.npix,
                  "flops_per_launch": flops, "avg_launch_us": per_call * 1e6,
                  "pixel_updates_per_s": updates * self.npix / per_call,
                  "hbm_gbs": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
                  # what the launch moves by its own accounting: every plane read / written once per LAUNCH
                  "bytes_moved_per_launch": self.coll_bytes_per_call,
                  "hbm_frac_of_bytes_moved": self.coll_bytes_per_call / per_call / 1e9 / HBM_PEAK_GBS,
                  "fp64_tflops": tflops,
                  "fp64_frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                  "note": (f"algorithmic bytes: 16*(NE+Nw) B per pixel-update when phonons are dynamic, x {updates} "
                           f"pixel-update(s) per pixel and launch; ~26*NE^2 = {26 * pairs} flop per pixel-update"
                           + ("; this launch is the double half-step pass: `frac` prices its two pixel-updates at the per-call "
                              "byte model although it moves the planes only once (`bytes_moved_per_launch`, "
                              "`hbm_frac_of_bytes_moved`) - the saving is the point of the fusion, the kernel itself is "
                              "bound by the fp64 units (`fp64_frac`)" if pair else ""))}
        if self.ne >= 20:      # 26 NE^2 flop against 16 (NE + Nw) B per pixel: above NE ~ 20 the fp64 vector rate bounds it
            return {"bound": "fp64", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tflops / FP64_VECTOR_PEAK_TFLOPS, **common}
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                **common}


class DecomposedADIWorkload:
    """BASELINE configs[4]: one N x N scalar field cut into a py x px grid of blocks, one block per rank, neighbour
    exchange of one reduced-rhs row per sweep over torch.distributed (RCCL).  Strong scaling: the grid is fixed."""

    def __init__(self, N: int, device):
        import torch.distributed as dist
        from .distributed import BlockTopology, HipBlockBackend, TorchDistTransport, choose_process_grid
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        py, px = choose_process_grid(self.world, N, N)
        self.topo = BlockTopology(N, N, py, px, rank)
        self.be = HipBlockBackend(self.topo, 1.0, 0.1, [6.0], [0.0] * 4, [0.0] * 4, device=device)
        self.transport = TorchDistTransport() if self.world > 1 else None
        j0, i0, ny, nx = self.topo.block
        rng = np.random.default_rng(1000 + rank)
        self.be.u.copy_(self.be.torch.as_tensor(1e-4 * (1.0 + rng.random((1, ny * nx))), device=self.be.device))
        self.N, self.nfield, self.grid = N, 1, [N, N]
        self.cell_updates_per_step = float(N) * N / self.world    # bench multiplies by world
        self.bytes_per_step = 32.0 * self.cell_updates_per_step
        self.path = f"rect-tiled partition ADI, {py}x{px} blocks of {ny}x{nx}, point-to-point interface rows"
        self.description = (f"{N}x{N} fp64 CN-ADI step, domain-decomposed {py}x{px} (one block per GPU), reflective walls, "
                            "D=6 dt=0.1 dx=1")
        self.scaling = "strong"

    def run(self, k: int):
        from .distributed import LocalTransport, block_adi_steps
        transport = self.transport
        if transport is None:
            class _Nop:
                def exchange(self, sends, recvs):
                    assert not sends and not recvs
            transport = _Nop()
        block_adi_steps(self.be, self.topo, transport, k)

    def roofline(self, nrep: int) -> dict:
        torch = self.be.torch
        dev = self.be.device
        k = 10
        self.run(2)
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        for _ in range(nrep):
            self.run(k)
        ev1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        per_sweep = ev0.elapsed_time(ev1) * 1e-3 / (nrep * (2 * k + 1))
        j0, i0, ny, nx = self.topo.block
        bytes_per_launch = 16.0 * ny * nx
        achieved = bytes_per_launch / per_sweep / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": "rect_x_kernel / rect_y_kernel on the local block (rank 0)",
                "bytes_per_launch": bytes_per_launch, "avg_launch_us": per_sweep * 1e6,
                "note": "per-rank sweep time including the neighbour exchange that follows it"}


def _global_field(torch, j0: int, i0: int, ny: int, nx: int, device):
    """Deterministic synthetic field 1e-4 (1 + hash(j, i)) evaluated on global coordinates, so that every rank fills its own
    cells AND its halos consistently without talking to anybody."""
    j = torch.arange(j0, j0 + ny, dtype=torch.float64, device=device)[:, None]
    i = torch.arange(i0, i0 + nx, dtype=torch.float64, device=device)[None, :]
    h = torch.sin(j * 12.9898 + i * 78.233) * 43758.5453
    return 1e-4 * (1.0 + (h - torch.floor(h)))


class OverlapDecomposedWorkload:
    """BASELINE configs[4] (8192 x 8192 on 2 x 4 GPUs) with the overlapped-halo decomposition: every rank runs the
    single-GPU kernels on its block + 64-cell halos and the halos are refreshed over RCCL once every S steps
    (``distributed.halo_steps_bound``); ``coupled=True`` adds the collision half-steps (full physics, NE = 12) on the
    decomposed grid.  Strong scaling: the global grid is fixed."""

    def __init__(self, N: int, device, coupled: bool = False, steps_per_exchange=None, topo=None, halo: int | None = None):
        """``topo``: a ``BlockTopology`` makes this the block of one VIRTUAL rank (several of them in one process, driven by
        ``distributed.lockstep_overlap_steps``: tests on one GPU); default: the rank of this process in torch.distributed.
        ``halo``: halo width in cells (default ``QPSIM_DD_HALO`` or 64; 128 quadruples the steps between refreshes)."""
        import os
        import torch.distributed as dist
        from .distributed import (HALO, BlockTopology, HipHaloPacking, HipOverlapBlock, OverlapBlock, TorchDistTransport,
                                  choose_process_grid)
        halo = int(os.environ.get("QPSIM_DD_HALO", HALO)) if halo is None else int(halo)
        if topo is None:
            self.world = dist.get_world_size() if dist.is_initialized() else 1
            rank = dist.get_rank() if dist.is_initialized() else 0
            py, px = choose_process_grid(self.world, N, N)
            self.topo = BlockTopology(N, N, py, px, rank)
            self.transport = TorchDistTransport() if self.world > 1 else None
        else:
            self.topo, self.world, self.transport = topo, topo.py * topo.px, None
            py, px = topo.py, topo.px
        self.coupled = coupled
        j0, i0, ny, nx = self.topo.block
        if not coupled:
            self.block = HipOverlapBlock(self.topo, 1.0, 0.1, [6.0], [0.0] * 4, [0.0] * 4, device=device, halo=halo,
                                         steps_per_exchange=steps_per_exchange)
            torch = self.block.torch
            ej, ei = self.block.ext_origin()
            self.block.u.copy_(_global_field(torch, ej, ei, self.block.ey, self.block.ex, self.block.device)[None])
            self.nfield, ne = 1, 1
            self.device = self.block.device
        else:
            import torch
            torch_mod = torch
            ne = 12
            E, _ = T.build_energy_grid(180.0, 1.0, 3.0, ne)
            dmax = float(np.max(T.diffusion_coefficients(E, 180.0, 6.0)))
            # collision half-steps between the sweeps: same locality argument, half the cadence for margin
            probe = OverlapBlock(self.topo, ne, 0.5 * 0.1 * dmax, halo=halo, steps_per_exchange=steps_per_exchange)
            spe = max(1, probe.steps_per_exchange // 2)
            dev = torch.device(device)
            ej, ei = probe.ext_origin()
            self.inner = CoupledWorkload(probe.ey, dev, nx=probe.ex, init_occupation=_global_field(
                torch, ej, ei, probe.ey, probe.ex, dev).reshape(-1))
            outer = self

            class _Block(HipHaloPacking, OverlapBlock):
                torch = torch_mod

                @property
                def u(self):       # the collision calls swap state / alt: always the current quasiparticle planes
                    return outer.inner.state.view(ne, self.ey, self.ex)

                @u.setter
                def u(self, value):
                    pass

                def advance(self, nsteps):
                    outer.inner.run(nsteps)

            self.block = _Block(self.topo, ne, 0.5 * 0.1 * dmax, halo=halo, steps_per_exchange=spe)
            self.nfield = ne
            self.device = dev
        self.N, self.grid = N, [N, N]
        self.cell_updates_per_step = float(N) * N * ne / self.world        # own cells only; bench multiplies by world
        self.bytes_per_step = 32.0 * self.cell_updates_per_step
        b = self.block
        self.halo_overhead = b.ey * b.ex / float(ny * nx) - 1.0
        self.path = (f"overlapped-halo decomposition {py}x{px}: block {ny}x{nx} + halo {b.halo} = {b.ey}x{b.ex} per rank "
                     f"(+{100 * self.halo_overhead:.1f} % cells), halo refresh every {b.steps_per_exchange} steps "
                     f"(one round: side strips + corner blocks packed into one send buffer, point-to-point over RCCL), "
                     f"rect-tiled partition ADI"
                     + (", fine tiles (32-cell chunks)" if getattr(getattr(b, "plan", None), "fine", False) else "")
                     + (" + register collision kernel" if coupled else ""))
        self.description = (f"{N}x{N} fp64 " + ("coupled step C(dt/2) D(dt) C(dt/2), NE=12, " if coupled else "CN-ADI step, ")
                            + f"domain-decomposed {py}x{px} (one block per GPU), reflective walls, D=6 dt=0.1 dx=1")
        self.scaling = "strong"

    def run(self, k: int, exchange: bool = True):
        from .distributed import overlap_adi_steps
        overlap_adi_steps(self.block, self.transport, k, exchange=exchange)

    def roofline(self, nrep: int) -> dict:
        torch = self.block.torch if not self.coupled else self.inner.eng.torch
        dev = self.device
        if self.coupled:
            return self.inner.roofline(nrep)
        k = max(1, min(10, self.block.steps_per_exchange))
        self.block.advance(2)
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
        for _ in range(nrep):
            self.block.advance(k)
        ev1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        per_sweep = ev0.elapsed_time(ev1) * 1e-3 / (nrep * (2 * k + 1))
        bytes_per_launch = 16.0 * self.block.ey * self.block.ex
        achieved = bytes_per_launch / per_sweep / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel": ("fine_x_kernel / fine_y_kernel" if getattr(getattr(self.block, "plan", None), "fine", False)
                           else "rect_x_kernel / rect_y_kernel")
                + " on the extended local block (rank 0)",
                "bytes_per_launch": bytes_per_launch, "avg_launch_us": per_sweep * 1e6,
                "note": "per-rank sweep on block + halos; the halo cells are redundant work and are NOT counted in `value`"}


def build(name: str, device):
    m = re.fullmatch(r"dd(\d+)(c?)", name)
    if m:   # dd<N>: scalar field, overlapped-halo decomposition; dd<N>c: coupled (collisions + ADI, NE = 12)
        return OverlapDecomposedWorkload(int(m.group(1)), device, coupled=bool(m.group(2)))
    m = re.fullmatch(r"ddx(\d+)", name)
    if m:   # ddx<N>: exact interface exchange after every sweep (qp_adi_rect_phase), the scheme for stiff steps
        return DecomposedADIWorkload(int(m.group(1)), device)
    m = re.fullmatch(r"adi(\d+)", name)
    if m:
        return ADIWorkload(int(m.group(1)), device)
    m = re.fullmatch(r"cn(\d+)", name)
    if m:   # cn<N>: the default scheme (unsplit CN) on an N x N rectangle
        return ExactCNWorkload(int(m.group(1)), device)
    m = re.fullmatch(r"ring(\d+)(?:x(\d+))?", name)
    if m:   # ring<N>[x<F>]: annulus mask inside N x N (masked tiled path), F fields
        return ADIWorkload(int(m.group(1)), device, nfield=int(m.group(2) or 1), ring=True)
    m = re.fullmatch(r"adi(\d+)x(\d+)", name)
    if m:   # adi<N>x<F>: F independent fields of N x N (ensemble batch)
        return ADIWorkload(int(m.group(1)), device, nfield=int(m.group(2)))
    if name == "c2":
        return CoupledWorkload(1024, device, recombination=True, scattering=False, dynamic_phonons=False,
                               label="BASELINE configs[1]: ")
    if name == "c4":   # BASELINE configs[3]: 512 members over 8 GPUs -> 64 independent 256^2 pixels-arrays per GPU
        return CoupledWorkload(256, device, recombination=True, scattering=True, dynamic_phonons=True, members=64,
                               label="BASELINE configs[3] (per-GPU share, 64 of 512 members): ")
    if name == "c3":
        return CoupledWorkload(4096, device, recombination=True, scattering=True, dynamic_phonons=True,
                               label="BASELINE configs[2]: ")
    m = re.fullmatch(r"coupled(\d+)(?:ne(\d+))?(?:gap(\d+))?", name)
    if m:   # coupled<N>[ne<NE>][gap<K>]: full physics on N x N with NE energy bins (ne50 = the reference's default
        # resolution), K gap classes (non-uniform gap; collisions only - the diffusivity stays uniform)
        ne = int(m.group(2) or 12)
        return CoupledWorkload(int(m.group(1)), device, ne=ne, fmax=3.0 if ne <= 24 else 10.0,
                               gap_classes=int(m.group(3) or 1))
    raise ValueError(f"unknown workload '{name}'")
