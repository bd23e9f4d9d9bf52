"""MI355X-native drop-in for ``qpsim.solver``: same run / step API, HIP kernels underneath.

``run_2d_crank_nicolson`` keeps the reference's signature, defaults, return tuple, error types and
messages (``/root/reference/qpsim/solver.py:999-1587``); the state lives on the GPU for the whole run
and only crosses PCIe at store points.  Additive keyword arguments (all optional):

``diffusion_scheme``
    ``"cn_exact"`` (default) reproduces the reference's unsplit Crank-Nicolson step to ``cn_rtol`` by an
    ADI-preconditioned iteration built from the same two sweep kernels; ``"adi"`` takes the single
    Peaceman-Rachford step (the benchmarked kernel; identical on one-cell-thick strips, O(dt^3) splitting
    error per step on true 2-D grids).
``device``
    torch device (default: current CUDA/HIP device).

Table builders and helpers the reference exports from ``qpsim.solver`` are re-exported here under
the same names.
"""
from __future__ import annotations

import warnings
from typing import Any, Callable

import numpy as np

from . import tables as _tb
from .engine import BoundaryAssignmentError, DiffusionOperator, Engine, compile_geometry
from .models import (
    BoundaryCondition,
    EdgeSegment,
    ExternalGenerationSpec,
    InitialConditionSpec,
    SimulationParameters,
    normalize_collision_solver_name,
)
from .safe_eval import compile_safe_expression

# names the reference exposes from qpsim.solver
build_energy_grid = _tb.build_energy_grid
integration_widths_from_centers = _tb.integration_widths_from_centers
_bcs_density_of_states = _tb.bcs_density_of_states
_dynes_density_of_states = _tb.dynes_density_of_states
thermal_phonon_occupation = _tb.thermal_phonon_occupation
thermal_qp_weights = _tb.thermal_qp_weights
recombination_kernel_base = _tb.recombination_kernel_base
scattering_kernel_base = _tb.scattering_kernel_base
recombination_kernel = _tb.recombination_kernel
scattering_kernel = _tb.scattering_kernel
_build_phonon_frequency_map = _tb.build_phonon_frequency_map
_KB_UEV_PER_K = _tb.KB_UEV_PER_K

__all__ = [
    "BoundaryAssignmentError", "run_2d_crank_nicolson", "reconstruct_field", "build_fixed_phonon_history",
    "evaluate_external_generation", "build_energy_grid", "integration_widths_from_centers",
    "thermal_phonon_occupation", "thermal_qp_weights", "recombination_kernel_base", "scattering_kernel_base",
    "recombination_kernel", "scattering_kernel", "apply_collision_step_fischer_catelani_uniform",
    "apply_collision_step_fischer_catelani_nonuniform", "apply_scattering_step", "apply_recombination_step",
    "build_laplacian_with_boundaries", "build_variable_diffusion_laplacian",
]


def reconstruct_field(mask: np.ndarray, values: np.ndarray) -> np.ndarray:
    """Interior values -> NaN-padded [ny, nx] frame (solver.py:215-218)."""
    out = np.full(mask.shape, np.nan, dtype=float)
    out[mask] = values
    return out


def build_fixed_phonon_history(*, mask: np.ndarray, times, bath_temperature: float,
                               phonon_energy_bins: np.ndarray | None = None):
    """Constant-temperature phonon outputs aligned with stored times (solver.py:373-426)."""
    mask_b = np.asarray(mask, dtype=bool)
    n = int(mask_b.sum())
    if n == 0:
        raise ValueError("Geometry mask has no interior points.")
    if len(times) <= 0:
        raise ValueError("times must contain at least one stored timepoint.")
    base = reconstruct_field(mask_b, np.full(n, float(bath_temperature)))
    frames = [base.copy() for _ in range(len(times))]
    eframes = None
    bins = None
    if phonon_energy_bins is not None:
        bins = np.asarray(phonon_energy_bins, dtype=float).copy()
        if bins.ndim != 1:
            raise ValueError("phonon_energy_bins must be a 1D array.")
        if np.any(~np.isfinite(bins)):
            raise ValueError("phonon_energy_bins must contain only finite values.")
        if np.any(bins < 0):
            raise ValueError("phonon_energy_bins must be non-negative.")
        per_bin = [reconstruct_field(mask_b, np.full(n, float(v)))
                   for v in thermal_phonon_occupation(bins, float(bath_temperature))]
        eframes = [[f.copy() for f in per_bin] for _ in range(len(times))]
    meta = {"mode": "fixed_temperature", "phonon_temperature_K": float(bath_temperature), "field_units": "K",
            "energy_frame_units": "occupation", "omega_bins_match_qp_energy_bins": bool(phonon_energy_bins is not None)}
    return frames, eframes, bins, meta


def _validated_generation(rates: np.ndarray, mode: str, shape: tuple[int, int]) -> np.ndarray:
    """Shape / finiteness / sign checks every generation mode goes through (messages of solver.py:889-902)."""
    if rates.shape != shape:
        raise ValueError(f"External generation mode '{mode}' returned invalid shape {rates.shape}; expected {shape}.")
    if not np.isfinite(rates).all():
        raise ValueError(f"External generation mode '{mode}' produced non-finite values.")
    if (rates < 0).any():
        raise ValueError(f"External generation mode '{mode}' produced negative values. "
                         "Generation rates must be non-negative.")
    return rates


class _CustomGeneration:
    """A ``custom`` generation expression g(E, x, y, t, params) compiled once, evaluated on [NE, n_spatial].

    Three evaluation strategies, tried in this order, each falling through to the next on ANY exception:
      grid      one call with E as a column and x, y as rows - only for expressions that are elementwise in E, x, y
                (``safe_eval.expression_is_elementwise``), where broadcasting cannot change a value;
      per bin   one call per energy bin with E a scalar and x, y arrays: the reference's vectorised attempt
                (solver.py:933-948), incl. its rule that a bin yields a scalar or exactly n_spatial values;
      per cell  one call per (bin, pixel) with plain floats: the reference's scalar fallback (solver.py:949-961), whose
                exceptions are the caller's.
    x, y are the normalised pixel-centre coordinates of the interior cells in argwhere order (solver.py:924-929)."""

    def __init__(self, spec: ExternalGenerationSpec, mask: np.ndarray):
        from .safe_eval import expression_is_elementwise
        body = spec.custom_body.strip() or "0.0"
        self._fn = compile_safe_expression(body, variable_names=("E", "x", "y", "t", "params"))
        self._gridwise = expression_is_elementwise(body, array_variables=("E", "x", "y"))
        self._params = dict(spec.custom_params or {})
        mask = np.asarray(mask, dtype=bool)
        ny, nx = mask.shape
        rows, cols = np.nonzero(mask)
        self._x = (cols + 0.5) / max(1, nx)
        self._y = (rows + 0.5) / max(1, ny)

    def _on_grid(self, E: np.ndarray, t: float, shape) -> np.ndarray:
        value = self._fn(E=E[:, None], x=self._x[None, :], y=self._y[None, :], t=t, params=self._params)
        return np.array(np.broadcast_to(np.asarray(value, dtype=float), shape))

    def _per_bin(self, E: np.ndarray, t: float, shape) -> np.ndarray:
        n = shape[1]
        out = np.empty(shape, dtype=float)
        for row, energy in zip(out, E):
            value = np.asarray(self._fn(E=energy, x=self._x, y=self._y, t=t, params=self._params), dtype=float)
            if value.ndim and value.size != n:
                raise ValueError("Vectorized custom generation must return a scalar or "
                                 f"exactly {n} values per energy bin; got {value.size}.")
            row[...] = value.reshape(-1) if value.ndim else float(value)
        return out

    def _per_cell(self, E: np.ndarray, t: float, shape) -> np.ndarray:
        cells = list(zip(self._x.tolist(), self._y.tolist()))
        return np.array([[float(self._fn(E=energy, x=x, y=y, t=t, params=self._params)) for x, y in cells]
                         for energy in E.tolist()], dtype=float).reshape(shape)

    def __call__(self, E_bins: np.ndarray, t: float, n_spatial: int) -> np.ndarray:
        E = np.asarray(E_bins, dtype=float)
        shape = (E.size, int(n_spatial))
        for attempt in ((self._on_grid,) if self._gridwise else ()) + (self._per_bin,):
            try:
                return attempt(E, t, shape)
            except Exception:      # noqa: BLE001 - any failure of a vectorised attempt means "try the next strategy"
                continue
        return self._per_cell(E, t, shape)


def evaluate_external_generation(spec: ExternalGenerationSpec, E_bins: np.ndarray, n_spatial: int, t: float,
                                 mask: np.ndarray, _compiled: "_CustomGeneration | None" = None) -> np.ndarray | None:
    """g_ext(E, x, t) as an [NE, n_spatial] host array, None for mode "none" (semantics of solver.py:878-964).

    The run loop only calls this for ``custom`` mode (constant / pulse rates are added on the device) and passes the
    expression it compiled once per run as ``_compiled``.
    """
    mode = spec.mode.strip().lower()
    shape = (len(E_bins), int(n_spatial))
    if mode == "constant":
        level = spec.rate
    elif mode == "pulse":
        level = spec.pulse_rate if spec.pulse_start <= t < spec.pulse_start + spec.pulse_duration else 0.0
    elif mode == "custom":
        custom = _compiled if _compiled is not None else _CustomGeneration(spec, mask)
        return _validated_generation(custom(E_bins, t, n_spatial), mode, shape)
    else:                      # "none" and anything validate() would have rejected
        return None
    return _validated_generation(np.full(shape, level, dtype=float), mode, shape)


def _crop_to_bounding_box(mask: np.ndarray, edges: list[EdgeSegment]):
    """(mask, edges) restricted to the bounding box of the interior cells; faces are re-indexed, ids kept."""
    rows = np.flatnonzero(mask.any(axis=1))
    cols = np.flatnonzero(mask.any(axis=0))
    r0, r1, c0, c1 = int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1
    if (r0, c0) == (0, 0) and (r1, c1) == mask.shape:
        return mask, edges
    from .models import BoundaryFace
    shifted = [EdgeSegment(e.edge_id, e.x0 - c0, e.y0 - r0, e.x1 - c0, e.y1 - r0, e.normal,
                           [BoundaryFace(f.row - r0, f.col - c0, f.direction) for f in e.faces]) for e in edges]
    return np.ascontiguousarray(mask[r0:r1, c0:c1]), shifted


def _device_frames_async(eng, planes, mask: np.ndarray):
    """Device planes -> ticket for host [n, ny, nx] frames on the FULL mask, NaN outside the interior (reconstruct_field
    semantics).  Padding and, for a cropped engine grid, the embedding into the full frame are done on the device; the copy
    to the host runs on a side stream into pinned memory while the time loop goes on (``ticket.result()`` waits for it)."""
    if (eng.ny, eng.nx) == mask.shape:
        return eng.download_frames_async(planes)
    r0 = int(np.flatnonzero(mask.any(axis=1))[0])
    c0 = int(np.flatnonzero(mask.any(axis=0))[0])
    return eng.download_frames_async(planes, full_shape=mask.shape, offset=(r0, c0))


def _device_frames(eng, planes, mask: np.ndarray) -> np.ndarray:
    return _device_frames_async(eng, planes, mask).result()


class _LazyOutputs:
    """Store points enqueue their downloads and go on; the host arrays are filled in when the copies have landed (when a
    staging slot is recycled, when a progress callback needs the frame, or before the run returns)."""

    def __init__(self):
        self._pending: list = []

    def add(self, ticket, consume) -> None:
        self._pending.append((ticket, consume))

    def flush(self) -> None:
        while self._pending:
            ticket, consume = self._pending.pop(0)
            consume(ticket.result())


def _step_plan(total_time: float, dt: float) -> tuple[int, float, int]:
    """(full steps, remainder dt or 0, total steps) (solver.py:1085-1089)."""
    full = int(np.floor(total_time / dt + 1e-12))
    rem = float(total_time - full * dt)
    if rem < 1e-12:
        rem = 0.0
    return full, rem, full + (1 if rem > 0.0 else 0)


def _color_limits(frames: list[np.ndarray]) -> list[float]:
    stack = np.stack(frames)
    lo, hi = float(np.nanmin(stack)), float(np.nanmax(stack))
    if abs(hi - lo) < 1e-12:
        hi = lo + 1e-9
    return [lo, hi]


def _notify(cb, t: float, frame: np.ndarray) -> None:
    if cb is None:
        return
    try:
        cb(float(t), np.array(frame, copy=True))
    except Exception:
        pass


class _Diffuser:
    """Per-run diffusion operators (regular dt and the optional short last step)."""

    def __init__(self, eng: Engine, nfield: int, dt: float, rem: float, scheme: str, rtol: float,
                 dcoef=None, dfield=None):
        if scheme not in ("cn_exact", "adi"):
            raise ValueError("diffusion_scheme must be 'cn_exact' or 'adi'.")
        self.eng, self.scheme, self.rtol = eng, scheme, rtol
        self.op = DiffusionOperator(eng, nfield, dt, dcoef=dcoef, dfield=dfield)
        self.op_last = DiffusionOperator(eng, nfield, rem, dcoef=dcoef, dfield=dfield) if rem > 0.0 else None
        self.iterations = 0

    def step(self, u, final: bool) -> None:
        op = self.op_last if final else self.op
        if op is None:
            raise RuntimeError("Internal error: final-step solver not initialized.")
        if self.scheme == "adi":
            self.eng.adi_step(op, u)
        else:
            self.iterations += self.eng.cn_exact_step(op, u, rtol=self.rtol)

    def advance(self, u, first: int, last: int, full_steps: int) -> None:
        """Steps ``first..last`` (1-based, inclusive) with nothing in between, as in the reference's scalar loop
        (solver.py:1540-1555).  With the ADI scheme the regular steps of the stretch are ONE library call: the carried
        right-hand side is never materialised in between (32 B instead of 48 B per cell-update)."""
        if self.scheme != "adi":
            for step in range(first, last + 1):
                self.step(u, step > full_steps)
            return
        regular = min(last, full_steps) - first + 1
        if regular > 0:
            self.eng.adi_steps(self.op, u, regular)
        if last > full_steps:
            self.step(u, True)


def _on_run_device(fn):
    """Runs ``fn`` with the HIP device of its ``device=`` argument current in the calling thread (kernel launches through
    the C ABI go to the thread's current device; the reference's GUI calls the solver from a worker thread)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        try:
            import torch
            usable = torch.cuda.is_available()
        except Exception:            # no torch / no GPU: let the body raise its own error after argument validation
            usable = False
        if not usable:
            return fn(*args, **kwargs)
        dev = kwargs.get("device")
        dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


@_on_run_device
def run_2d_crank_nicolson(
    mask: np.ndarray,
    edges: list[EdgeSegment],
    edge_conditions: dict[str, BoundaryCondition],
    initial_field: np.ndarray,
    diffusion_coefficient: float,
    dt: float,
    total_time: float,
    dx: float,
    store_every: int = 1,
    energy_gap: float = 0.0,
    energy_min_factor: float = 1.0,
    energy_max_factor: float = 10.0,
    num_energy_bins: int = 50,
    energy_weights: np.ndarray | None = None,
    enable_diffusion: bool = True,
    enable_recombination: bool = False,
    enable_scattering: bool = False,
    dynes_gamma: float = 0.0,
    collision_solver: str = "fischer_catelani_local",
    tau_0: float = 440.0,
    tau_s: float | None = None,
    tau_r: float | None = None,
    T_c: float = 1.2,
    bath_temperature: float = 0.1,
    external_generation: ExternalGenerationSpec | None = None,
    initial_condition_spec: InitialConditionSpec | None = None,
    gap_expression: str = "",
    precomputed: dict | None = None,
    pauli_warn_threshold: float | None = 0.5,
    pauli_error_threshold: float | None = 1.0,
    enforce_pauli: bool = True,
    pauli_density_floor: float = 1e-18,
    freeze_phonon_dynamics: bool = False,
    phonon_history_out: dict[str, Any] | None = None,
    progress_callback: Callable[[float, np.ndarray], None] | None = None,
    *,
    diffusion_scheme: str = "cn_exact",
    cn_rtol: float = 1e-13,
    device=None,
):
    """Run the 2-D Crank-Nicolson diffusion (+ local collisions) time loop on the GPU.

    Returns ``(times, frames, mass, color_limits, energy_frames_or_None, energy_bins_or_None)`` exactly
    as the reference does; ``phonon_history_out`` is cleared and filled the same way.
    """
    mask = np.asarray(mask, dtype=bool)
    initial_field = np.asarray(initial_field)
    if dt <= 0 or total_time <= 0:
        raise ValueError("dt and total_time must be positive.")
    if enable_diffusion and diffusion_coefficient <= 0:
        raise ValueError("Diffusion coefficient must be positive.")
    if store_every <= 0:
        store_every = 1
    if initial_field.shape != mask.shape:
        raise ValueError("Initial field shape must match mask shape.")
    n = int(np.sum(mask))
    if n == 0:
        raise ValueError("Geometry mask has no interior points.")
    if phonon_history_out is not None:
        phonon_history_out.clear()
    tau_s_eff = float(tau_s if tau_s is not None else tau_0)
    tau_r_eff = float(tau_r if tau_r is not None else tau_0)
    if enable_scattering and tau_s_eff <= 0:
        raise ValueError("tau_s must be positive when scattering is enabled.")
    if enable_recombination and tau_r_eff <= 0:
        raise ValueError("tau_r must be positive when recombination is enabled.")
    if external_generation is not None:
        external_generation.validate()

    # Device grid = bounding box of the mask.  Interior cells keep their argwhere (row-major) order under the crop, so
    # the packed [NE, n] layout of every host-side array is unchanged; frames are still rebuilt on the full mask.  A
    # mask whose interior is a solid rectangle inside a padded frame (the reference's built-in geometry) thereby
    # becomes a full rectangle on the device and takes the tiled ADI path.
    dev_mask, dev_edges = _crop_to_bounding_box(mask, edges)
    if enable_diffusion:
        geom = compile_geometry(dev_mask, dev_edges, edge_conditions, dx)
    else:  # no operator needed: boundary conditions are not consulted (solver.py:1081-1083)
        from .engine import CompiledGeometry, link_flags
        z = np.zeros(dev_mask.shape)
        geom = CompiledGeometry(dev_mask, float(dx), link_flags(dev_mask), z, z, z, z)
    full_steps, rem, total_steps = _step_plan(total_time, dt)
    eng = Engine(geom, device=device)
    eng.pin_stream()          # one stream for the whole run: skip the per-launch lookup
    stored = lambda step: step % store_every == 0 or step == total_steps  # noqa: E731

    if not energy_gap > 0.0:
        return _run_scalar(eng, mask, initial_field, diffusion_coefficient, dt, rem, full_steps, total_steps, dx,
                           stored, enable_diffusion, bath_temperature, phonon_history_out, progress_callback,
                           diffusion_scheme, cn_rtol)

    # ------------------------------------------------------------------ energy-resolved mode
    gap = energy_gap
    NE = num_energy_bins
    E_bins, dE = build_energy_grid(gap, energy_min_factor, energy_max_factor, NE)
    custom_state = None
    if initial_condition_spec is not None:
        from .initial_conditions import build_initial_qp_energy_state
        custom_state = build_initial_qp_energy_state(mask=mask, E_bins=E_bins, spec=initial_condition_spec)
    if precomputed is None and gap_expression.strip():          # auto-precompute (solver.py:1106-1124)
        from .precompute import precompute_arrays
        params = SimulationParameters(
            diffusion_coefficient=diffusion_coefficient, dt=dt, total_time=total_time, mesh_size=dx,
            energy_gap=energy_gap, energy_min_factor=energy_min_factor, energy_max_factor=energy_max_factor,
            num_energy_bins=num_energy_bins, dynes_gamma=dynes_gamma, gap_expression=gap_expression, tau_0=tau_0,
            tau_s=tau_s_eff, tau_r=tau_r_eff, T_c=T_c, bath_temperature=bath_temperature)
        precomputed = precompute_arrays(mask, edges, edge_conditions, params, include_collision_kernels=False)
    has_pre = precomputed is not None
    nonuniform = has_pre and not bool(precomputed.get("is_uniform", True))
    normalize_collision_solver_name(collision_solver)

    if has_pre:
        D_array = np.asarray(precomputed["D_array"], dtype=float)
    else:
        D_array = _tb.diffusion_coefficients(E_bins, gap, diffusion_coefficient)[:, None] * np.ones((1, n))

    diffuser = None
    if enable_diffusion:
        if nonuniform:                                           # variable D(x) per bin (solver.py:1145-1164)
            dfield = np.zeros((NE, eng.ncell))
            dfield[:, eng.mask_flat] = D_array
            diffuser = _Diffuser(eng, NE, dt, rem, diffusion_scheme, cn_rtol, dfield=dfield)
        else:                                                    # one scalar D per bin (solver.py:1166-1174)
            dcoef = [float(D_array[i, 0]) if D_array.ndim == 2 else float(D_array[i]) for i in range(NE)]
            diffuser = _Diffuser(eng, NE, dt, rem, diffusion_scheme, cn_rtol, dcoef=dcoef)

    # collision tables (solver.py:1189-1238): phonon grid, thermal phonons, per-gap-class kernels
    omega_bins, idx_diff, idx_sum, diff_sign = _build_phonon_frequency_map(E_bins)
    phonon_host = thermal_phonon_occupation(omega_bins, bath_temperature)[:, None] * np.ones((1, n), dtype=float)
    if initial_condition_spec is not None:
        from .initial_conditions import build_initial_phonon_energy_state
        phonon_host = build_initial_phonon_energy_state(mask=mask, omega_bins=omega_bins, spec=initial_condition_spec,
                                                        bath_temperature=bath_temperature)
    if nonuniform:
        gap_values = precomputed.get("gap_values")
        gap_values = np.full(n, gap, dtype=float) if gap_values is None else np.asarray(gap_values, dtype=float)
        class_gaps, cls = np.unique(gap_values, return_inverse=True)
    else:
        class_gaps, cls = np.array([gap], dtype=float), None
    rho_tab = np.stack([_dynes_density_of_states(E_bins, float(g), dynes_gamma) for g in class_gaps])
    kr_tab = (np.stack([recombination_kernel_base(E_bins, float(g), tau_r_eff, T_c) for g in class_gaps])
              if enable_recombination else None)
    ks_tab = (np.stack([scattering_kernel_base(E_bins, float(g), tau_s_eff, T_c) for g in class_gaps])
              if enable_scattering else None)
    ctab = eng.make_collision_tables(kr_tab, ks_tab, rho_tab, idx_diff, idx_sum, diff_sign, cls,
                                     gap_params=dict(E=E_bins, gaps=class_gaps, tau_r=tau_r_eff, tau_s=tau_s_eff, T_c=T_c))

    # initial quasiparticle state (solver.py:1240-1283)
    if custom_state is not None:
        state_host = np.asarray(custom_state, dtype=float)
        if state_host.shape != (NE, n):
            raise ValueError(f"Full custom quasiparticle profile must have shape ({NE}, {n}); got {state_host.shape}.")
        if not np.all(np.isfinite(state_host)):
            raise ValueError("Full custom quasiparticle profile produced non-finite values.")
        if np.any(state_host < 0):
            raise ValueError("Full custom quasiparticle profile must be non-negative.")
    else:
        spatial = initial_field[mask].astype(float)
        if energy_weights is not None:
            raw = np.asarray(energy_weights, dtype=float)
            if raw.ndim != 1:
                raise ValueError("energy_weights must be a 1D array.")
            if raw.shape[0] != NE:
                raise ValueError(f"energy_weights must have length {NE}, got {raw.shape[0]}.")
            if not np.all(np.isfinite(raw)):
                raise ValueError("energy_weights must contain only finite values.")
            if np.any(raw < 0):
                raise ValueError("energy_weights must be non-negative.")
        else:
            raw = _dynes_density_of_states(E_bins, gap, dynes_gamma)
        integral = np.sum(raw) * dE
        weights = raw / integral if integral > 0 else np.ones(NE, dtype=float) / (NE * dE)
        state_host = np.empty((NE, n), dtype=float)
        for i in range(NE):
            state_host[i] = spatial * weights[i]

    state = eng.upload_packed(state_host)
    state_alt = eng.empty(NE, eng.ncell)
    phonon = eng.upload_packed(phonon_host)
    coords = np.argwhere(mask)
    cell_to_px = np.cumsum(eng.mask_flat) - 1
    warned = False

    # The guard of step k is enqueued right after the step and examined after step k + GUARD_LAG has been enqueued (or
    # before anything is stored / returned), so the device does not idle during the host round trip and the host never
    # sleeps on an event.  Messages carry the step / time of the step that was checked, exactly as the reference's.
    pending_guard: list = []

    def guard_launch(step_idx: int, time_ns: float) -> None:
        pending_guard.append((eng.pauli_stats_launch(state, ctab, pauli_density_floor), step_idx, time_ns))

    def guard_flush(keep: int = 0) -> None:
        while len(pending_guard) > keep:
            ticket, step_idx, time_ns = pending_guard.pop(0)
            guard(step_idx, time_ns, eng.pauli_stats_result(ticket))

    def guard(step_idx: int, time_ns: float, stats=None) -> None:   # Pauli guard (solver.py:1296-1344)
        nonlocal warned
        if stats is None:
            stats = eng.pauli_stats(state, ctab, pauli_density_floor)
        max_occ, (ie_max, cell_max), forb = stats
        if forb is not None:
            r, c = coords[cell_to_px[forb[1]]]
            msg = (f"Detected non-zero quasiparticle density in forbidden state (rho≈0): step={step_idx}, "
                   f"t={time_ns:.6g} ns, E={E_bins[forb[0]]:.6g} μeV, pixel=({int(r)},{int(c)}).")
            if enforce_pauli:
                raise ValueError(msg)
            if not warned:
                warnings.warn(msg, stacklevel=3)
                warned = True
        r, c = coords[cell_to_px[cell_max]]
        if pauli_error_threshold is not None and max_occ > pauli_error_threshold:
            msg = (f"Pauli occupation exceeded limit: f={max_occ:.6g} > {pauli_error_threshold:.6g} "
                   f"at step={step_idx}, t={time_ns:.6g} ns, E={E_bins[ie_max]:.6g} μeV, pixel=({int(r)},{int(c)}).")
            if enforce_pauli:
                raise ValueError(msg)
            if not warned:
                warnings.warn(msg, stacklevel=3)
                warned = True
        if pauli_warn_threshold is not None and max_occ > pauli_warn_threshold and not warned:
            warnings.warn("High occupation detected (Pauli blocking regime): "
                          f"max f={max_occ:.6g} at step={step_idx}, t={time_ns:.6g} ns, "
                          f"E={E_bins[ie_max]:.6g} μeV, pixel=({int(r)},{int(c)}).", stacklevel=3)
            warned = True

    guard(0, 0.0)

    want_ph = phonon_history_out is not None
    ph_frames: list[np.ndarray] = []
    ph_eframes: list[list[np.ndarray]] = []
    ph_widths = integration_widths_from_centers(omega_bins, fallback_width=dE) if want_ph else None

    lazy = _LazyOutputs()

    def snapshot_phonons() -> None:                              # solver.py:1354-1360
        k = len(ph_frames)
        ph_eframes.append(None)
        ph_frames.append(None)
        lazy.add(_device_frames_async(eng, phonon, mask),        # NaN-padded on the device
                 lambda arr, k=k: ph_eframes.__setitem__(k, list(arr)))
        lazy.add(_device_frames_async(eng, eng.weighted_sum(phonon, ph_widths), mask),
                 lambda arr, k=k: ph_frames.__setitem__(k, arr[0]))

    times: list[float] = [0.0]
    frames: list[np.ndarray] = []
    energy_frames: list[list[np.ndarray]] = []
    mass: list[float] = []

    def store():                                                 # solver.py:1367-1374, 1480-1489
        # frames are formed on the device (energy integral, NaN padding), cross PCIe once on a side stream while the
        # next steps run, and are handed out as they are; only a progress callback forces the integrated frame now
        k = len(frames)
        frames.append(None)
        energy_frames.append(None)
        mass.append(None)
        t_int = _device_frames_async(eng, eng.energy_integral(state, dE), mask)

        def put_integrated(arr, k=k):
            frames[k] = arr[0]
            mass[k] = float(np.sum(arr[0][mask]) * dx * dx)     # same summation order as the reference's packed sum

        lazy.add(_device_frames_async(eng, state, mask), lambda arr, k=k: energy_frames.__setitem__(k, list(arr)))
        if want_ph:
            snapshot_phonons()
        if progress_callback is not None:
            put_integrated(t_int.result())
            return frames[k]
        lazy.add(t_int, put_integrated)
        return None

    _notify(progress_callback, 0.0, store())

    collisions = bool(enable_recombination or enable_scattering)
    gen_mode = "none" if external_generation is None else external_generation.mode.strip().lower()
    # the reference gates on the raw mode string (solver.py:1459)
    gen_active = external_generation is not None and external_generation.mode != "none"

    def collide(dt_col: float, guard_step=None):
        """One collision update; with ``guard_step = (step, time)`` the Pauli guard of that step is reduced by the same
        library call (the collision is then the last operation of the step).  Returns True when the guard was enqueued."""
        nonlocal state, state_alt
        if dt_col <= 0.0 or not collisions:
            return False
        if guard_step is None:
            eng.collide(ctab, state, state_alt, phonon, dE, dt_col, enable_recombination, enable_scattering,
                        not freeze_phonon_dynamics)
        else:
            ticket = eng.collide_guarded(ctab, state, state_alt, phonon, dE, dt_col, enable_recombination,
                                         enable_scattering, not freeze_phonon_dynamics, pauli_density_floor)
            pending_guard.append((ticket, guard_step[0], guard_step[1]))
        state, state_alt = state_alt, state
        return guard_step is not None

    # Pure diffusion (no collisions, no generation) with a guard that cannot fire (no thresholds, no forbidden bins):
    # nothing but diffusion steps lies between two store points, so the stretch is one call as in scalar mode.
    batch_diffusion = (enable_diffusion and not collisions and not gen_active and diffusion_scheme == "adi"
                       and pauli_error_threshold is None and pauli_warn_threshold is None
                       and float(np.min(rho_tab)) > 1e-30)
    def generation_amount(t_start: float, dt_of_step: float):
        """dt g_ext of a step that starts at ``t_start`` when it is one number for all bins and pixels (constant / pulse
        modes), None when it is not (custom mode); 0.0 without generation."""
        if not gen_active:
            return 0.0
        if gen_mode == "constant":
            return dt_of_step * float(external_generation.rate)
        if gen_mode == "pulse":
            on = external_generation.pulse_start <= t_start < (external_generation.pulse_start
                                                               + external_generation.pulse_duration)
            return dt_of_step * float(external_generation.pulse_rate) if on else 0.0
        return None if gen_mode == "custom" else 0.0

    current_time = 0.0
    done = 0
    custom_generation = None                                     # compiled at its first use, then reused every step
    # Strang steps that follow one another without a store point in between: the closing half-step of step k and the
    # opening half-step of step k + 1 run as ONE pass over the state (Engine.collide_pair_guarded) - `opened` says that
    # the generation term and the first half-step of the step now starting were already applied by that pass.
    pair_ok = bool(collisions and enable_diffusion and ctab.get("pair"))
    opened = False
    for step in range(1, total_steps + 1):                       # solver.py:1454-1494
        final = step > full_steps
        dt_step = rem if final else dt
        if batch_diffusion:
            current_time += dt_step
            if stored(step):
                diffuser.advance(state, done + 1, step, full_steps)
                done = step
                times.append(float(current_time))
                _notify(progress_callback, current_time, store())
            continue
        if gen_active and not opened:
            amount = generation_amount(current_time, dt_step)
            if amount is None:                                   # custom mode: evaluated on the host (solver.py:918-962)
                if custom_generation is None:
                    custom_generation = _CustomGeneration(external_generation, mask)
                g_ext = evaluate_external_generation(external_generation, E_bins, n, current_time, mask,
                                                     _compiled=custom_generation)
                if g_ext is not None:
                    eng.add_scaled(state, eng.upload_packed(g_ext), dt_step)
            elif gen_mode == "constant" or amount != 0.0:
                eng.add_constant(state, amount)
        guarded = False
        if collisions and enable_diffusion:                      # Strang: C(dt/2) D(dt) C(dt/2)
            if not opened:
                collide(0.5 * dt_step)
            opened = False
            diffuser.step(state, final)
            nxt_amount = None
            if pair_ok and step < total_steps and not stored(step) and dt_step > 0.0:
                dt_next = rem if step + 1 > full_steps else dt
                nxt_amount = generation_amount(current_time + dt_step, dt_next)
            if nxt_amount is not None:
                ticket = eng.collide_pair_guarded(ctab, state, state_alt, phonon, dE, 0.5 * dt_step, 0.5 * dt_next,
                                                  nxt_amount, enable_recombination, enable_scattering,
                                                  not freeze_phonon_dynamics, pauli_density_floor)
                pending_guard.append((ticket, step, current_time + dt_step))
                state, state_alt = state_alt, state
                guarded = opened = True
            else:
                guarded = collide(0.5 * dt_step, guard_step=(step, current_time + dt_step))
        else:
            diffuse_after = enable_diffusion and dt_step > 0.0
            guarded = collide(dt_step, guard_step=None if diffuse_after else (step, current_time + dt_step))
            if diffuse_after:
                diffuser.step(state, final)
        if not guarded:
            guard_launch(step, current_time + dt_step)
        guard_flush(keep=0 if stored(step) else eng.GUARD_LAG)
        current_time += dt_step
        if stored(step):
            times.append(float(current_time))
            _notify(progress_callback, current_time, store())
    guard_flush()
    lazy.flush()

    limits = _color_limits(frames)
    if phonon_history_out is not None:
        phonon_history_out.clear()
        phonon_history_out.update({
            "phonon_frames": ph_frames,
            "phonon_energy_frames": ph_eframes,
            "phonon_energy_bins": np.asarray(omega_bins, dtype=float).copy(),
            "phonon_metadata": {"mode": "dynamic_local_coupled", "field_units": "integrated_occupation",
                                "energy_frame_units": "occupation"},
        })
    return times, frames, mass, limits, energy_frames, E_bins


def _run_scalar(eng: Engine, mask, initial_field, D, dt, rem, full_steps, total_steps, dx, stored, enable_diffusion,
                bath_temperature, phonon_history_out, progress_callback, scheme, rtol):
    """Legacy scalar mode, energy_gap == 0 (solver.py:1517-1587)."""
    u_host = initial_field[mask].astype(float)
    u = eng.upload_packed(u_host[None, :])
    diffuser = _Diffuser(eng, 1, dt, rem, scheme, rtol, dcoef=[float(D)]) if enable_diffusion else None
    times = [0.0]
    frames = [reconstruct_field(mask, u_host)]
    mass = [float(np.sum(u_host) * dx * dx)]
    _notify(progress_callback, 0.0, frames[0])
    lazy = _LazyOutputs()
    t = 0.0
    done = 0
    for step in range(1, total_steps + 1):
        t += rem if step > full_steps else dt          # same accumulation order as the reference
        if stored(step):
            if diffuser is not None:                   # nothing happens between two store points but diffusion steps
                diffuser.advance(u, done + 1, step, full_steps)
            done = step
            times.append(float(t))
            k = len(frames)
            frames.append(None)
            mass.append(None)

            def put(arr, k=k):
                frames[k] = arr[0]
                mass[k] = float(np.sum(arr[0][mask]) * dx * dx)

            ticket = _device_frames_async(eng, u, mask)
            if progress_callback is not None:
                put(ticket.result())
                _notify(progress_callback, t, frames[k])
            else:
                lazy.add(ticket, put)
    lazy.flush()
    limits = _color_limits(frames)
    if phonon_history_out is not None:
        f, ef, bins, meta = build_fixed_phonon_history(mask=mask, times=times, bath_temperature=bath_temperature,
                                                       phonon_energy_bins=None)
        phonon_history_out.update({"phonon_frames": f, "phonon_energy_frames": ef, "phonon_energy_bins": bins,
                                   "phonon_metadata": meta})
    return times, frames, mass, limits, None, None


# ------------------------------------------------------------------------------------------------------------ #
# step API (solver.py:794-875): in-place collision steps on host arrays in the reference's packed layout
# ------------------------------------------------------------------------------------------------------------ #
def _collision_step_host(state, phonon_state, kr, ks, rho, idx_diff, idx_sum, sign, dE, dt, en_r, en_s, upd,
                         cls=None, device=None) -> None:
    from .engine import CompiledGeometry, link_flags
    n = state.shape[1]
    if phonon_state.shape[1] != n:
        raise ValueError("phonon_state shape does not match quasiparticle state.")
    mask = np.ones((1, n), dtype=bool)
    z = np.zeros(mask.shape)
    eng = Engine(CompiledGeometry(mask, 1.0, link_flags(mask), z, z, z, z), device=device)
    tab = eng.make_collision_tables(kr, ks, rho, idx_diff, idx_sum, sign, cls)
    if tab["nw"] > phonon_state.shape[0]:
        raise ValueError("phonon_state has fewer bins than the index maps address.")
    s_in = eng.upload_packed(state)
    s_out = eng.empty(*s_in.shape)
    ph = eng.upload_packed(phonon_state)
    eng.collide(tab, s_in, s_out, ph, dE, dt, en_r, en_s, upd)
    state[:, :] = eng.download_packed(s_out)
    if upd:
        phonon_state[:, :] = eng.download_packed(ph)


def _euler_call(state2d: np.ndarray, K_r, G_therm, K_s, rho, dE, dt, rhs_only, device=None) -> np.ndarray:
    import ctypes as C
    from . import _hip
    from .engine import require_gpu
    torch = require_gpu()
    lib = _hip.load()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    up = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)  # noqa: E731
    s_in = up(state2d)
    out = torch.empty_like(s_in)
    kr, g, ks, rh = up(K_r), up(G_therm), up(K_s), up(rho)
    ptr = lambda t: 0 if t is None else int(t.data_ptr())  # noqa: E731
    _hip.check(lib.qp_euler_collision(state2d.shape[0], state2d.shape[1], ptr(s_in), ptr(out), ptr(kr), ptr(g), ptr(ks),
                                      ptr(rh), float(dE), float(dt), int(rhs_only),
                                      int(torch.cuda.current_stream(dev).cuda_stream)), "qp_euler_collision")
    return out.cpu().numpy()


def apply_scattering_step(state: np.ndarray, K_s: np.ndarray, rho_bins: np.ndarray, dE: float, dt: float) -> None:
    """One forward-Euler step of fixed-bath scattering, in place on state[NE, n] (solver.py:551-580)."""
    state[:, :] = _euler_call(state, None, None, K_s, rho_bins, dE, dt, False)


def apply_recombination_step(state: np.ndarray, K_r: np.ndarray, G_therm: np.ndarray, dE: float, dt: float) -> None:
    """One forward-Euler step of recombination + thermal generation, in place (solver.py:583-605)."""
    state[:, :] = _euler_call(state, K_r, G_therm, None, None, dE, dt, False)


def _collision_rhs(n: np.ndarray, K_r, K_s, rho_bins, G_therm, dE: float) -> np.ndarray:
    """dn/dt of one pixel from recombination (if K_r and G_therm) and scattering (if K_s and rho) (solver.py:608-637)."""
    use_r = K_r is not None and G_therm is not None
    use_s = K_s is not None and rho_bins is not None
    if not (use_r or use_s):
        return np.zeros_like(n)
    return _euler_call(np.asarray(n, dtype=float)[:, None], K_r if use_r else None, G_therm if use_r else None,
                       K_s if use_s else None, rho_bins if use_s else None, dE, 0.0, True)[:, 0]


def _mask_to_index(mask: np.ndarray):
    """index_map[ny, nx] (-1 outside) and interior (row, col) list in argwhere order (solver.py:53-58)."""
    mask = np.asarray(mask, dtype=bool)
    index_map = -np.ones(mask.shape, dtype=int)
    index_map[mask] = np.arange(int(mask.sum()))
    return index_map, [tuple(map(int, rc)) for rc in np.argwhere(mask)]


def _assemble_csr(geom, D):
    """CSR matrix and source vector of the 5-point operator described by a CompiledGeometry (D scalar or packed [n])."""
    from scipy import sparse
    mask = geom.mask
    n = int(mask.sum())
    index = -np.ones(mask.shape, dtype=np.int64)
    index[mask] = np.arange(n)
    Dg = np.zeros(mask.shape)
    Dg[mask] = D
    inv_dx2 = 1.0 / (geom.dx * geom.dx)
    rows, cols, vals = [], [], []
    diag = -(geom.ex + geom.ey) * Dg * inv_dx2
    for bit, (dr, dc) in ((1, (0, -1)), (2, (0, 1)), (4, (-1, 0)), (8, (1, 0))):
        r, c = np.nonzero((geom.flags & bit) > 0)
        dp, dq = Dg[r, c], Dg[r + dr, c + dc]
        w = 2.0 * dp * dq / np.maximum(dp + dq, 1e-30) * inv_dx2
        rows.append(index[r, c])
        cols.append(index[r + dr, c + dc])
        vals.append(w)
        np.subtract.at(diag, (r, c), w)
    r, c = np.nonzero(mask)
    rows.append(index[r, c])
    cols.append(index[r, c])
    vals.append(diag[r, c])
    L = sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    return L, ((geom.sx + geom.sy) * Dg * inv_dx2)[mask], index


def build_laplacian_with_boundaries(mask, edges, edge_conditions, dx):
    """(L csr [n, n] in 1/dx^2 units, source [n], index_map) -- host-side equivalent of solver.py:152-212.

    The GPU path never forms this matrix; the function exists for callers that inspect the operator.
    """
    geom = compile_geometry(np.asarray(mask, dtype=bool), edges, edge_conditions, dx)
    L, src, index = _assemble_csr(geom, 1.0)
    index_map = np.where(geom.mask, index, -1).astype(int)
    return L, src, index_map


def build_variable_diffusion_laplacian(mask, edges, edge_conditions, dx, D_spatial):
    """(L_D csr, source) with harmonic-mean face diffusivities (host-side equivalent of solver.py:235-321)."""
    geom = compile_geometry(np.asarray(mask, dtype=bool), edges, edge_conditions, dx)
    L, src, _ = _assemble_csr(geom, np.asarray(D_spatial, dtype=float))
    return L, src


def apply_collision_step_fischer_catelani_uniform(state, phonon_state, K_r0, K_s0, rho_bins, omega_idx_diff,
                                                  omega_idx_sum, diff_sign, dE, dt, *, enable_recombination,
                                                  enable_scattering, update_phonons=True, device=None) -> None:
    """One coupled collision step with one kernel set for all pixels, in place (solver.py:794-831)."""
    _collision_step_host(state, phonon_state, None if K_r0 is None else np.asarray(K_r0)[None],
                         None if K_s0 is None else np.asarray(K_s0)[None], np.asarray(rho_bins)[None], omega_idx_diff,
                         omega_idx_sum, diff_sign, dE, dt, enable_recombination, enable_scattering, update_phonons,
                         device=device)


def apply_collision_step_fischer_catelani_nonuniform(state, phonon_state, K_r0_all, K_s0_all, rho_all, omega_idx_diff,
                                                     omega_idx_sum, diff_sign, dE, dt, *, enable_recombination,
                                                     enable_scattering, update_phonons=True, device=None) -> None:
    """Per-pixel kernels [n, NE, NE] / [n, NE]; de-duplicated into classes before upload (solver.py:834-875)."""
    rho_all = np.asarray(rho_all, dtype=float)
    if rho_all.shape[0] != state.shape[1]:
        raise ValueError("rho_all shape does not match quasiparticle state.")
    n = rho_all.shape[0]
    key = [rho_all.reshape(n, -1)]
    if K_r0_all is not None:
        key.append(np.asarray(K_r0_all, dtype=float).reshape(n, -1))
    if K_s0_all is not None:
        key.append(np.asarray(K_s0_all, dtype=float).reshape(n, -1))
    _, first, cls = np.unique(np.concatenate(key, axis=1), axis=0, return_index=True, return_inverse=True)
    cls = np.asarray(cls).reshape(-1)
    _collision_step_host(state, phonon_state, None if K_r0_all is None else np.asarray(K_r0_all)[first],
                         None if K_s0_all is None else np.asarray(K_s0_all)[first], rho_all[first], omega_idx_diff,
                         omega_idx_sum, diff_sign, dE, dt, enable_recombination, enable_scattering, update_phonons,
                         cls=cls, device=device)
