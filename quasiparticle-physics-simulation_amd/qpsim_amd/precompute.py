"""Coefficient producer for the time loop (drop-in for ``qpsim.precompute``).

``precompute_arrays`` returns the same dict of arrays the reference writes into its
``.precompute.npz`` sidecars (``qpsim/precompute.py:173-287``), with the same fingerprint layout
(``:44-76``) so sidecars made by either implementation validate against the other.  The solver
consumes only ``is_uniform``, ``D_array`` and ``gap_values`` (reference solver.py:1128,1132,1204);
the optional fixed-bath kernel payload is produced for compatibility.
"""
from __future__ import annotations

import hashlib
from typing import Any, Callable

import numpy as np

from .initial_conditions import evaluate_gap_expression
from .models import BoundaryCondition, EdgeSegment, SimulationParameters
from .tables import (
    build_energy_grid,
    dynes_density_of_states,
    recombination_kernel,
    scattering_kernel,
    thermal_qp_weights,
)

_REQUIRED_KEYS = ("fingerprint", "E_bins", "gap_values", "is_uniform", "D_array")
_KERNEL_KEYS = ("K_r", "K_s", "rho_bins", "G_therm", "K_r_all", "K_s_all", "rho_all", "G_therm_all")
_BASE_LABELS = ["energy_gap", "energy_min_factor", "energy_max_factor", "num_energy_bins", "dynes_gamma",
                "diffusion_coefficient", "n_spatial", "mask_hash", "gap_expression"]
_KERNEL_LABELS = ["tau_s", "tau_r", "T_c", "bath_temperature"]


def _mask_hash(mask: np.ndarray) -> float:
    """sha256(shape as int64 || packbits(mask)) folded into a float-exact integer (< 2^53) (precompute.py:18-27)."""
    m = np.asarray(mask, dtype=bool)
    h = hashlib.sha256()
    h.update(np.asarray(m.shape, dtype=np.int64).tobytes())
    h.update(np.packbits(m.astype(np.uint8, copy=False)).tobytes())
    return float(int.from_bytes(h.digest()[:8], "big") % (2 ** 53))


def _gap_expression_hash(gap_expression: str) -> float:
    """First 16 hex digits of sha256(expression) mod 2^53 (precompute.py:30-33)."""
    return float(int(hashlib.sha256(gap_expression.encode()).hexdigest()[:16], 16) % (2 ** 53))


def _scalar_flag(value: Any) -> bool:
    if isinstance(value, np.ndarray):
        return bool(value.reshape(-1)[0]) if value.size else False
    return bool(value)


def _taus(params: SimulationParameters) -> tuple[float, float]:
    tau_s = float(params.tau_s if params.tau_s is not None else params.tau_0)
    tau_r = float(params.tau_r if params.tau_r is not None else params.tau_0)
    return tau_s, tau_r


def _make_fingerprint(params: SimulationParameters, mask: np.ndarray, *, include_collision_kernels: bool) -> np.ndarray:
    """9 floats (+4 with kernels) identifying what the arrays depend on (precompute.py:44-76)."""
    vals = [params.energy_gap, params.energy_min_factor, params.energy_max_factor, float(params.num_energy_bins),
            params.dynes_gamma, params.diffusion_coefficient, float(int(np.sum(mask))), _mask_hash(mask),
            float(_gap_expression_hash(params.gap_expression))]
    if include_collision_kernels:
        tau_s, tau_r = _taus(params)
        vals += [tau_s, tau_r, params.T_c, params.bath_temperature]
    return np.array(vals, dtype=float)


def validate_precomputed(precomputed: dict[str, Any], params: SimulationParameters, mask: np.ndarray) -> str | None:
    """None if ``precomputed`` matches ``params``/``mask``, else a description of the mismatch (precompute.py:79-148)."""
    for key in _REQUIRED_KEYS:
        if key not in precomputed:
            return f"Precomputed file missing required key '{key}'."
    n_spatial, n_energy = int(np.sum(mask)), int(params.num_energy_bins)

    def numeric(key: str, flat: bool = True):
        try:
            arr = np.asarray(precomputed.get(key), dtype=float)
            return arr.reshape(-1) if flat else arr
        except Exception:
            return None

    e_bins = numeric("E_bins")
    if e_bins is None:
        return "Precomputed key 'E_bins' is not a valid numeric array."
    if e_bins.size != n_energy:
        return f"E_bins length mismatch: stored {e_bins.size} vs current {n_energy}."
    gaps = numeric("gap_values")
    if gaps is None:
        return "Precomputed key 'gap_values' is not a valid numeric array."
    if gaps.size != n_spatial:
        return f"gap_values length mismatch: stored {gaps.size} vs current {n_spatial}."
    d_array = numeric("D_array", flat=False)
    if d_array is None:
        return "Precomputed key 'D_array' is not a valid numeric array."
    if d_array.shape != (n_energy, n_spatial):
        return f"D_array shape mismatch: stored {tuple(d_array.shape)} vs current {(n_energy, n_spatial)}."
    stored = numeric("fingerprint")
    if stored is None:
        return "Precomputed key 'fingerprint' is not a valid numeric array."
    with_kernels = _scalar_flag(precomputed.get("include_collision_kernels",
                                                any(k in precomputed for k in _KERNEL_KEYS)))
    current = _make_fingerprint(params, mask, include_collision_kernels=with_kernels)
    if stored.shape != current.shape:
        return f"Fingerprint size mismatch: stored {stored.shape} vs current {current.shape}."
    if np.allclose(stored, current, rtol=1e-12, atol=1e-12):
        return None
    labels = _BASE_LABELS + (_KERNEL_LABELS if with_kernels else [])
    diffs = [f"{labels[i] if i < len(labels) else f'param[{i}]'}: stored={s}, current={c}"
             for i, (s, c) in enumerate(zip(stored, current))
             if abs(s - c) > 1e-12 * max(abs(s), abs(c), 1.0)]
    return "Parameter mismatch: " + "; ".join(diffs)


def estimate_precompute_memory(n_spatial: int, n_energy: int, is_uniform: bool,
                               include_collision_kernels: bool = False) -> int:
    """Bytes of the arrays ``precompute_arrays`` would return (precompute.py:151-170)."""
    total = 8 * (n_energy * n_spatial + n_energy + n_spatial)
    if include_collision_kernels:
        per = 2 * n_energy ** 2 + 2 * n_energy
        total += 8 * (per if is_uniform else n_spatial * per)
    return total


def _fixed_bath_tables(E, dE, gap, params: SimulationParameters):
    tau_s, tau_r = _taus(params)
    K_r = recombination_kernel(E, gap, tau_r, params.T_c, params.bath_temperature)
    K_s = scattering_kernel(E, gap, tau_s, params.T_c, params.bath_temperature)
    rho = dynes_density_of_states(E, gap, params.dynes_gamma)
    n_eq = thermal_qp_weights(E, gap, params.bath_temperature, params.dynes_gamma)
    return K_r, K_s, rho, 2.0 * n_eq * dE * (K_r @ n_eq)


def precompute_arrays(mask: np.ndarray, edges: list[EdgeSegment], edge_conditions: dict[str, BoundaryCondition],
                      params: SimulationParameters, progress_callback: Callable[[str], None] | None = None, *,
                      include_collision_kernels: bool = False) -> dict[str, Any]:
    """Gap map, D(E, x) and optional fixed-bath kernels (precompute.py:173-287)."""
    if params.energy_gap <= 0:
        raise ValueError("precompute_arrays requires energy_gap > 0.")
    say = progress_callback or (lambda _msg: None)
    mask = np.asarray(mask, dtype=bool)
    n_spatial, NE = int(mask.sum()), params.num_energy_bins
    E, dE = build_energy_grid(params.energy_gap, params.energy_min_factor, params.energy_max_factor, NE)
    say("Evaluating gap expression...")
    gap_values = evaluate_gap_expression(params.gap_expression, mask, params.energy_gap)
    unique_gaps = np.unique(gap_values)
    is_uniform = len(unique_gaps) == 1
    say(f"{'Uniform' if is_uniform else f'{len(unique_gaps)} unique'} gap values")

    # D(E, x) = D0 sqrt(1 - min(gap(x)/E, 1)^2)   (precompute.py:211-215)
    ratio = np.minimum(gap_values[None, :] / E[:, None], 1.0)
    D_array = params.diffusion_coefficient * np.sqrt(np.maximum(0.0, 1.0 - ratio ** 2))

    out: dict[str, Any] = {
        "fingerprint": _make_fingerprint(params, mask, include_collision_kernels=include_collision_kernels),
        "include_collision_kernels": np.array(bool(include_collision_kernels)),
        "E_bins": E, "gap_values": gap_values, "is_uniform": np.array(is_uniform), "D_array": D_array,
    }
    if include_collision_kernels and is_uniform:
        say("Computing uniform kernels...")
        out["K_r"], out["K_s"], out["rho_bins"], out["G_therm"] = _fixed_bath_tables(E, dE, float(unique_gaps[0]), params)
    elif include_collision_kernels:
        say("Computing per-pixel kernels (caching by unique gap)...")
        per_gap = [_fixed_bath_tables(E, dE, float(g), params) for g in unique_gaps]
        cls = np.searchsorted(unique_gaps, gap_values)
        for slot, name in enumerate(("K_r_all", "K_s_all", "rho_all", "G_therm_all")):
            out[name] = np.stack([t[slot] for t in per_gap])[cls]
    say("Precomputation complete." if include_collision_kernels
        else "Precomputation complete (diffusion/gap arrays only).")
    return out
