"""Host-side builders that turn an ``InitialConditionSpec`` / gap expression into arrays.

These produce the inputs of the device time loop (initial field, energy weights, full
quasiparticle / phonon states, gap map).  Semantics follow ``qpsim/initial_conditions.py``
(lines cited per function), including its own Boltzmann constant (``:20``, different in the
7th digit from the solver's) and its ``expm1``-based Bose-Einstein form -- both visible at
1e-10 and therefore kept.
"""
from __future__ import annotations

from typing import Any

import numpy as np

from .models import InitialConditionSpec
from .safe_eval import compile_safe_expression
from .tables import thermal_qp_weights

_KB_IC_UEV_PER_K = 86.173303  # initial_conditions.py:20

_DEFAULTS = InitialConditionSpec()
_DEFAULT_GAUSSIAN = {"amplitude": 1.0, "x0": 0.5, "y0": 0.5, "sigma": 0.12}


def _truthy(value: Any) -> bool:
    if isinstance(value, str):
        return value.strip().lower() in {"1", "true", "yes", "on"}
    return bool(value)


def default_initial_condition() -> InitialConditionSpec:
    """Centred gaussian x DOS spectrum, uniform thermal phonons (initial_conditions.py:31-55)."""
    return InitialConditionSpec(
        spatial_kind="gaussian", spatial_params=dict(_DEFAULT_GAUSSIAN), energy_kind="dos",
        phonon_spatial_kind="uniform", phonon_spatial_params={"value": 1.0}, phonon_energy_kind="bose_einstein",
    )


def _split(kind, params, body, cparams, fallback_kind, fallback_params, fallback_body):
    k = str(kind or "").strip().lower()
    if not k:
        return fallback_kind, dict(fallback_params), fallback_body, {}
    return k, dict(params or {}), str(body or fallback_body), dict(cparams or {})


def resolve_spatial_spec(spec: InitialConditionSpec):
    """(kind, params, custom_body, custom_params); empty kind -> default gaussian (:58-71)."""
    return _split(spec.spatial_kind, spec.spatial_params, spec.spatial_custom_body, spec.spatial_custom_params,
                  "gaussian", _DEFAULT_GAUSSIAN, _DEFAULTS.spatial_custom_body)


def resolve_energy_spec(spec: InitialConditionSpec):
    """Empty kind -> "dos" (:74-82)."""
    return _split(spec.energy_kind, spec.energy_params, spec.energy_custom_body, spec.energy_custom_params,
                  "dos", {}, _DEFAULTS.energy_custom_body)


def resolve_phonon_spatial_spec(spec: InitialConditionSpec):
    """Empty kind -> uniform 1.0 (:85-92)."""
    return _split(spec.phonon_spatial_kind, spec.phonon_spatial_params, spec.phonon_spatial_custom_body,
                  spec.phonon_spatial_custom_params, "uniform", {"value": 1.0}, _DEFAULTS.phonon_spatial_custom_body)


def resolve_phonon_energy_spec(spec: InitialConditionSpec):
    """Empty kind -> bose_einstein at the bath temperature (:95-102)."""
    return _split(spec.phonon_energy_kind, spec.phonon_energy_params, spec.phonon_energy_custom_body,
                  spec.phonon_energy_custom_params, "bose_einstein", {}, _DEFAULTS.phonon_energy_custom_body)


def resolve_qp_full_custom_spec(spec: InitialConditionSpec):
    return (_truthy(spec.qp_full_custom_enabled), str(spec.qp_full_custom_body or _DEFAULTS.qp_full_custom_body),
            dict(spec.qp_full_custom_params or {}))


def resolve_phonon_full_custom_spec(spec: InitialConditionSpec):
    return (_truthy(spec.phonon_full_custom_enabled),
            str(spec.phonon_full_custom_body or _DEFAULTS.phonon_full_custom_body),
            dict(spec.phonon_full_custom_params or {}))


def canonicalize_initial_condition(spec: InitialConditionSpec) -> InitialConditionSpec:
    """Spec with every empty kind/body replaced by its default (initial_conditions.py:141-177)."""
    sk, sp, sb, scp = resolve_spatial_spec(spec)
    ek, ep, eb, ecp = resolve_energy_spec(spec)
    pk, pp, pb, pcp = resolve_phonon_spatial_spec(spec)
    qk, qp, qb, qcp = resolve_phonon_energy_spec(spec)
    qf_on, qf_body, qf_params = resolve_qp_full_custom_spec(spec)
    pf_on, pf_body, pf_params = resolve_phonon_full_custom_spec(spec)
    return InitialConditionSpec(
        spatial_kind=sk, spatial_params=sp, spatial_custom_body=sb, spatial_custom_params=scp,
        energy_kind=ek, energy_params=ep, energy_custom_body=eb, energy_custom_params=ecp,
        qp_full_custom_enabled=bool(qf_on), qp_full_custom_body=qf_body, qp_full_custom_params=qf_params,
        phonon_spatial_kind=pk, phonon_spatial_params=pp, phonon_spatial_custom_body=pb, phonon_spatial_custom_params=pcp,
        phonon_energy_kind=qk, phonon_energy_params=qp, phonon_energy_custom_body=qb, phonon_energy_custom_params=qcp,
        phonon_full_custom_enabled=bool(pf_on), phonon_full_custom_body=pf_body, phonon_full_custom_params=pf_params)


def _pixel_centres(mask: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Normalised pixel-centre coordinates x, y in (0, 1) on the full grid."""
    ny, nx = mask.shape
    rows, cols = np.indices(mask.shape)
    return (cols + 0.5) / max(1, nx), (rows + 0.5) / max(1, ny)


def _evaluate_on_interior(fn, xs: np.ndarray, ys: np.ndarray, mask: np.ndarray, params: dict) -> np.ndarray:
    """f(x, y, params) over interior pixels: vectorised call first, per-pixel scalar calls otherwise (:189-213)."""
    mx, my = xs[mask], ys[mask]
    if mx.size == 0:
        return np.empty((0,), dtype=float)
    try:
        arr = np.asarray(fn(x=mx, y=my, params=params), dtype=float)
        if arr.ndim == 0:
            return np.full(mx.shape[0], float(arr), dtype=float)
        if arr.size == mx.size:
            return arr.reshape(mx.size)
        if arr.shape == mask.shape:
            return np.asarray(arr[mask], dtype=float)
    except Exception:
        pass
    return np.array([float(fn(x=float(a), y=float(b), params=params)) for a, b in zip(mx, my)], dtype=float)


def _spatial_profile(mask: np.ndarray, kind: str, params: dict, body: str, cparams: dict) -> np.ndarray:
    """Spatial profile on the full grid, zero outside the mask (initial_conditions.py:216-280)."""
    mask = np.asarray(mask, dtype=bool)
    if mask.ndim != 2:
        raise ValueError("Geometry mask must be 2D.")
    ny, nx = mask.shape
    xs, ys = _pixel_centres(mask)
    mode = str(kind or "").strip().lower()
    out = np.zeros(mask.shape, dtype=float)
    if mode == "gaussian":
        sigma = max(1e-6, float(params.get("sigma", 0.12)))
        r2 = (xs - float(params.get("x0", 0.5))) ** 2 + (ys - float(params.get("y0", 0.5))) ** 2
        out = float(params.get("amplitude", 1.0)) * np.exp(-r2 / (2.0 * sigma * sigma))
    elif mode == "uniform":
        out.fill(float(params.get("value", 1.0)))
    elif mode == "point":
        col = int(np.clip(round(float(params.get("x0", 0.5)) * (nx - 1)), 0, nx - 1))
        row = int(np.clip(round(float(params.get("y0", 0.5)) * (ny - 1)), 0, ny - 1))
        if not mask[row, col]:
            inside = np.argwhere(mask)
            if inside.size:
                row, col = (int(v) for v in inside[int(np.argmin((inside[:, 0] - row) ** 2 + (inside[:, 1] - col) ** 2))])
        if mask[row, col]:
            out[row, col] = float(params.get("value", 1.0))
    elif mode == "custom":
        fn = compile_safe_expression(body, variable_names=("x", "y", "params"))
        out[mask] = _evaluate_on_interior(fn, xs, ys, mask, cparams)
    else:
        raise ValueError(f"Unsupported spatial initial-condition kind: '{kind}'.")
    out[~mask] = 0.0
    if not np.all(np.isfinite(out[mask])):
        raise ValueError("Spatial initial-condition profile produced non-finite values.")
    return out


def build_initial_field(mask: np.ndarray, spec: InitialConditionSpec) -> np.ndarray:
    return _spatial_profile(mask, *resolve_spatial_spec(spec))


def build_initial_phonon_spatial_field(mask: np.ndarray, spec: InitialConditionSpec) -> np.ndarray:
    return _spatial_profile(mask, *resolve_phonon_spatial_spec(spec))


def evaluate_gap_expression(expression: str, mask: np.ndarray, energy_gap_default: float) -> np.ndarray:
    """Gap value per interior pixel; empty expression -> uniform default (initial_conditions.py:283-326)."""
    mask = np.asarray(mask, dtype=bool)
    n = int(mask.sum())
    if expression.strip():
        fn = compile_safe_expression(expression, variable_names=("x", "y", "params"))
        xs, ys = _pixel_centres(mask)
        vals = _evaluate_on_interior(fn, xs, ys, mask, {})
    else:
        vals = np.full(n, energy_gap_default, dtype=float)
    vals = np.asarray(vals, dtype=float).reshape(-1)
    if vals.size != n:
        raise ValueError(f"Gap expression returned {vals.size} values; expected {n} interior pixels.")
    if not np.all(np.isfinite(vals)):
        raise ValueError("Gap expression produced non-finite values.")
    if np.any(vals <= 0.0):
        raise ValueError("Gap expression must produce strictly positive values.")
    return vals


def _profile_over_bins(fn, bins: np.ndarray, label: str, **extra) -> np.ndarray:
    try:
        arr = np.asarray(fn(E=bins, **extra), dtype=float)
    except Exception:
        arr = np.asarray([float(fn(E=float(e), **extra)) for e in bins], dtype=float)
    arr = np.asarray(arr, dtype=float).reshape(-1)
    if arr.size == 1:
        arr = np.full_like(bins, float(arr[0]), dtype=float)
    if arr.size != bins.size:
        raise ValueError(f"Custom {label} must return {bins.size} values or a scalar; got {arr.size}.")
    return arr


def build_initial_energy_weights(E_bins: np.ndarray, gap: float, dynes_gamma: float, spec: InitialConditionSpec,
                                 bath_temperature: float) -> np.ndarray | None:
    """Energy weights, or None for the solver's DOS default / full custom state (initial_conditions.py:353-412)."""
    if resolve_qp_full_custom_spec(spec)[0]:
        return None
    kind, params, body, cparams = resolve_energy_spec(spec)
    E = np.asarray(E_bins, dtype=float)
    if kind in {"", "dos", "default", "bcs_dos"}:
        return None
    if kind == "fermi_dirac":
        return thermal_qp_weights(E, gap, float(params.get("temperature", bath_temperature)), dynes_gamma)
    if kind == "uniform":
        value = float(params.get("value", 1.0))
        if value < 0:
            raise ValueError("Uniform energy profile value must be non-negative.")
        return np.full_like(E, value, dtype=float)
    if kind == "custom":
        fn = compile_safe_expression(body.strip() or _DEFAULTS.energy_custom_body, variable_names=("E", "gap", "params"))
        arr = _profile_over_bins(fn, E, "energy profile", gap=float(gap), params=dict(cparams))
        if not np.all(np.isfinite(arr)):
            raise ValueError("Custom energy profile produced non-finite values.")
        if np.any(arr < 0):
            raise ValueError("Custom energy profile must be non-negative.")
        return arr
    raise ValueError(
        f"Unsupported energy initial-condition kind '{kind}'. Supported: dos, fermi_dirac, uniform, custom.")


def _to_energy_by_pixel(arr: np.ndarray, bins: np.ndarray, mask: np.ndarray, label: str) -> np.ndarray:
    """Coerce any of the accepted result shapes to [N_E, n_interior] (initial_conditions.py:415-453)."""
    nE = int(bins.size)
    ny, nx = mask.shape
    n = int(mask.sum())
    a = np.asarray(arr, dtype=float)
    if a.ndim == 0:
        return np.full((nE, n), float(a), dtype=float)
    if a.shape == (nE, n):
        return a
    if a.shape == (n, nE):
        return a.T
    if a.shape == (nE, ny, nx):
        return a[:, mask]
    if a.shape == (ny, nx, nE):
        return np.moveaxis(a, 2, 0)[:, mask]
    if a.shape == (ny, nx):
        return np.repeat(a[mask][None, :], nE, axis=0)
    if a.shape == (nE,):
        return np.repeat(a.reshape(nE, 1), n, axis=1)
    if a.shape == (n,):
        return np.repeat(a.reshape(1, n), nE, axis=0)
    if a.size == nE * n:
        return a.reshape(nE, n)
    raise ValueError(
        f"{label} expression returned shape {a.shape}; expected scalar, (N_E,), (N_x*N_y,), "
        f"(N_E, N_x*N_y), or full-grid shapes tied to mask {mask.shape}.")


def _full_custom_state(mask: np.ndarray, bins: np.ndarray, body: str, params: dict, label: str) -> np.ndarray:
    """Non-separable F(x, y, E) evaluated on interior pixels x bins (initial_conditions.py:456-507)."""
    mask = np.asarray(mask, dtype=bool)
    if mask.ndim != 2:
        raise ValueError("Geometry mask must be 2D.")
    bins = np.asarray(bins, dtype=float)
    if bins.size <= 0:
        raise ValueError("Energy bins must be non-empty for full custom profile evaluation.")
    fn = compile_safe_expression(body.strip(), variable_names=("x", "y", "E", "params"))
    ny, nx = mask.shape
    coords = np.argwhere(mask)
    xv = (coords[:, 1].astype(float) + 0.5) / max(1, nx)
    yv = (coords[:, 0].astype(float) + 0.5) / max(1, ny)
    try:
        raw = np.asarray(fn(x=xv[None, :], y=yv[None, :], E=bins[:, None], params=params), dtype=float)
    except Exception:
        raw = np.array([[float(fn(x=float(a), y=float(b), E=float(e), params=params)) for a, b in zip(xv, yv)]
                        for e in bins], dtype=float)
    state = _to_energy_by_pixel(raw, bins, mask, label)
    if not np.all(np.isfinite(state)):
        raise ValueError(f"{label} expression produced non-finite values.")
    if np.any(state < 0):
        raise ValueError(f"{label} expression must be non-negative.")
    return state


def build_initial_qp_energy_state(mask: np.ndarray, E_bins: np.ndarray, spec: InitialConditionSpec) -> np.ndarray | None:
    """Optional full quasiparticle state [N_E, n]; None unless enabled (initial_conditions.py:510-525)."""
    enabled, body, params = resolve_qp_full_custom_spec(spec)
    if not enabled:
        return None
    return _full_custom_state(mask, np.asarray(E_bins, dtype=float), body, params, "Full quasiparticle profile")


def _bose_einstein_ic(energies: np.ndarray, temperature: float) -> np.ndarray:
    """1/expm1(E / k_B T) with the IC module's constant and 700 clamp (initial_conditions.py:528-541)."""
    e = np.maximum(0.0, np.asarray(energies, dtype=float))
    if float(temperature) <= 0.0:
        return np.zeros_like(e)
    den = np.expm1(np.clip(e / (_KB_IC_UEV_PER_K * float(temperature)), 0.0, 700.0))
    return np.divide(1.0, den, out=np.zeros_like(e), where=den > 0.0)


def build_initial_phonon_energy_weights(omega_bins: np.ndarray, spec: InitialConditionSpec,
                                        bath_temperature: float) -> np.ndarray:
    """Phonon occupation per omega bin (initial_conditions.py:544-599)."""
    kind, params, body, cparams = resolve_phonon_energy_spec(spec)
    omega = np.asarray(omega_bins, dtype=float).reshape(-1)
    if omega.size == 0:
        raise ValueError("omega_bins must be non-empty.")
    if not np.all(np.isfinite(omega)):
        raise ValueError("omega_bins must contain finite values.")
    if np.any(omega < 0):
        raise ValueError("omega_bins must be non-negative.")
    if kind in {"", "bose_einstein", "be", "thermal"}:
        values = _bose_einstein_ic(omega, float(params.get("temperature", bath_temperature)))
    elif kind == "uniform":
        value = float(params.get("value", 1.0))
        if value < 0:
            raise ValueError("Uniform phonon energy profile value must be non-negative.")
        values = np.full_like(omega, value, dtype=float)
    elif kind == "custom":
        fn = compile_safe_expression(body.strip() or _DEFAULTS.phonon_energy_custom_body, variable_names=("E", "params"))
        values = _profile_over_bins(fn, omega, "phonon energy profile", params=dict(cparams))
    else:
        raise ValueError(
            f"Unsupported phonon energy initial-condition kind '{kind}'. Supported: bose_einstein, uniform, custom.")
    if not np.all(np.isfinite(values)):
        raise ValueError("Phonon energy profile produced non-finite values.")
    if np.any(values < 0):
        raise ValueError("Phonon energy profile must be non-negative.")
    return values


def build_initial_phonon_energy_state(mask: np.ndarray, omega_bins: np.ndarray, spec: InitialConditionSpec,
                                      bath_temperature: float) -> np.ndarray:
    """Phonon state [N_omega, n] = energy occupation x spatial profile, or full custom (initial_conditions.py:602-632)."""
    omega = np.asarray(omega_bins, dtype=float)
    enabled, body, params = resolve_phonon_full_custom_spec(spec)
    if enabled:
        return _full_custom_state(mask, omega, body, params, "Full phonon profile")
    mask = np.asarray(mask, dtype=bool)
    spatial = build_initial_phonon_spatial_field(mask, spec)[mask].reshape(1, -1)
    state = build_initial_phonon_energy_weights(omega, spec, bath_temperature).reshape(-1, 1) * spatial
    if not np.all(np.isfinite(state)):
        raise ValueError("Phonon initial state produced non-finite values.")
    if np.any(state < 0):
        raise ValueError("Phonon initial state must be non-negative.")
    return state
