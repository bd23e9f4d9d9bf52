"""Fast physics acceptance checks run through the GPU solver (counterpart of ``qpsim.validation``).

Five checks on 1 x nx strips with the reference's parameters and tolerances
(``qpsim/validation.py:76-365``): detailed balance of the fixed-bath scattering kernel, stationarity of the thermal
spectrum under the full loop with frozen thermal phonons, mass conservation of pure diffusion, of pure scattering, and
monotone decay under pure recombination.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import numpy as np

from .geometry import extract_edge_segments
from .models import BoundaryCondition, InitialConditionSpec, SimulationParameters
from .solver import run_2d_crank_nicolson
from .tables import KB_UEV_PER_K, build_energy_grid, scattering_kernel, thermal_qp_weights

_SECTIONS = ("detailed_balance", "thermal_stability", "pure_diffusion", "pure_scattering", "pure_recombination")


def _strip(nx: int):
    mask = np.ones((1, nx), dtype=bool)
    edges = extract_edge_segments(mask)
    return mask, edges, {e.edge_id: BoundaryCondition(kind="reflective") for e in edges}


def _thermal_phonons(T: float) -> InitialConditionSpec:
    return InitialConditionSpec(spatial_kind="uniform", spatial_params={"value": 1.0}, energy_kind="dos",
                                phonon_spatial_kind="uniform", phonon_spatial_params={"value": 1.0},
                                phonon_energy_kind="bose_einstein", phonon_energy_params={"temperature": float(T)})


@dataclass
class ValidationReport:
    detailed_balance: dict[str, Any]
    thermal_stability: dict[str, Any]
    pure_diffusion: dict[str, Any]
    pure_scattering: dict[str, Any]
    pure_recombination: dict[str, Any]

    @property
    def overall_passed(self) -> bool:
        return all(bool(getattr(self, s).get("passed", False)) for s in _SECTIONS)

    def as_dict(self) -> dict[str, Any]:
        out = {s: getattr(self, s) for s in _SECTIONS}
        out["overall_passed"] = self.overall_passed
        return out


def validate_detailed_balance(*, gap, energy_min_factor, energy_max_factor, num_energy_bins, tau_s, T_c,
                              bath_temperature, tolerance: float = 1e-9) -> dict[str, Any]:
    """K^s_ij = K^s_ji exp((Ei - Ej)/kT) for the fixed-bath kernel (validation.py:76-99)."""
    if bath_temperature <= 0:
        return {"passed": True, "max_relative_error": 0.0, "message": "Skipped (T_bath <= 0)."}
    E, _ = build_energy_grid(gap, energy_min_factor, energy_max_factor, num_energy_bins)
    K = scattering_kernel(E, gap, tau_s, T_c, bath_temperature)
    boltz = np.exp(np.clip((E[:, None] - E[None, :]) / (KB_UEV_PER_K * bath_temperature), -200.0, 200.0))
    err = float(np.max(np.abs(K - K.T * boltz)) / max(1e-30, float(np.max(np.abs(K)))))
    return {"passed": err <= tolerance, "max_relative_error": err, "tolerance": tolerance}


def validate_thermal_stability(*, nx, dt, steps, diffusion_coefficient, gap, energy_min_factor, energy_max_factor,
                               num_energy_bins, dynes_gamma, tau_s, tau_r, T_c, bath_temperature,
                               tolerance: float = 1e-6, **solver_kw) -> dict[str, Any]:
    """The thermal spectrum with thermal phonons must not drift (validation.py:102-167)."""
    mask, edges, bcs = _strip(nx)
    E, dE = build_energy_grid(gap, energy_min_factor, energy_max_factor, num_energy_bins)
    n_eq = thermal_qp_weights(E, gap, bath_temperature, dynes_gamma)
    out = run_2d_crank_nicolson(
        mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.full((1, nx), float(np.sum(n_eq) * dE)),
        diffusion_coefficient=diffusion_coefficient, dt=dt, total_time=steps * dt, dx=1.0, store_every=1,
        energy_gap=gap, energy_min_factor=energy_min_factor, energy_max_factor=energy_max_factor,
        num_energy_bins=num_energy_bins, energy_weights=n_eq, enable_diffusion=True, enable_recombination=True,
        enable_scattering=True, dynes_gamma=dynes_gamma, tau_s=tau_s, tau_r=tau_r, T_c=T_c,
        bath_temperature=bath_temperature, initial_condition_spec=_thermal_phonons(bath_temperature),
        freeze_phonon_dynamics=True, **solver_kw)
    frames = out[4]
    if frames is None:
        return {"passed": False, "max_relative_drift": float("inf"), "tolerance": tolerance}
    first = np.array([f[0, :] for f in frames[0]])
    last = np.array([f[0, :] for f in frames[-1]])
    drift = float(np.max(np.abs(last - first)) / max(1e-20, float(np.max(np.abs(first)))))
    return {"passed": drift <= tolerance, "max_relative_drift": drift, "tolerance": tolerance}


def _mass_drift(mass) -> float:
    return float(abs(mass[-1] - mass[0]) / max(1e-20, abs(mass[0])))


def validate_pure_diffusion(*, nx, dt, total_time, diffusion_coefficient, tolerance: float = 1e-10,
                            **solver_kw) -> dict[str, Any]:
    """Reflective walls conserve the integral (validation.py:170-195)."""
    mask, edges, bcs = _strip(nx)
    x = (np.arange(nx, dtype=float) + 0.5) / nx
    mass = run_2d_crank_nicolson(mask=mask, edges=edges, edge_conditions=bcs,
                                 initial_field=(1.0 + 0.4 * np.cos(2.0 * np.pi * x))[None, :],
                                 diffusion_coefficient=diffusion_coefficient, dt=dt, total_time=total_time, dx=1.0,
                                 store_every=1, energy_gap=0.0, enable_diffusion=True, **solver_kw)[2]
    d = _mass_drift(mass)
    return {"passed": d <= tolerance, "mass_relative_drift": d, "tolerance": tolerance}


def validate_pure_scattering(*, nx, dt, steps, gap, energy_min_factor, energy_max_factor, num_energy_bins, dynes_gamma,
                             tau_s, T_c, bath_temperature, tolerance: float = 2e-5, **solver_kw) -> dict[str, Any]:
    """Scattering only redistributes in energy (validation.py:198-241)."""
    mask, edges, bcs = _strip(nx)
    E, _ = build_energy_grid(gap, energy_min_factor, energy_max_factor, num_energy_bins)
    mass = run_2d_crank_nicolson(
        mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.full((1, nx), 2e-4), diffusion_coefficient=6.0,
        dt=dt, total_time=steps * dt, dx=1.0, store_every=1, energy_gap=gap, energy_min_factor=energy_min_factor,
        energy_max_factor=energy_max_factor, num_energy_bins=num_energy_bins,
        energy_weights=np.exp(-((E - 2.6 * gap) / (0.6 * gap)) ** 2), enable_diffusion=False,
        enable_recombination=False, enable_scattering=True, dynes_gamma=dynes_gamma, tau_s=tau_s, T_c=T_c,
        bath_temperature=bath_temperature, initial_condition_spec=_thermal_phonons(bath_temperature),
        freeze_phonon_dynamics=True, **solver_kw)[2]
    d = _mass_drift(mass)
    return {"passed": d <= tolerance, "mass_relative_drift": d, "tolerance": tolerance}


def validate_pure_recombination(*, dt, steps, gap, tau_r, T_c, tolerance_nonincreasing: float = 1e-15,
                                **solver_kw) -> dict[str, Any]:
    """At T = 0 the density can only fall (validation.py:244-283)."""
    mask, edges, bcs = _strip(1)
    mass = run_2d_crank_nicolson(
        mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.array([[1e-3]]), diffusion_coefficient=6.0,
        dt=dt, total_time=steps * dt, dx=1.0, store_every=1, energy_gap=gap, energy_min_factor=1.5,
        energy_max_factor=1.5, num_energy_bins=1, enable_diffusion=False, enable_recombination=True,
        enable_scattering=False, dynes_gamma=0.0, tau_r=tau_r, T_c=T_c, bath_temperature=0.0,
        initial_condition_spec=_thermal_phonons(0.0), freeze_phonon_dynamics=True, **solver_kw)[2]
    ok = all(mass[i + 1] <= mass[i] + tolerance_nonincreasing for i in range(len(mass) - 1))
    return {"passed": bool(ok), "mass_start": mass[0], "mass_end": mass[-1]}


def run_fast_validation_suite(params: SimulationParameters | None = None, **solver_kw) -> ValidationReport:
    """All five checks with the reference's default parameter set (validation.py:286-365)."""
    p = params or SimulationParameters(
        diffusion_coefficient=6.0, dt=0.1, total_time=1.0, mesh_size=1.0, energy_gap=180.0, energy_min_factor=1.0,
        energy_max_factor=4.0, num_energy_bins=24, dynes_gamma=0.18, enable_diffusion=True, enable_recombination=True,
        enable_scattering=True, tau_s=440.0, tau_r=440.0, T_c=1.2, bath_temperature=0.1)
    tau_s = float(p.tau_s if p.tau_s is not None else p.tau_0)
    tau_r = float(p.tau_r if p.tau_r is not None else p.tau_0)
    grid = dict(gap=p.energy_gap, energy_min_factor=p.energy_min_factor, energy_max_factor=p.energy_max_factor)
    return ValidationReport(
        detailed_balance=validate_detailed_balance(**grid, num_energy_bins=p.num_energy_bins, tau_s=tau_s, T_c=p.T_c,
                                                   bath_temperature=p.bath_temperature),
        thermal_stability=validate_thermal_stability(
            nx=16, dt=min(0.1, p.dt), steps=5, diffusion_coefficient=p.diffusion_coefficient, **grid,
            num_energy_bins=p.num_energy_bins, dynes_gamma=p.dynes_gamma, tau_s=tau_s, tau_r=tau_r, T_c=p.T_c,
            bath_temperature=p.bath_temperature, **solver_kw),
        pure_diffusion=validate_pure_diffusion(nx=64, dt=min(0.2, p.dt), total_time=2.0,
                                               diffusion_coefficient=p.diffusion_coefficient, **solver_kw),
        pure_scattering=validate_pure_scattering(
            nx=8, dt=min(0.05, p.dt), steps=10, **grid, num_energy_bins=max(12, p.num_energy_bins),
            dynes_gamma=p.dynes_gamma, tau_s=tau_s, T_c=p.T_c, bath_temperature=p.bath_temperature, **solver_kw),
        pure_recombination=validate_pure_recombination(dt=min(0.1, p.dt), steps=20, gap=p.energy_gap, tau_r=tau_r,
                                                       T_c=p.T_c, **solver_kw),
    )
