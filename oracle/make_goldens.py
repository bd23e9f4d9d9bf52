#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference implementation.

TEST INFRASTRUCTURE -- runs ONLY in the build container, where the reference
checkout is mounted read-only at /root/reference.  It imports the reference
package (``qpsim``), calls its public functions on fixed inputs, and writes the
inputs + outputs as small ``.npz`` fixtures under ``tests/golden/``.  Nothing from
the reference's source text is stored; a fixture is arrays + a JSON description of
the call.  The reference never travels to the GPU box, the fixtures do.

Usage (from the repo root):
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py

Two kinds of fixtures are written:

* ``tests/golden/runs/<name>.npz`` -- one recorded call of the reference
  ``run_2d_crank_nicolson`` (qpsim/solver.py:999): every keyword argument (arrays
  as arrays, dataclasses as JSON) and every output.  They are captured with a
  recording wrapper placed around the reference function while the reference's
  own harnesses (validation.py, test_cases.py) or hand-written calls that restate
  the *inputs* of the reference's tests run.
* ``tests/golden/<topic>.npz`` -- direct calls of table/helper functions
  (energy grid, DOS, kernels, phonon map, per-pixel collision update, edge
  extraction, precompute, initial conditions, external generation).
"""
from __future__ import annotations

import dataclasses
import json
import os
import sys
import warnings
from pathlib import Path

import numpy as np

REF_ROOT = Path("/root/reference")
REPO_ROOT = Path(__file__).resolve().parents[1]
GOLDEN = REPO_ROOT / "tests" / "golden"
RUNS = GOLDEN / "runs"

sys.dont_write_bytecode = True
sys.path.insert(0, str(REF_ROOT))
sys.path.insert(0, str(REF_ROOT / "tests"))

import qpsim.solver as ref_solver  # noqa: E402
import qpsim.test_cases as ref_cases  # noqa: E402
import qpsim.validation as ref_validation  # noqa: E402
from qpsim import geometry as ref_geometry  # noqa: E402
from qpsim import initial_conditions as ref_ic  # noqa: E402
from qpsim import precompute as ref_precompute  # noqa: E402
from qpsim.models import (  # noqa: E402
    BoundaryCondition,
    ExternalGenerationSpec,
    InitialConditionSpec,
    SimulationParameters,
)

_ORIG_RUN = ref_solver.run_2d_crank_nicolson


# --------------------------------------------------------------------------- #
# serialisation helpers
# --------------------------------------------------------------------------- #
def edges_to_json(edges) -> list[dict]:
    return [
        {
            "edge_id": e.edge_id,
            "x0": e.x0, "y0": e.y0, "x1": e.x1, "y1": e.y1,
            "normal": e.normal,
            "faces": [[int(f.row), int(f.col), f.direction] for f in e.faces],
        }
        for e in edges
    ]


def bcs_to_json(bcs) -> dict:
    return {k: {"kind": v.kind, "value": v.value, "aux_value": v.aux_value} for k, v in bcs.items()}


def _jsonable(obj):
    if dataclasses.is_dataclass(obj):
        return {k: _jsonable(v) for k, v in dataclasses.asdict(obj).items()}
    if isinstance(obj, dict):
        return {str(k): _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, (np.floating, np.integer)):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    return obj


class Recorder:
    """Wraps the reference solver entry point and stores every call."""

    def __init__(self) -> None:
        self.calls: list[dict] = []
        self.keep_t0_energy = True

    def __call__(self, *args, **kwargs):
        if args:
            raise RuntimeError("recorder expects keyword-only calls")
        ph_out = kwargs.get("phonon_history_out")
        want_ph = ph_out is not None
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            result = _ORIG_RUN(**kwargs)
        times, frames, mass, clim, eframes, ebins = result
        rec = {
            "kwargs": kwargs,
            "times": np.asarray(times, dtype=float),
            "frames": np.stack(frames),
            "mass": np.asarray(mass, dtype=float),
            "color_limits": np.asarray(clim, dtype=float),
            "energy_frames": None if eframes is None else np.stack([np.stack(ts) for ts in eframes]),
            "E_bins": None if ebins is None else np.asarray(ebins, dtype=float),
            "warnings": [str(w.message) for w in caught],
        }
        if want_ph:
            rec["phonon"] = {
                "phonon_frames": np.stack(ph_out["phonon_frames"]),
                "phonon_energy_frames": None
                if ph_out["phonon_energy_frames"] is None
                else np.stack([np.stack(ts) for ts in ph_out["phonon_energy_frames"]]),
                "phonon_energy_bins": None
                if ph_out["phonon_energy_bins"] is None
                else np.asarray(ph_out["phonon_energy_bins"], dtype=float),
                "phonon_metadata": dict(ph_out["phonon_metadata"]),
            }
        self.calls.append(rec)
        return result


def save_run(name: str, rec: dict, *, tol: float, scheme_note: str, extra: dict | None = None,
             final_only: bool = False) -> None:
    kw = dict(rec["kwargs"])
    arrays: dict[str, np.ndarray] = {}
    meta: dict = {"name": name, "tol": tol, "note": scheme_note, "scalars": {}}
    arrays["mask"] = np.asarray(kw.pop("mask"), dtype=bool)
    arrays["initial_field"] = np.asarray(kw.pop("initial_field"), dtype=float)
    meta["edges"] = edges_to_json(kw.pop("edges"))
    meta["edge_conditions"] = bcs_to_json(kw.pop("edge_conditions"))
    kw.pop("phonon_history_out", None)
    kw.pop("progress_callback", None)
    meta["want_phonon_history"] = "phonon" in rec
    ew = kw.pop("energy_weights", None)
    if ew is not None:
        arrays["energy_weights"] = np.asarray(ew, dtype=float)
    pre = kw.pop("precomputed", None)
    if pre is not None:
        meta["precomputed_keys"] = sorted(pre.keys())
        for k, v in pre.items():
            arrays[f"pre__{k}"] = np.asarray(v)
    gen = kw.pop("external_generation", None)
    meta["external_generation"] = None if gen is None else _jsonable(gen)
    ics = kw.pop("initial_condition_spec", None)
    meta["initial_condition_spec"] = None if ics is None else _jsonable(ics)
    for k, v in kw.items():
        meta["scalars"][k] = _jsonable(v)
    meta["warnings"] = rec["warnings"]
    meta["final_only"] = bool(final_only)
    if extra:
        meta["extra"] = _jsonable(extra)

    arrays["out_times"] = rec["times"]
    arrays["out_mass"] = rec["mass"]
    arrays["out_color_limits"] = rec["color_limits"]
    if final_only:
        arrays["out_frames"] = rec["frames"][-1:]
    else:
        arrays["out_frames"] = rec["frames"]
    if rec["energy_frames"] is not None:
        arrays["out_energy_frames"] = rec["energy_frames"][-1:] if final_only else rec["energy_frames"]
        arrays["out_E_bins"] = rec["E_bins"]
    if "phonon" in rec:
        ph = rec["phonon"]
        arrays["out_phonon_frames"] = ph["phonon_frames"][-1:] if final_only else ph["phonon_frames"]
        if ph["phonon_energy_frames"] is not None:
            arrays["out_phonon_energy_frames"] = (
                ph["phonon_energy_frames"][-1:] if final_only else ph["phonon_energy_frames"]
            )
            arrays["out_phonon_energy_bins"] = ph["phonon_energy_bins"]
        meta["phonon_metadata"] = _jsonable(ph["phonon_metadata"])
    arrays["meta_json"] = np.array(json.dumps(meta))
    RUNS.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(RUNS / f"{name}.npz", **arrays)
    size = (RUNS / f"{name}.npz").stat().st_size
    print(f"  run {name:<48s} {size/1024:8.1f} KiB")


def install(rec: Recorder) -> None:
    ref_solver.run_2d_crank_nicolson = rec
    ref_cases.run_2d_crank_nicolson = rec
    ref_validation.run_2d_crank_nicolson = rec


def uninstall() -> None:
    ref_solver.run_2d_crank_nicolson = _ORIG_RUN
    ref_cases.run_2d_crank_nicolson = _ORIG_RUN
    ref_validation.run_2d_crank_nicolson = _ORIG_RUN


def line_geometry(nx: int):
    mask = np.ones((1, nx), dtype=bool)
    edges = ref_geometry.extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition(kind="reflective") for e in edges}
    return mask, edges, bcs


def rect_geometry(ny: int, nx: int, bc: BoundaryCondition | None = None):
    mask = np.ones((ny, nx), dtype=bool)
    edges = ref_geometry.extract_edge_segments(mask)
    bc = bc or BoundaryCondition(kind="reflective")
    return mask, edges, {e.edge_id: bc for e in edges}


def frozen_thermal_ic(bath_temperature: float) -> InitialConditionSpec:
    # inputs of tests/test_old_mkid_simulation_parity.py:66-76 and validation.py:23-33
    return InitialConditionSpec(
        spatial_kind="uniform",
        spatial_params={"value": 1.0},
        energy_kind="dos",
        energy_params={},
        phonon_spatial_kind="uniform",
        phonon_spatial_params={"value": 1.0},
        phonon_energy_kind="bose_einstein",
        phonon_energy_params={"temperature": float(bath_temperature)},
    )


# --------------------------------------------------------------------------- #
# recorded solver runs
# --------------------------------------------------------------------------- #
def gen_crosscheck() -> None:
    """G1: tests/test_mkid_crosscheck.py:108-165 configuration + that test's own 1-D reference."""
    import test_mkid_crosscheck as xc

    nx, ne, dt, steps = 48, 12, 0.1, 12
    gap, fmin, fmax, D0, gamma, tau, T_c, T_b, rate = 180.0, 1.0, 3.0, 6.0, 0.18, 400.0, 1.2, 0.1, 2e-8
    mask, edges, bcs = line_geometry(nx)
    E_bins, dE = ref_solver.build_energy_grid(gap, fmin, fmax, ne)
    initial_spatial = 1e-4 + 2e-4 * np.exp(-(((np.arange(nx) + 0.5) / nx - 0.3) ** 2) / (2.0 * 0.06 ** 2))
    weights = ref_solver.thermal_qp_weights(E_bins, gap, T_b, gamma)
    weights = weights / (np.sum(weights) * dE)
    rec = Recorder()
    rec(
        mask=mask, edges=edges, edge_conditions=bcs, initial_field=initial_spatial.reshape(1, nx),
        diffusion_coefficient=D0, dt=dt, total_time=dt * steps, dx=1.0, store_every=1,
        energy_gap=gap, energy_min_factor=fmin, energy_max_factor=fmax, num_energy_bins=ne,
        energy_weights=weights, enable_diffusion=True, enable_recombination=True, enable_scattering=True,
        dynes_gamma=gamma, tau_0=tau, T_c=T_c, bath_temperature=T_b,
        external_generation=ExternalGenerationSpec(mode="constant", rate=rate),
    )
    K_r = ref_solver.recombination_kernel(E_bins, gap, tau, T_c, T_b)
    K_s = ref_solver.scattering_kernel(E_bins, gap, tau, T_c, T_b)
    rho = ref_solver._dynes_density_of_states(E_bins, gap, gamma)
    n_th = ref_solver.thermal_qp_weights(E_bins, gap, T_b, gamma)
    D_bins = D0 * np.sqrt(np.maximum(0.0, 1.0 - (gap / E_bins) ** 2))
    state_ref = xc._mkid_like_reference_1d(
        nx=nx, ne=ne, dt=dt, steps=steps, dE=dE, D_bins=D_bins, K_r=K_r, K_s=K_s, rho=rho,
        n_thermal=n_th, weights=weights, initial_spatial=initial_spatial, generation_rate=rate,
    )
    save_run("xcheck_mkid_1x48_ne12", rec.calls[0], tol=1e-10,
             scheme_note="strip: ADI == unsplit CN; north_star parity target 1e-10",
             extra={"mkid_like_reference_1d": state_ref})


def gen_test_suite() -> None:
    """G2/G4: qpsim/test_cases.py generators (strip 1x100, rectangle 36x56, donut 64x64, ODE cases)."""
    rec = Recorder()
    install(rec)
    try:
        ref_cases._generate_strip_geometry_group(
            nx=100, dx=1.0, diffusion_coefficient=25.0, dt=0.05, total_time=8.0, store_every=8)
        n_strip = len(rec.calls)
        ref_cases._generate_rectangle_geometry_group(
            dx=1.0, diffusion_coefficient=25.0, dt=0.05, total_time=2.0, store_every=20)
        n_rect = len(rec.calls)
        ref_cases._generate_polygon_donut_geometry_group(
            dx=1.0, diffusion_coefficient=25.0, dt=0.05, total_time=2.0, store_every=20)
        n_donut = len(rec.calls)
        ref_cases._generate_recombination_test_group()
        n_rec = len(rec.calls)
        ref_cases._generate_scattering_test_group()
    finally:
        uninstall()
    for i, call in enumerate(rec.calls):
        if i < n_strip:
            save_run(f"suite_strip_{i:02d}", call, tol=1e-11, scheme_note="strip scalar CN, all BC kinds")
        elif i < n_rect:
            save_run(f"suite_rect_{i - n_strip:02d}", call, tol=1e-9,
                     scheme_note="2-D rectangle: compare in cn_exact mode; adi differs by splitting error")
        elif i < n_donut:
            save_run(f"suite_donut_{i - n_rect:02d}", call, tol=1e-9,
                     scheme_note="2-D masked donut: compare in cn_exact mode")
        elif i < n_rec:
            save_run(f"suite_recomb_{i - n_donut:02d}", call, tol=1e-10, scheme_note="collision ODE case",
                     final_only=call["frames"].shape[0] > 12)
        else:
            save_run(f"suite_scatter_{i - n_rec:02d}", call, tol=1e-10, scheme_note="collision ODE case",
                     final_only=call["frames"].shape[0] > 12)


def gen_validation() -> None:
    """G8: validation.run_fast_validation_suite (validation.py:286-365) calls + report values."""
    rec = Recorder()
    install(rec)
    try:
        report = ref_validation.run_fast_validation_suite().as_dict()
    finally:
        uninstall()
    names = ["thermal_stability", "pure_diffusion", "pure_scattering", "pure_recombination"]
    assert len(rec.calls) == len(names)
    for nm, call in zip(names, rec.calls):
        save_run(f"validation_{nm}", call, tol=1e-10, scheme_note="validation suite strip case")
    (GOLDEN / "validation_report.json").write_text(json.dumps(_jsonable(report), indent=1))


def gen_legacy_parity_inputs() -> None:
    """G3: the 'current solver' halves of tests/test_old_mkid_simulation_parity.py (legacy side absent)."""
    rec = Recorder()
    gap, fmin, fmax, D0 = 180.0, 1.0, 3.0, 6.0
    # :189-221 collision-only single step, delta spectrum
    ne, nx, k, n0 = 18, 32, 9, 1e-6
    mask, edges, bcs = line_geometry(nx)
    _, dE = ref_solver.build_energy_grid(gap, fmin, fmax, ne)
    w = np.zeros(ne)
    w[k] = 1.0
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.full((1, nx), n0 * dE),
        diffusion_coefficient=D0, dt=0.001, total_time=0.001, dx=1.0, store_every=1, energy_gap=gap,
        energy_min_factor=fmin, energy_max_factor=fmax, num_energy_bins=ne, energy_weights=w,
        enable_diffusion=False, enable_recombination=True, enable_scattering=True, dynes_gamma=0.0,
        collision_solver="fischer_catelani_local", tau_s=400.0, tau_r=500.0, T_c=1.2, bath_temperature=0.0,
        initial_condition_spec=frozen_thermal_ic(0.0), freeze_phonon_dynamics=True)
    save_run("legacy_collision_single_step", rec.calls[-1], tol=1e-11, scheme_note="collision only")
    # :237-352 frozen phonons, three profiles
    x = (np.arange(nx, dtype=float) + 0.5) / nx
    for T_b, kk, steps, kind in [(0.0, 3, 1, "flat"), (0.10, 7, 6, "gaussian"), (0.20, 12, 8, "skewed")]:
        if kind == "flat":
            prof = np.full(nx, 1.1e-6)
        elif kind == "gaussian":
            prof = 8e-7 + 2.1e-6 * np.exp(-((x - 0.37) ** 2) / (2.0 * 0.09 ** 2))
        else:
            prof = 5e-7 + 1.9e-6 * (0.2 + 0.8 * x ** 1.5)
        w = np.zeros(ne)
        w[kk] = 1.0
        rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=(prof * dE)[None, :],
            diffusion_coefficient=D0, dt=0.001, total_time=steps * 0.001, dx=1.0, store_every=max(1, steps),
            energy_gap=gap, energy_min_factor=fmin, energy_max_factor=fmax, num_energy_bins=ne,
            energy_weights=w, enable_diffusion=False, enable_recombination=True, enable_scattering=True,
            dynes_gamma=0.0, collision_solver="fischer_catelani_local", tau_s=400.0, tau_r=500.0, T_c=1.2,
            bath_temperature=T_b, initial_condition_spec=frozen_thermal_ic(T_b), freeze_phonon_dynamics=True)
        save_run(f"legacy_frozen_phonons_{kind}", rec.calls[-1], tol=1e-11, scheme_note="collision only")
    # :409-439 diffusion-only single CN step
    ne, nx, k = 12, 48, 2
    mask, edges, bcs = line_geometry(nx)
    _, dE = ref_solver.build_energy_grid(gap, fmin, fmax, ne)
    x = (np.arange(nx, dtype=float) + 0.5) / nx
    prof = 1e-6 + 2e-6 * np.exp(-((x - 0.33) ** 2) / (2.0 * 0.08 ** 2))
    w = np.zeros(ne)
    w[k] = 1.0
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=(prof * dE)[None, :],
        diffusion_coefficient=D0, dt=0.01, total_time=0.01, dx=1.0, store_every=1, energy_gap=gap,
        energy_min_factor=fmin, energy_max_factor=fmax, num_energy_bins=ne, energy_weights=w,
        enable_diffusion=True, enable_recombination=False, enable_scattering=False, dynes_gamma=0.0,
        tau_s=400.0, tau_r=500.0, T_c=1.2, bath_temperature=0.0)
    save_run("legacy_diffusion_single_step", rec.calls[-1], tol=1e-12, scheme_note="strip CN step, 1e-12 in reference test")
    # :525-565 pulse injection through custom generation, diffusion only
    ne, nx, steps, dt = 14, 60, 40, 0.01
    mask, edges, bcs = line_geometry(nx)
    E_bins, _ = ref_solver.build_energy_grid(gap, fmin, fmax, ne)
    gen = ExternalGenerationSpec(
        mode="custom",
        custom_body=(
            "((t <= params['t_end']) and (abs(E - params['E_inj']) < params.get('E_tol', 1e-12))) "
            "* np.where(np.arange(params['nx']) == params['px'], params['rate'], 0.0)"
        ),
        custom_params={"px": nx // 3, "nx": nx, "rate": 1e-5, "t_end": 0.10, "E_inj": float(E_bins[4]), "E_tol": 1e-9},
    )
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.zeros((1, nx)),
        diffusion_coefficient=D0, dt=dt, total_time=steps * dt, dx=1.0, store_every=1, energy_gap=gap,
        energy_min_factor=fmin, energy_max_factor=fmax, num_energy_bins=ne, enable_diffusion=True,
        enable_recombination=False, enable_scattering=False, dynes_gamma=0.0, external_generation=gen)
    save_run("legacy_pulse_injection", rec.calls[-1], tol=1e-11, scheme_note="custom generation + strip diffusion",
             final_only=True)
    # :762-814 variable-D diffusion through precompute
    nx, ne, steps, dt, gap0, slope = 80, 10, 20, 0.05, 180.0, 0.4
    mask, edges, bcs = line_geometry(nx)
    params = SimulationParameters(
        diffusion_coefficient=D0, dt=dt, total_time=steps * dt, mesh_size=1.0, store_every=steps,
        energy_gap=gap0, energy_min_factor=1.4, energy_max_factor=2.6, num_energy_bins=ne, dynes_gamma=0.0,
        gap_expression=f"return {gap0} * (1.0 + {slope} * (x - 0.5))", collision_solver="fischer_catelani_local",
        enable_diffusion=True, enable_recombination=False, enable_scattering=False, tau_s=1e30, tau_r=1e30,
        T_c=1.2, bath_temperature=0.0)
    pre = ref_precompute.precompute_arrays(mask, edges, bcs, params)
    _, dE = ref_solver.build_energy_grid(gap0, 1.4, 2.6, ne)
    x = (np.arange(nx, dtype=float) + 0.5) / nx
    prof = 1e-6 + 4e-6 * np.exp(-((x - 0.33) ** 2) / (2.0 * 0.08 ** 2))
    w = np.zeros(ne)
    w[ne - 2] = 1.0
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=(prof * dE)[None, :],
        diffusion_coefficient=D0, dt=dt, total_time=steps * dt, dx=1.0, store_every=steps, energy_gap=gap0,
        energy_min_factor=1.4, energy_max_factor=2.6, num_energy_bins=ne, energy_weights=w,
        enable_diffusion=True, enable_recombination=False, enable_scattering=False, dynes_gamma=0.0,
        tau_s=1e30, tau_r=1e30, T_c=1.2, bath_temperature=0.0, precomputed=pre)
    save_run("legacy_variable_diffusion", rec.calls[-1], tol=1e-11, scheme_note="variable D(x) strip")


def gen_regressions() -> None:
    """G9 + regression inputs (tests/test_regressions.py:232-271,320-395; test_phonon_scaffold.py:128-153)."""
    rec = Recorder()
    mask, edges, bcs = rect_geometry(2, 2)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.full((2, 2), 3.0),
        diffusion_coefficient=1.0, dt=0.2, total_time=1.0, dx=1.0, store_every=1)
    save_run("regress_stationary_2x2", rec.calls[-1], tol=1e-12, scheme_note="uniform field stays put")
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.array([[1.0, 2.0], [0.5, 4.0]]),
        diffusion_coefficient=1.0, dt=0.3, total_time=1.0, dx=1.0, store_every=1)
    save_run("regress_remainder_step_2x2", rec.calls[-1], tol=1e-9, scheme_note="remainder dt; 2-D -> cn_exact")
    # precompute uniform == direct (3x3, NE=10, recombination)
    mask, edges, bcs = rect_geometry(3, 3)
    params = SimulationParameters(
        diffusion_coefficient=6.0, dt=1.0, total_time=3.0, mesh_size=1.0, store_every=1, energy_gap=180.0,
        energy_max_factor=5.0, num_energy_bins=10, enable_diffusion=True, enable_recombination=True,
        tau_0=440.0, T_c=1.2, bath_temperature=0.1)
    pre = ref_precompute.precompute_arrays(mask, edges, bcs, params)
    common = dict(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.full((3, 3), 1.0),
                  diffusion_coefficient=6.0, dt=1.0, total_time=3.0, dx=1.0, store_every=1, energy_gap=180.0,
                  energy_max_factor=5.0, num_energy_bins=10, enable_diffusion=True, enable_recombination=True,
                  tau_0=440.0, T_c=1.2, bath_temperature=0.1)
    rec(**common, precomputed=pre)
    save_run("regress_precompute_uniform_3x3", rec.calls[-1], tol=1e-9, scheme_note="2-D NE=10 recombination")
    rec(**common)
    save_run("regress_direct_uniform_3x3", rec.calls[-1], tol=1e-9, scheme_note="2-D NE=10 recombination")
    # non-uniform gap 4x4, NE=5 (variable-D operators + nonuniform kernels when collisions on)
    mask, edges, bcs = rect_geometry(4, 4)
    params = SimulationParameters(
        diffusion_coefficient=6.0, dt=1.0, total_time=2.0, mesh_size=1.0, store_every=1, energy_gap=180.0,
        energy_max_factor=5.0, num_energy_bins=5, enable_diffusion=True, gap_expression="return 180 + 20 * x")
    pre = ref_precompute.precompute_arrays(mask, edges, bcs, params)
    init = 1.0 + 0.5 * np.cos(np.arange(16.0)).reshape(4, 4)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=1.0,
        total_time=2.0, dx=1.0, store_every=1, energy_gap=180.0, energy_max_factor=5.0, num_energy_bins=5,
        enable_diffusion=True, precomputed=pre)
    save_run("regress_nonuniform_gap_4x4", rec.calls[-1], tol=1e-9, scheme_note="variable D 2-D")
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=1e-4 * init, diffusion_coefficient=6.0, dt=0.5,
        total_time=2.0, dx=1.0, store_every=2, energy_gap=180.0, energy_max_factor=3.0, num_energy_bins=6,
        enable_diffusion=True, enable_recombination=True, enable_scattering=True, precomputed=None,
        gap_expression="return 180 + 20 * x", bath_temperature=0.15, phonon_history_out={})
    save_run("regress_nonuniform_gap_collisions_4x4", rec.calls[-1], tol=1e-9,
             scheme_note="auto-precompute from gap_expression, nonuniform kernels, dynamic phonons")
    # phonon scaffold 2x3, NE=6, full physics, dynamic phonons
    mask, edges, bcs = rect_geometry(2, 3)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.full((2, 3), 0.25),
        diffusion_coefficient=6.0, dt=0.2, total_time=1.0, dx=1.0, store_every=1, energy_gap=180.0,
        energy_min_factor=1.0, energy_max_factor=3.0, num_energy_bins=6, enable_diffusion=True,
        enable_recombination=True, enable_scattering=True, tau_0=440.0, T_c=1.2, bath_temperature=0.1,
        phonon_history_out={})
    save_run("phonon_scaffold_2x3_ne6", rec.calls[-1], tol=1e-9, scheme_note="2-D full physics with phonon history")
    # 16x16 / NE=8 variant with mixed BCs and dynamic phonons
    mask = np.ones((16, 16), dtype=bool)
    edges = ref_geometry.extract_edge_segments(mask)
    bcs = {}
    for e in edges:
        bcs[e.edge_id] = {
            "left": BoundaryCondition(kind="dirichlet", value=2e-5),
            "right": BoundaryCondition(kind="absorbing"),
            "up": BoundaryCondition(kind="neumann", value=1e-6),
            "down": BoundaryCondition(kind="robin", value=0.3, aux_value=2e-6),
        }[e.normal]
    rng = np.random.default_rng(7)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=1e-4 * (1.0 + rng.random((16, 16))),
        diffusion_coefficient=6.0, dt=0.1, total_time=1.0, dx=1.0, store_every=5, energy_gap=180.0,
        energy_min_factor=1.0, energy_max_factor=3.0, num_energy_bins=8, enable_diffusion=True,
        enable_recombination=True, enable_scattering=True, dynes_gamma=0.2, tau_s=400.0, tau_r=500.0, T_c=1.2,
        bath_temperature=0.2, external_generation=ExternalGenerationSpec(mode="pulse", pulse_start=0.2,
                                                                         pulse_duration=0.35, pulse_rate=3e-8),
        phonon_history_out={})
    save_run("full_physics_16x16_ne8_mixed_bc", rec.calls[-1], tol=1e-9,
             scheme_note="2-D, four BC kinds, pulse generation, dynamic phonons")
    # scalar-mode phonon scaffold
    mask, edges, bcs = rect_geometry(3, 4)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.arange(12.0).reshape(3, 4),
        diffusion_coefficient=2.0, dt=0.1, total_time=0.3, dx=0.5, store_every=1, bath_temperature=0.125,
        phonon_history_out={})
    save_run("scalar_fixed_phonon_history_3x4", rec.calls[-1], tol=1e-9, scheme_note="scalar mode, dx != 1")
    # masked (non-rectangular) scalar case with holes and mixed BCs
    mask = np.ones((9, 11), dtype=bool)
    mask[3:6, 4:7] = False
    mask[0, 0] = False
    mask[8, 9:] = False
    edges = ref_geometry.extract_edge_segments(mask)
    kinds = [BoundaryCondition(kind="reflective"), BoundaryCondition(kind="dirichlet", value=0.4),
             BoundaryCondition(kind="absorbing"), BoundaryCondition(kind="neumann", value=-0.05),
             BoundaryCondition(kind="robin", value=0.5, aux_value=0.1)]
    bcs = {e.edge_id: kinds[i % len(kinds)] for i, e in enumerate(edges)}
    rng = np.random.default_rng(11)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=rng.random((9, 11)),
        diffusion_coefficient=3.0, dt=0.07, total_time=1.0, dx=0.8, store_every=5)
    save_run("masked_holes_mixed_bc_9x11", rec.calls[-1], tol=1e-9, scheme_note="masked 2-D, all BC kinds, remainder step")
    # single-bin energy mode (build_energy_grid num_energy_bins == 1 branch)
    mask, edges, bcs = line_geometry(1)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.array([[1e-3]]), diffusion_coefficient=6.0,
        dt=0.1, total_time=2.0, dx=1.0, store_every=1, energy_gap=180.0, energy_min_factor=1.5,
        energy_max_factor=1.5, num_energy_bins=1, enable_diffusion=False, enable_recombination=True,
        enable_scattering=False, dynes_gamma=0.0, tau_r=440.0, T_c=1.2, bath_temperature=0.0,
        initial_condition_spec=frozen_thermal_ic(0.0), freeze_phonon_dynamics=True)
    save_run("single_bin_recombination_1x1", rec.calls[-1], tol=1e-11, scheme_note="NE=1")
    # custom IC spec (qp full custom + point phonon) on a small rectangle
    mask, edges, bcs = rect_geometry(4, 5)
    spec = InitialConditionSpec(
        spatial_kind="gaussian", spatial_params={"amplitude": 1e-4, "x0": 0.4, "y0": 0.6, "sigma": 0.2},
        energy_kind="fermi_dirac", energy_params={"temperature": 0.4},
        qp_full_custom_enabled=True,
        qp_full_custom_body="return 1e-5 * np.exp(-((x-0.5)**2 + (y-0.5)**2) / 0.1) * np.exp(-(E-180.0) / 200.0)",
        phonon_spatial_kind="gaussian", phonon_spatial_params={"amplitude": 1.0, "x0": 0.5, "y0": 0.5, "sigma": 0.3},
        phonon_energy_kind="bose_einstein", phonon_energy_params={"temperature": 0.5})
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=np.zeros((4, 5)), diffusion_coefficient=6.0,
        dt=0.1, total_time=0.5, dx=1.0, store_every=5, energy_gap=180.0, energy_min_factor=1.0,
        energy_max_factor=3.0, num_energy_bins=6, enable_diffusion=True, enable_recombination=True,
        enable_scattering=True, T_c=1.2, bath_temperature=0.1, initial_condition_spec=spec, phonon_history_out={})
    save_run("custom_ic_spec_4x5_ne6", rec.calls[-1], tol=1e-9, scheme_note="full custom qp state + IC-spec phonons")


def gen_config1() -> None:
    """BASELINE config 1: 64x64 full mask, 200 CN steps, scalar and NE=8 (frames at t=T only)."""
    rec = Recorder()
    N = 64
    mask, edges, bcs = rect_geometry(N, N)
    init = 1e-4 * (1.0 + np.random.default_rng(0).random((N, N)))
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
        total_time=20.0, dx=1.0, store_every=200)
    save_run("config1_64x64_scalar_200steps", rec.calls[-1], tol=1e-9, scheme_note="BASELINE configs[0], scalar",
             final_only=True)
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
        total_time=2.0, dx=1.0, store_every=20, energy_gap=180.0, energy_min_factor=1.0, energy_max_factor=3.0,
        num_energy_bins=8, enable_diffusion=True)
    save_run("config1_64x64_ne8_20steps", rec.calls[-1], tol=1e-9, scheme_note="BASELINE configs[0], NE=8 diffusion",
             final_only=True)
    N = 24
    mask, edges, bcs = rect_geometry(N, N)
    init = 1e-4 * (1.0 + np.random.default_rng(1).random((N, N)))
    rec(mask=mask, edges=edges, edge_conditions=bcs, initial_field=init, diffusion_coefficient=6.0, dt=0.1,
        total_time=1.0, dx=1.0, store_every=10, energy_gap=180.0, energy_min_factor=1.0, energy_max_factor=3.0,
        num_energy_bins=12, enable_diffusion=True, enable_recombination=True, enable_scattering=True,
        tau_0=440.0, T_c=1.2, bath_temperature=0.1, phonon_history_out={})
    save_run("mkid_24x24_ne12_full_physics", rec.calls[-1], tol=1e-9,
             scheme_note="SURVEY 8d energy-resolved parameters (gap 180, factors 1-3, NE=12) at small N",
             final_only=True)


# --------------------------------------------------------------------------- #
# direct function goldens
# --------------------------------------------------------------------------- #
def gen_tables() -> None:
    out: dict[str, np.ndarray] = {}
    cases = [
        ("a", 180.0, 1.0, 3.0, 12, 0.0, 400.0, 500.0, 1.2, 0.1),
        ("b", 180.0, 1.0, 10.0, 50, 0.18, 440.0, 440.0, 1.2, 0.1),
        ("c", 180.0, 1.0, 3.0, 24, 0.0, 400.0, 500.0, 1.2, 0.0),
        ("d", 200.0, 1.4, 2.6, 6, 5.0, 300.0, 900.0, 1.5, 0.8),
        ("e", 180.0, 1.0, 4.0, 24, 0.18, 440.0, 440.0, 1.2, 0.1),
        ("f", 180.0, 1.0, 10.0, 18, 0.0, 440.0, 440.0, 1.2, 0.3),
    ]
    meta = {}
    for tag, gap, fmin, fmax, ne, gamma, tau_s, tau_r, T_c, T_b in cases:
        E, dE = ref_solver.build_energy_grid(gap, fmin, fmax, ne)
        om, idx_d, idx_s, sgn = ref_solver._build_phonon_frequency_map(E)
        out[f"{tag}_E"] = E
        out[f"{tag}_dE"] = np.array(dE)
        out[f"{tag}_rho"] = ref_solver._dynes_density_of_states(E, gap, gamma)
        out[f"{tag}_bcs"] = ref_solver._bcs_density_of_states(E, gap)
        out[f"{tag}_qp_weights"] = ref_solver.thermal_qp_weights(E, gap, T_b, gamma)
        out[f"{tag}_Kr0"] = ref_solver.recombination_kernel_base(E, gap, tau_r, T_c)
        out[f"{tag}_Ks0"] = ref_solver.scattering_kernel_base(E, gap, tau_s, T_c)
        out[f"{tag}_Kr"] = ref_solver.recombination_kernel(E, gap, tau_r, T_c, T_b)
        out[f"{tag}_Ks"] = ref_solver.scattering_kernel(E, gap, tau_s, T_c, T_b)
        out[f"{tag}_omega"] = om
        out[f"{tag}_idx_diff"] = idx_d
        out[f"{tag}_idx_sum"] = idx_s
        out[f"{tag}_sign"] = sgn
        out[f"{tag}_nph"] = ref_solver.thermal_phonon_occupation(om, T_b)
        out[f"{tag}_widths"] = ref_solver.integration_widths_from_centers(om, fallback_width=dE)
        meta[tag] = dict(gap=gap, fmin=fmin, fmax=fmax, ne=ne, gamma=gamma, tau_s=tau_s, tau_r=tau_r, T_c=T_c, T_b=T_b)
    E1, dE1 = ref_solver.build_energy_grid(180.0, 1.5, 1.5, 1)
    out["single_E"] = E1
    out["single_dE"] = np.array(dE1)
    out["widths_single"] = ref_solver.integration_widths_from_centers(np.array([3.0]), fallback_width=0.7)
    out["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(GOLDEN / "tables.npz", **out)
    print("  tables.npz")


def gen_collision_vectors() -> None:
    """G5: single calls of the per-pixel coupled update (solver.py:703-791) and the step wrappers (:794-875)."""
    out: dict[str, np.ndarray] = {}
    meta = {}
    rng = np.random.default_rng(20260227)
    idx = 0
    for ne, fmax in [(6, 3.0), (12, 3.0), (24, 4.0), (50, 10.0), (18, 10.0)]:
        gap, gamma, T_c = 180.0, (0.18 if ne != 12 else 0.0), 1.2
        E, dE = ref_solver.build_energy_grid(gap, 1.0, fmax, ne)
        rho = ref_solver._dynes_density_of_states(E, gap, gamma)
        Kr0 = ref_solver.recombination_kernel_base(E, gap, 500.0, T_c)
        Ks0 = ref_solver.scattering_kernel_base(E, gap, 400.0, T_c)
        om, idx_d, idx_s, sgn = ref_solver._build_phonon_frequency_map(E)
        for T_b in (0.0, 0.1, 0.8):
            nph_eq = ref_solver.thermal_phonon_occupation(om, T_b)
            for (en_r, en_s) in [(True, True), (True, False), (False, True), (False, False)]:
                if ne == 50 and not (en_r and en_s) and T_b != 0.1:
                    continue
                npx = 5
                # occupations from tiny to near Pauli blocking; one pixel exactly zero
                f_occ = rng.random((ne, npx)) * np.array([1e-6, 1e-3, 0.2, 0.9, 0.0])[None, :]
                state = f_occ * np.maximum(rho, 0.0)[:, None]
                ph = nph_eq[:, None] * (0.5 + rng.random((om.size, npx))) + 1e-3 * rng.random((om.size, npx))
                ph[:, 4] = 0.0
                for dt in (0.05, 5.0):
                    s_new = state.copy()
                    p_new = ph.copy()
                    ref_solver.apply_collision_step_fischer_catelani_uniform(
                        s_new, p_new, Kr0 if en_r else None, Ks0 if en_s else None, rho, idx_d, idx_s, sgn,
                        dE, dt, enable_recombination=en_r, enable_scattering=en_s, update_phonons=True)
                    tag = f"c{idx:03d}"
                    out[f"{tag}_state_in"] = state
                    out[f"{tag}_ph_in"] = ph
                    out[f"{tag}_state_out"] = s_new
                    out[f"{tag}_ph_out"] = p_new
                    meta[tag] = dict(ne=ne, fmax=fmax, gap=gap, gamma=gamma, T_c=T_c, tau_r=500.0, tau_s=400.0,
                                     T_b=T_b, en_r=en_r, en_s=en_s, dt=dt, dE=float(dE))
                    idx += 1
    # non-uniform kernels: three gap classes over 7 pixels
    ne = 10
    E, dE = ref_solver.build_energy_grid(180.0, 1.0, 3.0, ne)
    om, idx_d, idx_s, sgn = ref_solver._build_phonon_frequency_map(E)
    gaps = np.array([175.0, 180.0, 192.5, 180.0, 175.0, 192.5, 180.0])
    rho_all = np.stack([ref_solver._dynes_density_of_states(E, g, 0.1) for g in gaps])
    Kr_all = np.stack([ref_solver.recombination_kernel_base(E, g, 450.0, 1.2) for g in gaps])
    Ks_all = np.stack([ref_solver.scattering_kernel_base(E, g, 410.0, 1.2) for g in gaps])
    state = rng.random((ne, 7)) * 0.3 * rho_all.T
    ph = ref_solver.thermal_phonon_occupation(om, 0.3)[:, None] * (0.5 + rng.random((om.size, 7)))
    s_new, p_new = state.copy(), ph.copy()
    ref_solver.apply_collision_step_fischer_catelani_nonuniform(
        s_new, p_new, Kr_all, Ks_all, rho_all, idx_d, idx_s, sgn, dE, 0.3,
        enable_recombination=True, enable_scattering=True, update_phonons=True)
    out["nonuni_gaps"] = gaps
    out["nonuni_state_in"], out["nonuni_ph_in"] = state, ph
    out["nonuni_state_out"], out["nonuni_ph_out"] = s_new, p_new
    meta["nonuni"] = dict(ne=ne, fmax=3.0, gap=180.0, gamma=0.1, T_c=1.2, tau_r=450.0, tau_s=410.0, dt=0.3, dE=float(dE))
    # dead-code explicit Euler helpers (solver.py:551-605)
    E, dE = ref_solver.build_energy_grid(180.0, 1.0, 3.0, 8)
    rho = ref_solver._dynes_density_of_states(E, 180.0, 0.0)
    K_s = ref_solver.scattering_kernel(E, 180.0, 400.0, 1.2, 0.2)
    K_r = ref_solver.recombination_kernel(E, 180.0, 500.0, 1.2, 0.2)
    n_eq = ref_solver.thermal_qp_weights(E, 180.0, 0.2, 0.0)
    G = 2.0 * n_eq * dE * (K_r @ n_eq)
    st = rng.random((8, 4)) * 1e-3 * rho[:, None]
    a = st.copy()
    ref_solver.apply_scattering_step(a, K_s, rho, dE, 0.05)
    b = st.copy()
    ref_solver.apply_recombination_step(b, K_r, G, dE, 0.05)
    out["euler_state_in"], out["euler_scat_out"], out["euler_recomb_out"], out["euler_G"] = st, a, b, G
    out["euler_rhs_px0"] = ref_solver._collision_rhs(st[:, 0], K_r, K_s, rho, G, dE)
    out["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(GOLDEN / "collision_vectors.npz", **out)
    print(f"  collision_vectors.npz ({idx} uniform cases)")


def gen_geometry() -> None:
    masks = {}
    masks["strip_1x7"] = np.ones((1, 7), dtype=bool)
    masks["col_5x1"] = np.ones((5, 1), dtype=bool)
    masks["rect_3x5"] = np.ones((3, 5), dtype=bool)
    m = np.ones((9, 11), dtype=bool)
    m[3:6, 4:7] = False
    m[0, 0] = False
    m[8, 9:] = False
    masks["holes_9x11"] = m
    masks["scaffold_3x4"] = np.array([[1, 1, 0, 0], [0, 1, 1, 0], [0, 0, 1, 1]], dtype=bool)
    intrinsic = ref_geometry.create_intrinsic_geometry()
    masks["intrinsic_64x120"] = np.asarray(intrinsic.mask, dtype=bool)
    donut, *_ = ref_cases._polygon_donut_mask(64, 64)
    masks["donut_64x64"] = np.asarray(donut, dtype=bool)
    rng = np.random.default_rng(3)
    masks["random_12x13"] = rng.random((12, 13)) > 0.35
    out = {}
    meta = {}
    for name, mask in masks.items():
        edges = ref_geometry.extract_edge_segments(mask)
        out[f"{name}_mask"] = mask
        meta[name] = edges_to_json(edges)
    out["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(GOLDEN / "geometry_edges.npz", **out)
    print("  geometry_edges.npz")


def gen_operators() -> None:
    """A2/A3: assembled Laplacians and sources (solver.py:152-212, :235-321) on small masks, as dense arrays."""
    out = {}
    meta = {}
    m = np.ones((5, 6), dtype=bool)
    m[2, 2:4] = False
    m[4, 0] = False
    edges = ref_geometry.extract_edge_segments(m)
    kinds = [BoundaryCondition(kind="reflective"), BoundaryCondition(kind="dirichlet", value=0.4),
             BoundaryCondition(kind="absorbing"), BoundaryCondition(kind="neumann", value=-0.05),
             BoundaryCondition(kind="robin", value=0.5, aux_value=0.1)]
    bcs = {e.edge_id: kinds[i % len(kinds)] for i, e in enumerate(edges)}
    L, src, index_map = ref_solver.build_laplacian_with_boundaries(m, edges, bcs, 0.8)
    rng = np.random.default_rng(5)
    Dsp = 1.0 + rng.random(int(m.sum())) * 5.0
    LD, srcD = ref_solver.build_variable_diffusion_laplacian(m, edges, bcs, 0.8, Dsp)
    out["mask"] = m
    out["L"] = L.toarray()
    out["source"] = src
    out["index_map"] = index_map
    out["D_spatial"] = Dsp
    out["L_D"] = LD.toarray()
    out["source_D"] = srcD
    meta["edges"] = edges_to_json(edges)
    meta["edge_conditions"] = bcs_to_json(bcs)
    meta["dx"] = 0.8
    out["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(GOLDEN / "operators.npz", **out)
    print("  operators.npz")


def gen_precompute_ic_generation() -> None:
    out = {}
    meta = {}
    # precompute (precompute.py:173-287), uniform and non-uniform, with kernels
    mask = np.ones((3, 4), dtype=bool)
    mask[0, 3] = False
    edges = ref_geometry.extract_edge_segments(mask)
    bcs = {e.edge_id: BoundaryCondition(kind="reflective") for e in edges}
    for tag, expr in [("uni", ""), ("non", "return 180 * (1.0 + 0.3 * (x - 0.5)) + 5 * y")]:
        p = SimulationParameters(
            diffusion_coefficient=6.0, dt=0.1, total_time=0.1, mesh_size=1.0, energy_gap=180.0,
            energy_min_factor=1.0, energy_max_factor=3.0, num_energy_bins=7, dynes_gamma=0.05,
            gap_expression=expr, enable_recombination=True, enable_scattering=True, tau_s=400.0, tau_r=500.0,
            T_c=1.2, bath_temperature=0.1)
        for kern in (False, True):
            pre = ref_precompute.precompute_arrays(mask, edges, bcs, p, include_collision_kernels=kern)
            for k, v in pre.items():
                out[f"pre_{tag}_{int(kern)}__{k}"] = np.asarray(v)
        meta[f"pre_{tag}"] = _jsonable(p)
    out["pre_mask"] = mask
    meta["pre_edges"] = edges_to_json(edges)
    out["mask_hash"] = np.array(ref_precompute._mask_hash(mask))
    out["gap_expr_hash"] = np.array(ref_precompute._gap_expression_hash("return 180 + 20 * x"))
    # initial conditions (initial_conditions.py:216-280, 353-412, 544-632)
    m2 = np.ones((6, 7), dtype=bool)
    m2[2:4, 3] = False
    out["ic_mask"] = m2
    specs = {
        "gauss": InitialConditionSpec(spatial_kind="gaussian", spatial_params={"amplitude": 2.0, "x0": 0.3, "y0": 0.6, "sigma": 0.2}),
        "uniform": InitialConditionSpec(spatial_kind="uniform", spatial_params={"value": 0.7}),
        "point_in": InitialConditionSpec(spatial_kind="point", spatial_params={"value": 3.0, "x0": 0.1, "y0": 0.9}),
        "point_hole": InitialConditionSpec(spatial_kind="point", spatial_params={"value": 3.0, "x0": 0.5, "y0": 0.4}),
        "custom": InitialConditionSpec(spatial_kind="custom", spatial_custom_body="return params['a'] * np.sin(3*x) + y", spatial_custom_params={"a": 0.5}),
        "default": InitialConditionSpec(),
    }
    E, dE = ref_solver.build_energy_grid(180.0, 1.0, 3.0, 9)
    om, *_ = ref_solver._build_phonon_frequency_map(E)
    out["ic_E"], out["ic_omega"] = E, om
    for nm, sp in specs.items():
        out[f"ic_field_{nm}"] = ref_ic.build_initial_field(m2, sp)
        meta[f"ic_{nm}"] = _jsonable(sp)
    especs = {
        "fd": InitialConditionSpec(energy_kind="fermi_dirac", energy_params={"temperature": 0.35}),
        "fd_default_T": InitialConditionSpec(energy_kind="fermi_dirac"),
        "uni": InitialConditionSpec(energy_kind="uniform", energy_params={"value": 2.5}),
        "cust": InitialConditionSpec(energy_kind="custom", energy_custom_body="return np.exp(-(E - gap) / params['w'])", energy_custom_params={"w": 90.0}),
    }
    for nm, sp in especs.items():
        out[f"ic_ew_{nm}"] = ref_ic.build_initial_energy_weights(E, 180.0, 0.1, sp, 0.2)
        meta[f"icw_{nm}"] = _jsonable(sp)
    pspecs = {
        "be": InitialConditionSpec(phonon_energy_kind="bose_einstein", phonon_energy_params={"temperature": 0.4}),
        "be_bath": InitialConditionSpec(),
        "uni": InitialConditionSpec(phonon_energy_kind="uniform", phonon_energy_params={"value": 0.02},
                                    phonon_spatial_kind="gaussian", phonon_spatial_params={"amplitude": 1.5, "x0": 0.5, "y0": 0.5, "sigma": 0.25}),
        "full": InitialConditionSpec(phonon_full_custom_enabled=True,
                                     phonon_full_custom_body="return 0.01 * (1 + x) * np.exp(-E / 300.0) + 0 * y"),
    }
    for nm, sp in pspecs.items():
        out[f"ic_ph_{nm}"] = ref_ic.build_initial_phonon_energy_state(m2, om, sp, 0.15)
        meta[f"icp_{nm}"] = _jsonable(sp)
    qfull = InitialConditionSpec(qp_full_custom_enabled=True,
                                 qp_full_custom_body="return 1e-4 * np.exp(-((x-0.5)**2 + (y-0.5)**2) / 0.05) * np.exp(-E / 500.0)")
    out["ic_qp_full"] = ref_ic.build_initial_qp_energy_state(m2, E, qfull)
    meta["icq_full"] = _jsonable(qfull)
    out["gap_values_expr"] = ref_ic.evaluate_gap_expression("return 180 + 20 * x - 3 * y", m2, 180.0)
    out["gap_values_default"] = ref_ic.evaluate_gap_expression("", m2, 180.0)
    # external generation (solver.py:878-964)
    n = int(m2.sum())
    gens = {
        "const": (ExternalGenerationSpec(mode="constant", rate=2e-8), 0.3),
        "pulse_in": (ExternalGenerationSpec(mode="pulse", pulse_start=0.2, pulse_duration=0.3, pulse_rate=5e-7), 0.2),
        "pulse_end": (ExternalGenerationSpec(mode="pulse", pulse_start=0.2, pulse_duration=0.3, pulse_rate=5e-7), 0.5),
        "pulse_before": (ExternalGenerationSpec(mode="pulse", pulse_start=0.2, pulse_duration=0.3, pulse_rate=5e-7), 0.1),
        "custom_vec": (ExternalGenerationSpec(mode="custom", custom_body="return params['g'] * np.exp(-E / 400.0) * (1 + x) * (t < 1.0)", custom_params={"g": 1e-7}), 0.4),
        "custom_scalar": (ExternalGenerationSpec(mode="custom", custom_body="return 3e-9", custom_params={}), 0.4),
    }
    for nm, (sp, t) in gens.items():
        out[f"gen_{nm}"] = ref_solver.evaluate_external_generation(sp, E, n, t, m2)
        meta[f"gen_{nm}"] = {"spec": _jsonable(sp), "t": t}
    out["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(GOLDEN / "host_side.npz", **out)
    print("  host_side.npz")


def main() -> None:
    os.makedirs(GOLDEN, exist_ok=True)
    print("writing fixtures under", GOLDEN)
    gen_tables()
    gen_collision_vectors()
    gen_geometry()
    gen_operators()
    gen_precompute_ic_generation()
    gen_crosscheck()
    gen_validation()
    gen_legacy_parity_inputs()
    gen_regressions()
    gen_config1()
    gen_test_suite()
    total = sum(p.stat().st_size for p in GOLDEN.rglob("*") if p.is_file())
    print(f"total fixture size: {total/1e6:.2f} MB")


if __name__ == "__main__":
    main()
