"""Problem-definition types of the hot path (drop-in for ``qpsim.models``).

Pure data: field names, defaults and validation rules follow the reference's
``qpsim/models.py:8-198`` so the same keyword arguments, ``edge_conditions`` dicts and
``SimulationParameters(...)`` calls work unchanged.  The storage/GUI result containers of
the reference (``SetupData``, ``SimulationResultData``, ``Test*Data``; models.py:201-267)
belong to subsystems that are out of scope here and are not mirrored.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from datetime import datetime, timezone
from typing import Any

# registries = the "plugin surface" (models.py:8-16)
BOUNDARY_KINDS = {"reflective", "neumann", "dirichlet", "absorbing", "robin"}
COLLISION_SOLVERS = {"fischer_catelani_local"}
EXTERNAL_GENERATION_MODES = {"none", "constant", "pulse", "custom"}

_VALUE_KINDS = ("neumann", "dirichlet", "robin")

_SPATIAL_BODY = "return np.exp(-((x-0.5)**2 + (y-0.5)**2) / 0.02)"
_FULL_BODY = "return np.exp(-((x-0.5)**2 + (y-0.5)**2) / 0.02) * np.exp(-E / 500.0)"
_ONES_BODY = "return np.ones_like(E)"


def utc_now_iso() -> str:
    return datetime.now(timezone.utc).isoformat()


def normalize_collision_solver_name(value: str) -> str:
    """Lower-cased solver key; anything not registered is an error (models.py:23-30)."""
    key = str(value).strip().lower()
    if key in COLLISION_SOLVERS:
        return key
    raise ValueError(
        f"Unsupported collision solver '{value}'. Supported values: {', '.join(sorted(COLLISION_SOLVERS))}."
    )


@dataclass
class BoundaryCondition:
    """One boundary condition; ``value``/``aux_value`` meaning depends on ``kind`` (models.py:33-49).

    dirichlet: value = g;  neumann: value = outward flux q;  robin: value = beta, aux_value = gamma.
    """
    kind: str
    value: float | None = None
    aux_value: float | None = None

    def normalized_kind(self) -> str:
        return self.kind.strip().lower()

    def validate(self) -> None:
        k = self.normalized_kind()
        if k not in BOUNDARY_KINDS:
            raise ValueError(f"Unsupported boundary condition kind: {self.kind}")
        if k in _VALUE_KINDS and self.value is None:
            raise ValueError(f"Boundary condition '{k}' requires a numeric value")


@dataclass
class BoundaryFace:
    row: int
    col: int
    direction: str  # "up" | "down" | "left" | "right" (outward normal of the face)


@dataclass
class EdgeSegment:
    edge_id: str
    x0: float
    y0: float
    x1: float
    y1: float
    normal: str
    faces: list[BoundaryFace]


@dataclass
class GeometryData:
    name: str
    source_path: str
    layer: int
    mesh_size: float
    mask: list[list[int]]
    edges: list[EdgeSegment]
    bounds: list[float] | None = None


@dataclass
class InitialConditionSpec:
    """Split spatial x energy initial condition for quasiparticles and phonons (models.py:81-108)."""
    spatial_kind: str = ""
    spatial_params: dict[str, Any] = field(default_factory=dict)
    spatial_custom_body: str = _SPATIAL_BODY
    spatial_custom_params: dict[str, Any] = field(default_factory=dict)
    energy_kind: str = ""
    energy_params: dict[str, Any] = field(default_factory=dict)
    energy_custom_body: str = _ONES_BODY
    energy_custom_params: dict[str, Any] = field(default_factory=dict)
    qp_full_custom_enabled: bool = False
    qp_full_custom_body: str = _FULL_BODY
    qp_full_custom_params: dict[str, Any] = field(default_factory=dict)
    phonon_spatial_kind: str = ""
    phonon_spatial_params: dict[str, Any] = field(default_factory=dict)
    phonon_spatial_custom_body: str = "return 1.0"
    phonon_spatial_custom_params: dict[str, Any] = field(default_factory=dict)
    phonon_energy_kind: str = ""
    phonon_energy_params: dict[str, Any] = field(default_factory=dict)
    phonon_energy_custom_body: str = _ONES_BODY
    phonon_energy_custom_params: dict[str, Any] = field(default_factory=dict)
    phonon_full_custom_enabled: bool = False
    phonon_full_custom_body: str = _FULL_BODY
    phonon_full_custom_params: dict[str, Any] = field(default_factory=dict)


@dataclass
class ExternalGenerationSpec:
    """External generation g_ext(E, x, t) (models.py:111-136)."""
    mode: str = "none"
    rate: float = 0.0
    pulse_start: float = 0.0
    pulse_duration: float = 10.0
    pulse_rate: float = 0.0
    custom_body: str = "return 0.0"
    custom_params: dict[str, Any] = field(default_factory=dict)

    def normalized_mode(self) -> str:
        return self.mode.strip().lower()

    def validate(self) -> None:
        if self.normalized_mode() not in EXTERNAL_GENERATION_MODES:
            raise ValueError(
                f"Unsupported external generation mode '{self.mode}'. "
                f"Supported: {', '.join(sorted(EXTERNAL_GENERATION_MODES))}."
            )
        for label, val in (("constant rate", self.rate), ("pulse rate", self.pulse_rate),
                           ("pulse_duration", self.pulse_duration)):
            if val < 0:
                raise ValueError(f"External generation {label} must be non-negative.")


@dataclass
class SimulationParameters:
    """Run parameters with the reference's tau aliasing and range checks (models.py:139-198)."""
    diffusion_coefficient: float
    dt: float
    total_time: float
    mesh_size: float
    store_every: int = 1
    energy_gap: float = 0.0
    energy_min_factor: float = 1.0
    energy_max_factor: float = 10.0
    num_energy_bins: int = 50
    dynes_gamma: float = 0.0
    gap_expression: str = ""
    collision_solver: str = "fischer_catelani_local"
    enable_diffusion: bool = True
    enable_recombination: bool = False
    enable_scattering: bool = False
    tau_0: float = 440.0
    tau_s: float | None = None
    tau_r: float | None = None
    T_c: float = 1.2
    bath_temperature: float = 0.1
    export_phonon_history: bool = False
    external_generation: ExternalGenerationSpec = field(default_factory=ExternalGenerationSpec)

    def __post_init__(self) -> None:
        self.collision_solver = normalize_collision_solver_name(self.collision_solver)
        # tau_0 is a convenience default for tau_s / tau_r and is then re-derived from them
        self.tau_s = float(self.tau_0) if self.tau_s is None else self.tau_s
        self.tau_r = float(self.tau_0) if self.tau_r is None else self.tau_r
        self.tau_0 = float(0.5 * (self.tau_s + self.tau_r))
        for name, val in (("dt", self.dt), ("total_time", self.total_time), ("mesh_size", self.mesh_size)):
            if val <= 0:
                raise ValueError(f"{name} must be positive.")
        if self.bath_temperature < 0:
            raise ValueError("bath_temperature must be non-negative.")
        if self.enable_recombination or self.enable_scattering:
            for name, val in (("T_c", self.T_c), ("tau_s", self.tau_s), ("tau_r", self.tau_r)):
                if val <= 0:
                    raise ValueError(f"{name} must be positive when recombination or scattering is enabled.")
        if self.energy_gap > 0:
            if self.energy_min_factor < 1.0:
                raise ValueError("energy_min_factor must be >= 1.0 when energy_gap > 0.")
            if self.energy_max_factor <= self.energy_min_factor:
                raise ValueError("energy_max_factor must be > energy_min_factor when energy_gap > 0.")
            if self.num_energy_bins < 2:
                raise ValueError("num_energy_bins must be >= 2 when energy_gap > 0.")
        self.external_generation.validate()
