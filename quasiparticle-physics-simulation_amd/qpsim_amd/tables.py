"""Host-side coefficient tables for the collision / diffusion kernels.

Everything here is O(NE^2) or smaller and is built once per run on the host with NumPy, then
uploaded; the device kernels index these tables verbatim.  Keeping the table construction in
NumPy guarantees the same bits as the reference for the pieces that are sensitive to
evaluation order (``np.unique(np.round(., 12))`` defines the phonon bins, the 1e-30 floors,
the 500-exponent clamps).  Formulas follow ``qpsim/solver.py`` (lines cited per function).
"""
from __future__ import annotations

import numpy as np

# Boltzmann constant in micro-eV per kelvin (solver.py:347)
KB_UEV_PER_K = 86.17333262145


def build_energy_grid(gap: float, energy_min_factor: float, energy_max_factor: float,
                      num_energy_bins: int) -> tuple[np.ndarray, float]:
    """Bin centres E_i = E_min + (i + 1/2) dE and the bin width (solver.py:61-84).

    A single bin sits at the midpoint and integrates with unit weight.
    """
    if gap <= 0:
        raise ValueError("gap must be positive.")
    if num_energy_bins <= 0:
        raise ValueError("num_energy_bins must be >= 1.")
    e_lo = energy_min_factor * gap
    e_hi = energy_max_factor * gap
    if num_energy_bins == 1:
        return np.array([0.5 * (e_lo + e_hi)], dtype=float), 1.0
    if e_hi <= e_lo:
        raise ValueError("energy_max_factor must be > energy_min_factor for num_energy_bins > 1.")
    width = (e_hi - e_lo) / float(num_energy_bins)
    return e_lo + (np.arange(num_energy_bins, dtype=float) + 0.5) * width, width


def integration_widths_from_centers(centers: np.ndarray, *, fallback_width: float = 1.0) -> np.ndarray:
    """Widths of the cells whose faces are the midpoints between centres (solver.py:87-109)."""
    c = np.asarray(centers, dtype=float).reshape(-1)
    if c.size == 0:
        raise ValueError("centers must be non-empty.")
    if c.size == 1:
        return np.array([float(fallback_width)], dtype=float)
    if np.any(~np.isfinite(c)):
        raise ValueError("centers must contain finite values.")
    if np.any(np.diff(c) <= 0):
        raise ValueError("centers must be strictly increasing.")
    faces = np.empty(c.size + 1, dtype=float)
    faces[1:-1] = 0.5 * (c[:-1] + c[1:])
    faces[0] = c[0] - 0.5 * (c[1] - c[0])
    faces[-1] = c[-1] + 0.5 * (c[-1] - c[-2])
    widths = np.diff(faces)
    if np.any(widths <= 0):
        raise ValueError("Derived non-positive integration width from centers.")
    return widths


def bcs_density_of_states(E: np.ndarray, gap: float) -> np.ndarray:
    """rho = E / sqrt(E^2 - gap^2) above the gap, 0 at and below it (solver.py:324-329)."""
    E = np.asarray(E, dtype=float)
    rho = np.zeros_like(E)
    above = E > gap
    rho[above] = E[above] / np.sqrt(E[above] ** 2 - gap ** 2)
    return rho


def dynes_density_of_states(E: np.ndarray, gap: float, gamma: float) -> np.ndarray:
    """Re{(E - i G)/sqrt((E - i G)^2 - gap^2)} clipped at 0; BCS when G <= 0 (solver.py:332-342)."""
    if gamma <= 0:
        return bcs_density_of_states(E, gap)
    z = np.asarray(E, dtype=float) - 1j * gamma
    with np.errstate(invalid="ignore"):
        val = np.real(z / np.sqrt(z ** 2 - gap ** 2))
    return np.maximum(val, 0.0)


def thermal_phonon_occupation(omega_bins: np.ndarray, temperature: float) -> np.ndarray:
    """Bose-Einstein n(omega, T); exponent clamped at 500, non-finite -> 0 (solver.py:350-370)."""
    omega = np.asarray(omega_bins, dtype=float)
    if omega.ndim != 1:
        raise ValueError("omega_bins must be a 1D array.")
    if np.any(~np.isfinite(omega)):
        raise ValueError("omega_bins must contain only finite values.")
    if np.any(omega < 0):
        raise ValueError("omega_bins must be non-negative.")
    if temperature <= 0:
        return np.zeros_like(omega)
    kT = KB_UEV_PER_K * float(temperature)
    arg = np.minimum(omega / max(kT, 1e-30), 500.0)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        occ = 1.0 / (np.exp(arg) - 1.0)
    occ[~np.isfinite(occ)] = 0.0
    return np.maximum(occ, 0.0)


def thermal_qp_weights(E_bins: np.ndarray, gap: float, temperature: float,
                       dynes_gamma: float = 0.0) -> np.ndarray:
    """Un-normalised equilibrium spectrum rho(E) f_FD(E, T), zero chemical potential (solver.py:429-460)."""
    rho = dynes_density_of_states(E_bins, gap, dynes_gamma)
    if temperature <= 0:
        return np.zeros_like(rho)
    kT = KB_UEV_PER_K * temperature
    return rho * (1.0 / (np.exp(np.minimum(np.asarray(E_bins, dtype=float) / kT, 500.0)) + 1.0))


def recombination_kernel_base(E_bins: np.ndarray, gap: float, tau_0: float, T_c: float) -> np.ndarray:
    """K^r_0(Ei, Ej) = (1/tau) ((Ei+Ej)/kTc)^2 / kTc (1 + gap^2/(Ei Ej)) (solver.py:463-474)."""
    E = np.asarray(E_bins, dtype=float)
    kTc = KB_UEV_PER_K * T_c
    pair_sum = E[:, None] + E[None, :]
    pair_prod = E[:, None] * E[None, :]
    return (1.0 / tau_0) * (pair_sum / kTc) ** 2 / kTc * (1.0 + gap ** 2 / np.maximum(pair_prod, 1e-30))


def scattering_kernel_base(E_bins: np.ndarray, gap: float, tau_0: float, T_c: float) -> np.ndarray:
    """K^s_0(Ei, Ej) = (1/tau) (Ei-Ej)^2 / kTc^3 max(1 - gap^2/(Ei Ej), 0), zero diagonal (solver.py:477-490)."""
    E = np.asarray(E_bins, dtype=float)
    kTc = KB_UEV_PER_K * T_c
    pair_diff = E[:, None] - E[None, :]
    pair_prod = E[:, None] * E[None, :]
    coherence = np.maximum(1.0 - gap ** 2 / np.maximum(pair_prod, 1e-30), 0.0)
    K = (1.0 / tau_0) * pair_diff ** 2 / kTc ** 3 * coherence
    np.fill_diagonal(K, 0.0)
    return K


def recombination_kernel(E_bins: np.ndarray, gap: float, tau_0: float, T_c: float,
                         bath_temperature: float) -> np.ndarray:
    """K^r with a fixed-temperature phonon bath: K^r_0 (1 + n_BE(Ei+Ej)) (solver.py:493-516)."""
    E = np.asarray(E_bins, dtype=float)
    kTp = KB_UEV_PER_K * bath_temperature
    pair_sum = E[:, None] + E[None, :]
    if kTp > 0:
        stim = 1.0 / (np.exp(np.minimum(pair_sum / kTp, 500.0)) - 1.0) + 1.0
    else:
        stim = np.ones_like(pair_sum, dtype=float)
    return recombination_kernel_base(E, gap, tau_0, T_c) * stim


def scattering_kernel(E_bins: np.ndarray, gap: float, tau_0: float, T_c: float,
                      bath_temperature: float) -> np.ndarray:
    """K^s with a fixed bath: emission (Ei > Ej) 1 + n_BE, absorption n_BE, diagonal 0 (solver.py:519-548)."""
    E = np.asarray(E_bins, dtype=float)
    kTp = KB_UEV_PER_K * bath_temperature
    pair_diff = E[:, None] - E[None, :]
    if kTp > 0:
        with np.errstate(divide="ignore", invalid="ignore"):
            n_be = 1.0 / (np.exp(np.minimum(np.abs(pair_diff) / kTp, 500.0)) - 1.0)
        occ = np.where(pair_diff > 0, 1.0 + n_be, n_be)
    else:
        occ = np.where(pair_diff > 0, 1.0, 0.0)
    np.fill_diagonal(occ, 0.0)
    return scattering_kernel_base(E, gap, tau_0, T_c) * occ


def build_phonon_frequency_map(E_bins: np.ndarray):
    """Phonon bins and pair->bin maps of the local coupled update (solver.py:668-683).

    omega = sorted unique of round(|Ei-Ej| U (Ei+Ej), 12); returns (omega[NW], idx_diff[NE,NE],
    idx_sum[NE,NE], sign[NE,NE] int8 = sign(Ei - Ej)).
    """
    E = np.asarray(E_bins, dtype=float)
    if E.ndim != 1:
        raise ValueError("E_bins must be a 1D array.")
    absdiff = np.abs(E[:, None] - E[None, :])
    pair_sum = E[:, None] + E[None, :]
    omega, inverse = np.unique(np.round(np.concatenate([absdiff.ravel(), pair_sum.ravel()]), 12),
                               return_inverse=True)
    npairs = E.size * E.size
    shape = (E.size, E.size)
    return (omega, inverse[:npairs].reshape(shape), inverse[npairs:].reshape(shape),
            np.sign(E[:, None] - E[None, :]).astype(np.int8))


def diffusion_coefficients(E_bins: np.ndarray, gap: float, D0: float) -> np.ndarray:
    """D(E) = D0 sqrt(max(0, 1 - (gap/E)^2)) per energy bin (solver.py:1135)."""
    E = np.asarray(E_bins, dtype=float)
    return D0 * np.sqrt(np.maximum(0.0, 1.0 - (gap / E) ** 2))
