"""ctypes binding of ``libqpsim_hip.so`` (C ABI declared in ``include/qpsim_hip.h``).

There is no CPU fallback: if the shared library is missing or no HIP device is visible the
first compute call raises ``RuntimeError`` -- build with ``python __graft_entry__.py`` or
``make -C quasiparticle-physics-simulation_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_NAME = "libqpsim_hip.so"
_LIB_PATH = Path(__file__).resolve().parent / LIB_NAME
_lib = None

c_dp = C.c_void_p  # device pointers travel as raw addresses


class _SizedStructure(C.Structure):
    """Structures whose first member is ``struct_size`` (the library rejects a mismatch with its own sizeof)."""

    @classmethod
    def make(cls, *values):
        """Instance with ``struct_size = sizeof`` and the remaining members from ``values`` in declaration order."""
        return cls(C.sizeof(cls), *values)


class GridDesc(_SizedStructure):
    """Mirror of ``qp_grid_desc``."""
    _fields_ = [("struct_size", C.c_uint32), ("ny", C.c_int32), ("nx", C.c_int32), ("nfield", C.c_int32),
                ("flags", c_dp), ("ex", c_dp), ("ey", c_dp), ("sx", c_dp), ("sy", c_dp),
                ("dcoef", c_dp), ("dfield", c_dp)]


class CollisionTables(_SizedStructure):
    """Mirror of ``qp_collision_tables``."""
    _fields_ = [("struct_size", C.c_uint32), ("ne", C.c_int32), ("nw", C.c_int32), ("nclass", C.c_int32),
                ("kr0", c_dp), ("ks0", c_dp), ("rho", c_dp), ("idx_diff", c_dp), ("idx_sum", c_dp),
                ("sign", c_dp), ("cls", c_dp), ("diag_bin", c_dp), ("anti_bin", c_dp), ("flags", C.c_uint32),
                ("gap_sq", c_dp), ("kr_amp", c_dp), ("ks_amp", c_dp), ("pair_inv", c_dp),
                ("ks0_diag", c_dp), ("kr0_anti2", c_dp)]


class RectPlan(C.Structure):
    """Opaque ``qp_adi_rect_plan``; only ever handled by pointer."""


class TilePlan(C.Structure):
    """Opaque ``qp_adi_tile_plan``; only ever handled by pointer."""


# name -> (restype, argtypes); must list every symbol declared in include/qpsim_hip.h
SIGNATURES = {
    "qp_version": (C.c_int, []),
    "qp_last_error": (C.c_char_p, []),
    "qp_build_info": (C.c_char_p, []),
    "qp_halo_pack": (C.c_int, [c_dp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, c_dp,
                               c_dp]),
    "qp_stencil_combine": (C.c_int, [C.POINTER(GridDesc), C.c_double, c_dp, c_dp, c_dp, C.c_double, C.c_double,
                                     C.c_double, C.c_double, C.c_double, c_dp]),
    "qp_stencil_combine_norm": (C.c_int, [C.POINTER(GridDesc), C.c_double, c_dp, c_dp, c_dp, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_double, c_dp, c_dp, c_dp]),
    "qp_implicit_sweep": (C.c_int, [C.POINTER(GridDesc), C.c_double, C.c_int, c_dp, c_dp, c_dp, c_dp]),
    "qp_collision_step": (C.c_int, [C.POINTER(CollisionTables), c_dp, C.c_int64, c_dp, c_dp, c_dp, c_dp,
                                    C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, c_dp]),
    "qp_collision_guard_workspace_bytes": (C.c_int64, [C.c_int64]),
    "qp_collision_step_guarded": (C.c_int, [C.POINTER(CollisionTables), c_dp, C.c_int64, c_dp, c_dp, c_dp, c_dp,
                                            C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, c_dp, c_dp,
                                            c_dp, c_dp]),
    "qp_euler_collision": (C.c_int, [C.c_int32, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_double, C.c_double,
                                     C.c_int, c_dp]),
    "qp_add_constant": (C.c_int, [c_dp, C.c_int64, C.c_int32, c_dp, C.c_double, c_dp]),
    "qp_add_scaled": (C.c_int, [C.c_int64, c_dp, c_dp, C.c_double, c_dp]),
    "qp_pauli_workspace_bytes": (C.c_int64, []),
    "qp_pauli_stats": (C.c_int, [c_dp, c_dp, c_dp, c_dp, C.c_int32, C.c_int32, C.c_int64, C.c_double, c_dp, c_dp,
                                 c_dp, c_dp]),
    "qp_energy_integrate": (C.c_int, [c_dp, C.c_int32, C.c_int64, C.c_double, c_dp, c_dp]),
    "qp_weighted_sum": (C.c_int, [c_dp, c_dp, C.c_int32, C.c_int64, c_dp, c_dp]),
    "qp_absmax": (C.c_int, [c_dp, C.c_int64, c_dp, c_dp, c_dp]),
    "qp_axpy": (C.c_int, [C.c_int64, C.c_double, c_dp, c_dp, c_dp]),
    "qp_cheb_update": (C.c_int, [C.c_int64, C.c_double, c_dp, C.c_double, c_dp, c_dp, c_dp]),
    "qp_adi_rect_combine": (C.c_int, [C.POINTER(RectPlan), c_dp, c_dp, c_dp, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, c_dp, c_dp, c_dp]),
    "qp_adi_rect_plan_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32,
                                          C.POINTER(C.POINTER(RectPlan))]),
    "qp_adi_rect_plan_decoupled": (C.c_int, [C.POINTER(RectPlan), C.c_int32]),
    "qp_adi_rect_plan_fine": (C.c_int, [C.POINTER(RectPlan)]),
    "qp_adi_rect_plan_create_pr": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.c_double, C.POINTER(RectPlan),
                                             C.POINTER(C.POINTER(RectPlan))]),
    "qp_adi_rect_pr_iteration": (C.c_int, [C.POINTER(RectPlan), c_dp, c_dp, c_dp]),
    "qp_adi_rect_pr_cycle": (C.c_int, [C.POINTER(C.POINTER(RectPlan)), C.c_int32, c_dp, c_dp, c_dp]),
    "qp_adi_rect_plan_destroy": (C.c_int, [C.POINTER(RectPlan)]),
    "qp_adi_rect_steps": (C.c_int, [C.POINTER(RectPlan), c_dp, C.c_int32, c_dp]),
    "qp_adi_rect_solve": (C.c_int, [C.POINTER(RectPlan), c_dp, c_dp]),
    "qp_adi_rect_plan_create_block": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_double),
                                                C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.c_int32,
                                                C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.POINTER(RectPlan))]),
    "qp_adi_rect_phase": (C.c_int, [C.POINTER(RectPlan), C.c_int32, c_dp, c_dp]),
    "qp_adi_rect_iface_halo": (C.c_int, [C.POINTER(RectPlan), C.c_int32, C.c_int32, C.c_int32, c_dp, c_dp]),
    "qp_adi_rect_set_field_halo": (C.c_int, [C.POINTER(RectPlan), C.c_int32, c_dp, c_dp]),
    "qp_nan_pad": (C.c_int, [c_dp, C.c_int64, C.c_int32, c_dp, C.c_double, c_dp, c_dp]),
    "qp_collision_register_kernel_available": (C.c_int, [C.c_int32]),
    "qp_collision_register_kernel_classes": (C.c_int, [C.c_int32]),
    "qp_collision_onepass_available": (C.c_int, [C.c_int32]),
    "qp_collision_pair_available": (C.c_int, [C.c_int32]),
    "qp_collision_double_step_guarded": (C.c_int, [C.POINTER(CollisionTables), c_dp, C.c_int64, c_dp, c_dp, c_dp, C.c_double,
                                                   C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double,
                                                   c_dp, c_dp, c_dp, c_dp]),
    "qp_adi_tile_plan_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_double), c_dp, c_dp,
                                          c_dp, c_dp, c_dp, C.POINTER(C.POINTER(TilePlan))]),
    "qp_adi_tile_plan_create_var": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp,
                                              c_dp, C.POINTER(C.POINTER(TilePlan))]),
    "qp_adi_tile_plan_destroy": (C.c_int, [C.POINTER(TilePlan)]),
    "qp_adi_tile_plan_info": (C.c_int, [C.POINTER(TilePlan), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "qp_adi_tile_steps": (C.c_int, [C.POINTER(TilePlan), c_dp, C.c_int32, c_dp]),
    "qp_adi_tile_solve": (C.c_int, [C.POINTER(TilePlan), c_dp, c_dp]),
}


class HipLibraryMissing(RuntimeError):
    pass


def library_path() -> Path:
    return Path(os.environ.get("QPSIM_HIP_LIBRARY", _LIB_PATH))


def load():
    """Load the shared library and attach signatures (idempotent)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise HipLibraryMissing(
            f"{path} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(or `make -C quasiparticle-physics-simulation_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def build_info() -> dict:
    """``qp_build_info()`` parsed, plus the path of the library that answered."""
    import json
    info = json.loads(load().qp_build_info().decode())
    info["library"] = str(library_path())
    return info


class QPHipError(RuntimeError):
    status = 0


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().qp_last_error()
        err = QPHipError(f"{what or 'libqpsim_hip call'} failed with status {rc}: "
                         f"{msg.decode(errors='replace') if msg else ''}")
        err.status = rc
        raise err
