// Shared host/device helpers of libqpsim_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "qpsim_hip.h"

namespace qp {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return QP_ERR_LAUNCH;
  }
  return QP_OK;
}

#define QP_REQUIRE(cond, msg)                      \
  do {                                             \
    if (!(cond)) {                                 \
      qp::set_error("%s: %s", __func__, msg);      \
      return QP_ERR_INVALID_ARGUMENT;              \
    }                                              \
  } while (0)

// Geometry description handed to kernels by value.
struct GridView {
  int ny, nx;
  int nfield;
  const uint8_t* flags;
  const double* ex;
  const double* ey;
  const double* sx;
  const double* sy;
  const double* dcoef;
  const double* dfield;
};

inline GridView make_view(const qp_grid_desc* g) {
  GridView v;
  v.ny = g->ny; v.nx = g->nx; v.nfield = g->nfield;
  v.flags = g->flags; v.ex = g->ex; v.ey = g->ey; v.sx = g->sx; v.sy = g->sy;
  v.dcoef = g->dcoef; v.dfield = g->dfield;
  return v;
}

int validate_grid(const qp_grid_desc* g, const char* who);

// Partial result of the Pauli-guard reduction (qp_misc.hip): largest occupation with its linear index ie*ncell + p
// (np.argmax order: first maximum in C order, a NaN counts as the maximum) and the first forbidden linear index or -1.
struct PauliPartial {
  double maxf;
  long maxidx;
  long forb;
};
constexpr int kGuardMergeBlocks = 512;
// reduces `nparts` partials (written by the collision kernels, one per wave) to out_vals[0], out_idx[0..1];
// `scratch` holds kGuardMergeBlocks partials
void pauli_finish(const PauliPartial* parts, long nparts, PauliPartial* scratch, double* out_vals, long* out_idx,
                  hipStream_t stream);

// out[0] = max over `nparts` per-block maxima (second stage of qp_absmax, shared with qp_adi_rect_combine)
void absmax_finish(const double* parts, int nparts, double* out, hipStream_t stream);

// Harmonic-mean face diffusivity (solver.py:283).
__device__ __forceinline__ double face_mean(double dp, double dq) {
  return 2.0 * dp * dq / fmax(dp + dq, 1e-30);
}

// Diffusivity of field `b` at cell `p`.
__device__ __forceinline__ double cell_d(const GridView& g, int b, long p, long ncell) {
  return g.dfield ? g.dfield[(long)b * ncell + p] : g.dcoef[b];
}

// Coupling weight across the face between p and its neighbour q (both interior).
__device__ __forceinline__ double face_d(const GridView& g, int b, long p, long q, long ncell, double dp) {
  return g.dfield ? face_mean(dp, g.dfield[(long)b * ncell + q]) : dp;
}

}  // namespace qp
