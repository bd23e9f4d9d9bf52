// General diffusion kernels: any mask, any per-face boundary condition, uniform or spatially varying D.
// Correctness-first path (one thread per grid line for the implicit sweep); the full-rectangle uniform-D
// fast path lives in qp_adi_rect.hip.
#include <stdarg.h>

#include "qp_common.h"

namespace qp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int validate_grid(const qp_grid_desc* g, const char* who) {
  if (!g) { set_error("%s: grid descriptor is NULL", who); return QP_ERR_INVALID_ARGUMENT; }
  if (g->struct_size != sizeof(qp_grid_desc)) {
    set_error("%s: qp_grid_desc.struct_size is %u, this library expects %zu (binding built against another header revision)",
              who, g->struct_size, sizeof(qp_grid_desc));
    return QP_ERR_INVALID_ARGUMENT;
  }
  if (g->ny <= 0 || g->nx <= 0 || g->nfield <= 0) {
    set_error("%s: ny, nx, nfield must be positive (got %d, %d, %d)", who, g->ny, g->nx, g->nfield);
    return QP_ERR_INVALID_ARGUMENT;
  }
  if (!g->flags || !g->ex || !g->ey || !g->sx || !g->sy) {
    set_error("%s: flags/ex/ey/sx/sy must be non-NULL", who);
    return QP_ERR_INVALID_ARGUMENT;
  }
  if ((g->dcoef == nullptr) == (g->dfield == nullptr)) {
    set_error("%s: exactly one of dcoef / dfield must be given", who);
    return QP_ERR_INVALID_ARGUMENT;
  }
  return QP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// out = c0 u + cx r Lx u + cy r Ly u + cs r D (sx + sy) + cr rin
// ---------------------------------------------------------------------------------------------------------
// A block walks (field, row) pairs, its threads the cells of the row (no 64-bit division per cell).  Boundary terms are
// read only where a link is missing - they belong to boundary faces, and a cell with all four links has none - which
// saves 32 of the 57 B per cell this kernel moved on masks that are mostly interior.  `part` non-NULL: every block leaves
// max |out| of its cells there (NaN propagates as +inf).
__global__ void __launch_bounds__(256) stencil_combine_kernel(GridView g, double r, const double* __restrict__ u,
                                                              const double* __restrict__ rin,
                                                              double* __restrict__ out, double c0, double cx,
                                                              double cy, double cs, double cr, double* __restrict__ part) {
  const long ncell = (long)g.ny * g.nx;
  double m = 0.0;
  for (int line = blockIdx.x; line < g.nfield * g.ny; line += gridDim.x) {
    const int b = line / g.ny, j = line - b * g.ny;
    const double* ub = u + (long)b * ncell;
    for (int i = threadIdx.x; i < g.nx; i += blockDim.x) {
      const long p = (long)j * g.nx + i;
      const long t = (long)b * ncell + p;
      const unsigned f = g.flags[p];
      double res = 0.0;
      if (f & QP_FLAG_ACTIVE) {
        const double up = ub[p];
        const double dp = cell_d(g, b, p, ncell);
        double lx = 0.0, ly = 0.0, src = 0.0;
        if ((f & 15u) != 15u) {
          lx = -g.ex[p] * dp * up;
          ly = -g.ey[p] * dp * up;
          src = g.sx[p] + g.sy[p];
        }
        if (f & QP_FLAG_LINK_XM) lx += face_d(g, b, p, p - 1, ncell, dp) * (ub[p - 1] - up);
        if (f & QP_FLAG_LINK_XP) lx += face_d(g, b, p, p + 1, ncell, dp) * (ub[p + 1] - up);
        if (f & QP_FLAG_LINK_YM) ly += face_d(g, b, p, p - g.nx, ncell, dp) * (ub[p - g.nx] - up);
        if (f & QP_FLAG_LINK_YP) ly += face_d(g, b, p, p + g.nx, ncell, dp) * (ub[p + g.nx] - up);
        res = c0 * up + cx * (r * lx) + cy * (r * ly) + cs * (r * dp * src);
        if (rin) res += cr * rin[t];
      }
      out[t] = res;
      const double v = fabs(res);
      m = (v != v) ? __builtin_huge_val() : fmax(m, v);
    }
  }
  if (part) {
    __shared__ double sm[256];
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
      __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
  }
}

// ---------------------------------------------------------------------------------------------------------
// (I - r L_dir) x = rhs, one thread per (field, line); forward pass stores c', d' in scratch.
// ---------------------------------------------------------------------------------------------------------
template <int DIR>
__global__ void __launch_bounds__(128) thomas_lines_kernel(GridView g, double r, const double* __restrict__ rhs,
                                                           double* __restrict__ x, double* __restrict__ cp,
                                                           double* __restrict__ dp_) {
  const long ncell = (long)g.ny * g.nx;
  const int nlines = DIR == 0 ? g.ny : g.nx;
  const int len = DIR == 0 ? g.nx : g.ny;
  const long stride = DIR == 0 ? 1 : g.nx;
  const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= (long)nlines * g.nfield) return;
  const int b = (int)(line / nlines);
  const int l = (int)(line - (long)b * nlines);
  const long base = DIR == 0 ? (long)l * g.nx : (long)l;
  const long fo = (long)b * ncell;
  const unsigned LM = DIR == 0 ? QP_FLAG_LINK_XM : QP_FLAG_LINK_YM;
  const unsigned LP = DIR == 0 ? QP_FLAG_LINK_XP : QP_FLAG_LINK_YP;
  const double* e = DIR == 0 ? g.ex : g.ey;

  double cprev = 0.0, dprev = 0.0;
  for (int k = 0; k < len; ++k) {
    const long p = base + (long)k * stride;
    const unsigned f = g.flags[p];
    double ck = 0.0, dk = 0.0;
    if (f & QP_FLAG_ACTIVE) {
      const double dcell = cell_d(g, b, p, ncell);
      const double wm = (f & LM) ? r * face_d(g, b, p, p - stride, ncell, dcell) : 0.0;
      const double wp = (f & LP) ? r * face_d(g, b, p, p + stride, ncell, dcell) : 0.0;
      const double diag = 1.0 + wm + wp + r * e[p] * dcell;
      const double inv = 1.0 / (diag + wm * cprev);  // a = -wm
      ck = -wp * inv;
      dk = (rhs[fo + p] + wm * dprev) * inv;
    }
    cp[fo + p] = ck;
    dp_[fo + p] = dk;
    cprev = ck;
    dprev = dk;
  }
  double xn = 0.0;
  for (int k = len - 1; k >= 0; --k) {
    const long p = base + (long)k * stride;
    const double xv = dp_[fo + p] - cp[fo + p] * xn;
    x[fo + p] = xv;
    xn = xv;
  }
}

}  // namespace qp

extern "C" {

int qp_version(void) { return 200; }

const char* qp_last_error(void) { return qp::g_err; }

int qp_stencil_combine_norm(const qp_grid_desc* g, double r, const double* u, const double* rin, double* out, double c0,
                            double cx, double cy, double cs, double cr, void* workspace, double* norm_out, void* stream) {
  int rc = qp::validate_grid(g, "qp_stencil_combine");
  if (rc) return rc;
  QP_REQUIRE(u && out, "u and out must be non-NULL");
  QP_REQUIRE(rin || cr == 0.0, "rin is NULL but cr != 0");
  QP_REQUIRE(u != out, "u and out must not alias (neighbour reads)");
  QP_REQUIRE(norm_out == nullptr || workspace, "workspace is required with norm_out");
  long blocks = (long)g->ny * g->nfield;
  const long cap = norm_out ? 1024 : 8192;           // with a norm: the partial slots of the reduction workspace
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(qp::stencil_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     qp::make_view(g), r, u, cr == 0.0 ? nullptr : rin, out, c0, cx, cy, cs, cr,
                     norm_out ? (double*)workspace : nullptr);
  if (norm_out) qp::absmax_finish((const double*)workspace, (int)blocks, norm_out, (hipStream_t)stream);
  return qp::check_launch("qp_stencil_combine");
}

int qp_stencil_combine(const qp_grid_desc* g, double r, const double* u, const double* rin, double* out, double c0,
                       double cx, double cy, double cs, double cr, void* stream) {
  return qp_stencil_combine_norm(g, r, u, rin, out, c0, cx, cy, cs, cr, nullptr, nullptr, stream);
}

int qp_implicit_sweep(const qp_grid_desc* g, double r, int dir, const double* rhs, double* x, double* scratch,
                      void* stream) {
  int rc = qp::validate_grid(g, "qp_implicit_sweep");
  if (rc) return rc;
  QP_REQUIRE(rhs && x && scratch, "rhs, x and scratch must be non-NULL");
  QP_REQUIRE(dir == 0 || dir == 1, "dir must be 0 (x) or 1 (y)");
  const long ncell = (long)g->ny * g->nx;
  const long nlines = (long)(dir == 0 ? g->ny : g->nx) * g->nfield;
  double* cp = scratch;
  double* dp = scratch + ncell * g->nfield;
  const unsigned blocks = (unsigned)((nlines + 127) / 128);
  if (dir == 0)
    hipLaunchKernelGGL(qp::thomas_lines_kernel<0>, dim3(blocks), dim3(128), 0, (hipStream_t)stream,
                       qp::make_view(g), r, rhs, x, cp, dp);
  else
    hipLaunchKernelGGL(qp::thomas_lines_kernel<1>, dim3(blocks), dim3(128), 0, (hipStream_t)stream,
                       qp::make_view(g), r, rhs, x, cp, dp);
  return qp::check_launch("qp_implicit_sweep");
}

}  // extern "C"
