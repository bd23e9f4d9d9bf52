// Fast CN-ADI path for full rectangles with one diffusivity per field and one boundary condition per side.
//
// Peaceman-Rachford step, a = r D (r = dt / (2 dx^2)):
//     (I - a Lx) u*  = (I + a Ly) u  + a S          S = boundary-face sources (1/dx^2 units)
//     (I - a Ly) u'  = (I + a Lx) u* + a S
// Every grid line is cut into chunks of 64 cells (the edge of a 64 x 64 tile).  A line system A x = d is solved
// by the partition (SPIKE) method:
//     local   y_p = A_p^-1 d_p                       (per chunk, no communication)
//     reduced F_p, E_p = first / last unknown of chunk p from a banded 2P x 2P system whose right-hand side is
//             (y_p[0], y_p[last]) = (g_p . d_p, h_p . d_p),  g_p, h_p = first / last column of A_p^-1
//     final   A_p x_p = d_p + a E_{p-1} e_first + a F_{p+1} e_last       (per chunk again)
// A_p, g_p, h_p and the LU factors of the reduced matrix depend only on (field, direction, chunk position), so
// the host tabulates them once per plan and the kernels read them as wave-uniform scalars.
//
// One wave owns one 64 x 64 tile.  Two tile kernels alternate, each reading and writing every cell once:
//   x-kernel:  load d (rhs of the x-solve) -> finish the x-solve with the ghost values E/F -> u* ->
//              rhs2 = (I + a Lx) u* + a S -> store -> dot products g_y, h_y per column -> reduced rhs for y
//   y-kernel:  load rhs2 -> finish the y-solve -> u' -> [store u' | rhs1' = (I + a Ly) u' + a S -> store ->
//              dot products g_x, h_x per row -> reduced rhs for the NEXT step's x-solve]
// plus a one-thread-per-line kernel for the banded reduced systems.  Consecutive diffusion steps therefore cost
// 2 x (8 B read + 8 B write) = 32 B per cell-update; a step that starts from a materialised field u pays one
// extra pass (the "entry" variant of the y-kernel forms rhs1 from u and its two halo rows).
//
// Global accesses: lane <-> column, one 512 B row segment per wave instruction.  x-direction work needs
// lane <-> row, so the tile is transposed (v_permlane32_swap + 32 x 33 LDS blocks, transpose64) at each change of
// direction.  y-direction work happens entirely in registers (64 doubles per lane).
//
// Where the carried planes stay cached (<= 192 MiB) and r D <~ 0.32 the same passes run on FINE tiles - chunks of 32
// cells, 32 x 64 / 64 x 32 tiles, XCD-consistent block mapping (qp_adi_fine.inc); and the plans of the Peaceman-Rachford
// cycle for the unsplit Crank-Nicolson system (qp_adi_rect_plan_create_pr, qp_adi_rect_pr_cycle) reuse both families with
// a source-plane argument.
#include <algorithm>
#include <cmath>
#include <vector>

#include "qp_tile_common.h"

namespace qp {

struct RectDims {
  int ny, nx, nfield;           // LOCAL block extent
  int py, px;                   // local chunks per column / per row
  int gny, gnx;                 // global grid extent (== ny, nx without decomposition)
  int j0, i0;                   // global offset of the local block (multiples of 64)
  int gpy, gpx;                 // global chunks per column / per row
  int stream;                   // non-temporal plane accesses (working set beyond the Infinity Cache)
};

struct RectView {
  RectDims d;
  const double* alpha;          // [nfield]
  const double* tab;            // [2 dirs][nfield][4 variants][T_NSLOT][TS]
  int compact;                  // every table of the plan has the compact form (table_is_compact): kernels <..., true>
  const double* ctab;           // [2 dirs][nfield][4 variants][2 parts][CT_PART] compact tables (valid when `compact`)
  const double* lu[2];          // per dir: [nfield][5][2P]  (l1, l2, uinv, u1, u2)
  // reduced right-hand sides, per dir [nfield][2P+2][nlines]: row 0 = y[last] of the chunk before the local block,
  // rows 1..2P = (y_0[0], y_0[last], y_1[0], ...), row 2P+1 = y[0] of the chunk after the block.  The two halo rows
  // are filled by the neighbouring rank (domain decomposition) and unused otherwise.
  double* iface[2];
  double* z[2];                 // per dir: [nfield][2P][nlines]   reduced solutions (F_0, E_0, F_1, E_1, ...)
  // per dir [nfield][P+1][3]: (s, t, 1/(1 - s t)) of the interface between local chunks q-1 and q (q = 0 and q = P are
  // the interfaces to the neighbouring blocks)
  const double* icoef[2];
  const double* uhalo[2];       // [nfield][nx]: field row just above / below the local block (entry pass, decomposed)
  int decoupled[2];             // per dir: far couplings underflow -> every interface is an independent 2 x 2 system
  double other_src[2][2];       // [dir][lo/hi]: a-less source of the faces normal to `dir` (x: sx_lo, sx_hi)
  // Peaceman-Rachford iteration passes (<..., SRC = true>): bscale * bsrc is added to every right-hand side formed
  const double* bsrc;           // [nfield][ny*nx]
  double bscale;
};

// coefficient source of one kernel phase: the full table of (dir, field, variant) in registers, or a part of its compact form
template <bool COMPACT, int FIRST, int COUNT>
__device__ __forceinline__ auto coef_source(const RectView& v, int dir, int b, int variant, int lane) {
  const long t = ((long)dir * v.d.nfield + b) * 4 + variant;
  if constexpr (COMPACT) {
    return CoefCompact{as_const(v.ctab + (t * 2 + (FIRST >= T_EW ? 1 : 0)) * CT_PART)};
  } else {
    CoefFull c;
    load_tab<FIRST, COUNT>(v.tab + t * T_NSLOT * TS, lane, c.r);
    return c;
  }
}

__device__ __forceinline__ int chunk_variant(int p, int P) {
  // 0 interior, 1 first, 2 last, 3 single
  return (p == 0 ? 1 : 0) | (p == P - 1 ? 2 : 0);
}

__device__ __forceinline__ const double* table_ptr(const RectView& v, int dir, int b, int variant) {
  return v.tab + ((((long)dir * v.d.nfield + b) * 4 + variant) * T_NSLOT) * TS;
}


// STREAM is a template parameter of the kernels: the headline configuration (cached accesses) keeps exactly the code it
// had before the non-temporal variants existed (a run-time branch cost ~3 % there)
template <int STREAM>
__device__ __forceinline__ TileCoord tile_coord(const RectDims& d) {
  TileCoord t;
  int id = blockIdx.x;
  t.tx = id % d.px;
  id /= d.px;
  t.ty = id % d.py;
  t.b = id / d.py;
  t.j0 = t.ty * TS;
  t.i0 = t.tx * TS;
  t.nr = min(TS, d.ny - t.j0);
  t.nc = min(TS, d.nx - t.i0);
  t.stream = STREAM;
  return t;
}

// Values of the solved line just outside chunk p: gl = E_{p-1} (last unknown of the previous chunk),
// gr = F_{p+1} (first unknown of the next chunk).
//
// The reduced system couples F_p, E_p to E_{p-1} and F_{p+1} with weights a g_p[0], a h_p[last] (O(1)) and
// a g_p[last], a h_p[0] (the far corners of A_p^-1, which decay like rho^63 along the chunk).  When the plan found the
// far weights below 1e-22 for every chunk (`decoupled`), dropping them perturbs the solution far below fp64 rounding and
// each interface (E_p, F_{p+1}) becomes its own 2 x 2 system
//      E - s F = y_p[last],   F - t E = y_{p+1}[0],     s = a h_p[last], t = a g_{p+1}[0]
// which is solved here from the four reduced right-hand sides next to the chunk; otherwise the banded solve of
// rect_reduced_kernel has produced z.
//
// Two halves: `ghost_prefetch` issues the loads (unconditional, always in bounds: rows 0 and 2P+1 of `iface` are the halo
// slots) at the very start of a kernel, ahead of the tile rows; `ghost_finish` does the arithmetic where the values are
// needed.  Loading them at the point of use put one or two exposed memory round trips into the middle of every tile.
struct GhostRaw {
  double q0, q1, q2, q3;
};

template <int DIR>
__device__ __forceinline__ GhostRaw ghost_prefetch(const RectView& v, int b, int p, long line) {
  const int P = DIR == 0 ? v.d.px : v.d.py;
  const long nlines = DIR == 0 ? v.d.ny : v.d.nx;
  if (line > nlines - 1) line = nlines - 1;      // lanes beyond the grid: any valid address, result discarded
  GhostRaw g;
  if (QP_ABL & 1) { g.q0 = g.q1 = g.q2 = g.q3 = 0.0; return g; }
  if (v.decoupled[DIR]) {
    const double* ir = v.iface[DIR] + (long)b * (2 * P + 2) * nlines + line;
    g.q0 = ir[(long)(2 * p) * nlines];
    g.q1 = ir[(long)(2 * p + 1) * nlines];
    g.q2 = ir[(long)(2 * p + 2) * nlines];
    g.q3 = ir[(long)(2 * p + 3) * nlines];
  } else {
    const double* z = v.z[DIR] + (long)b * 2 * P * nlines + line;
    g.q0 = z[(long)max(2 * p - 1, 0) * nlines];
    g.q1 = z[(long)min(2 * p + 2, 2 * P - 1) * nlines];
    g.q2 = g.q3 = 0.0;
  }
  return g;
}

template <int DIR>
__device__ __forceinline__ void ghost_finish(const RectView& v, int b, int p, bool on, const GhostRaw& g, double& gl,
                                             double& gr) {
  const int P = DIR == 0 ? v.d.px : v.d.py;
  const int pg = p + (DIR == 0 ? v.d.i0 : v.d.j0) / TS;          // global chunk index
  const int PG = DIR == 0 ? v.d.gpx : v.d.gpy;
  gl = 0.0;
  gr = 0.0;
  if (v.decoupled[DIR]) {
    const ctab_t ic = as_const(v.icoef[DIR] + (long)b * (P + 1) * 3);
    if (pg > 0) gl = fma(ic[p * 3], g.q1, g.q0) * ic[p * 3 + 2];
    if (pg < PG - 1) gr = fma(ic[(p + 1) * 3 + 1], g.q2, g.q3) * ic[(p + 1) * 3 + 2];
  } else {
    if (p > 0) gl = g.q0;
    if (p < P - 1) gr = g.q1;
  }
  if (!on) gl = gr = 0.0;
}

// ---------------------------------------------------------------------------------------------------------
// x-kernel: finish the x-solve, apply the explicit x-operator, eliminate along y.   buf: rhs1 -> rhs2 in place
// ---------------------------------------------------------------------------------------------------------
// EXPLICIT = false is the plain solve (I - a Lx)^-1 used by the exact-CN preconditioner: no explicit operator, no sources.
#ifdef QP_FORCE_WAVES      // occupancy experiments (tools/ablate.sh): waves per SIMD forced through the register budget
#define QP_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(QP_FORCE_WAVES, QP_FORCE_WAVES)))
#else
#define QP_WAVES_ATTR
#endif

template <bool EXPLICIT, int STREAM, bool COMPACT, bool SRC = false>
__global__ void __launch_bounds__(64) QP_WAVES_ATTR rect_x_kernel(RectView v, double* __restrict__ buf) {
  __shared__ double lds[LDS_DOUBLES];
  const int lane = threadIdx.x;
  const TileCoord t = tile_coord<STREAM>(v.d);
  const long ncell = (long)v.d.ny * v.d.nx;
  double* plane = buf + (long)t.b * ncell;
  const double a = as_const(v.alpha)[t.b];
  const int row = t.j0 + lane;         // lane = row between the transposes
  const bool row_on = lane < t.nr;
  // small loads first: interface values of this tile's rows, the x-table (solve + explicit operator), the elimination
  // slots of the y-table
  const GhostRaw graw = ghost_prefetch<0>(v, t.b, t.tx, row);
  const int varx = chunk_variant(t.tx + v.d.i0 / TS, v.d.gpx), vary = chunk_variant(t.ty + v.d.j0 / TS, v.d.gpy);
  const auto cx = coef_source<COMPACT, T_W, T_EW - T_W>(v, 0, t.b, varx, lane);
  const auto cy = coef_source<COMPACT, T_EW, T_NSLOT - T_EW>(v, 1, t.b, vary, lane);
  double e[TS];
  load_cols(plane, t, v.d.nx, lane, e);
  if constexpr (COMPACT) {
    const CoefCompact parts[2] = {cx, cy};
    warm_scalar_cache(parts);
  }
  transpose64(e, lds, lane);
  double gl, gr;
  ghost_finish<0>(v, t.b, t.tx, row_on, graw, gl, gr);
  e[0] = fma(a, gl, e[0]);
  e[TS - 1] = fma(a, gr, e[TS - 1]);   // gr != 0 only for full-length chunks
  thomas64(e, cx);
  double srow = 0.0;                   // sources of the y-faces (up/down) belong to rows 0 and ny-1
  if (row + v.d.j0 == 0) srow += a * v.other_src[1][0];
  if (row + v.d.j0 == v.d.gny - 1) srow += a * v.other_src[1][1];
  if (EXPLICIT) explicit64(e, gl, gr, cx, row_on ? srow : 0.0);
  transpose64(e, lds, lane);
  // lane = column again
  if (SRC) add_source_cols(v.bsrc + (long)t.b * ncell, t, v.d.nx, lane, v.bscale, e);
  store_cols(plane, t, v.d.nx, lane, e);
  if (QP_ABL & 8) return;
  double yf, yl;
  ends64(e, cy, yf, yl);
  if (lane < t.nc) {
    double* ir = v.iface[1] + (long)t.b * (2 * v.d.py + 2) * v.d.nx;
    ir[(long)(2 * t.ty + 1) * v.d.nx + t.i0 + lane] = yf;
    ir[(long)(2 * t.ty + 2) * v.d.nx + t.i0 + lane] = yl;
  }
}

// ---------------------------------------------------------------------------------------------------------
// y-kernel.  MODE 0 (entry): src = u, no solve, halo rows from u;  dst = rhs1, x-elimination
//            MODE 1 (carry): src = rhs2, y-solve, rhs1' = (I + a Ly) u' + a S; dst = rhs1', x-elimination
//            MODE 2 (exit):  src = rhs2, y-solve, dst = u'
//            MODE 3 (reduce): src = rhs of an x-solve; only its reduced right-hand sides are formed (nothing stored)
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int STREAM, bool COMPACT, bool SRC = false>
__global__ void __launch_bounds__(64) QP_WAVES_ATTR rect_y_kernel(RectView v, const double* src, double* dst) {  // src may alias dst
  __shared__ double lds[LDS_DOUBLES];
  const int lane = threadIdx.x;
  const TileCoord t = tile_coord<STREAM>(v.d);
  const long ncell = (long)v.d.ny * v.d.nx;
  const double* splane = src + (long)t.b * ncell;
  double* dplane = dst + (long)t.b * ncell;
  const double a = as_const(v.alpha)[t.b];
  const int col = t.i0 + lane;
  const bool col_on = lane < t.nc;
  // small loads first (see rect_x_kernel)
  GhostRaw graw;
  if (MODE == 1 || MODE == 2) graw = ghost_prefetch<1>(v, t.b, t.ty, col);
  const int varx = chunk_variant(t.tx + v.d.i0 / TS, v.d.gpx), vary = chunk_variant(t.ty + v.d.j0 / TS, v.d.gpy);
  const auto cy = coef_source<COMPACT, T_W, MODE == 3 ? 0 : T_EW - T_W>(v, 1, t.b, vary, lane);
  const auto cx = coef_source<COMPACT, T_EW, MODE == 2 ? 0 : T_NSLOT - T_EW>(v, 0, t.b, varx, lane);
  double gu = 0.0, gd = 0.0;           // values of the field just above / below the tile
  if (MODE == 0 && col_on) {
    if (t.ty > 0) gu = splane[(long)(t.j0 - 1) * v.d.nx + col];
    else if (v.d.j0 > 0) gu = v.uhalo[0][(long)t.b * v.d.nx + col];                 // row owned by the rank above
    if (t.ty < v.d.py - 1) gd = splane[(long)(t.j0 + TS) * v.d.nx + col];
    else if (v.d.j0 + v.d.ny < v.d.gny) gd = v.uhalo[1][(long)t.b * v.d.nx + col];  // row owned by the rank below
  }
  double e[TS];
  load_cols(splane, t, v.d.nx, lane, e);
  if constexpr (COMPACT) {
    if (MODE == 2) {
      const CoefCompact parts[1] = {cy};
      warm_scalar_cache(parts);
    } else {
      const CoefCompact parts[2] = {cy, cx};
      warm_scalar_cache(parts);
    }
  }
  if (MODE == 1 || MODE == 2) {
    ghost_finish<1>(v, t.b, t.ty, col_on, graw, gu, gd);
    e[0] = fma(a, gu, e[0]);
    e[TS - 1] = fma(a, gd, e[TS - 1]);
    thomas64(e, cy);
  }
  if (MODE == 2) {
    store_cols(dplane, t, v.d.nx, lane, e);
    return;
  }
  double scol = 0.0;                   // sources of the x-faces (left/right) belong to columns 0 and nx-1
  if (col + v.d.i0 == 0) scol += a * v.other_src[0][0];
  if (col + v.d.i0 == v.d.gnx - 1) scol += a * v.other_src[0][1];
  if (MODE != 3) {
    explicit64(e, gu, gd, cy, col_on ? scol : 0.0);
    if (SRC) add_source_cols(v.bsrc + (long)t.b * ncell, t, v.d.nx, lane, v.bscale, e);
    store_cols(dplane, t, v.d.nx, lane, e);
  }
  if (QP_ABL & 8) return;
  transpose64(e, lds, lane);
  double yf, yl;
  ends64(e, cx, yf, yl);
  if (lane < t.nr) {
    double* ir = v.iface[0] + (long)t.b * (2 * v.d.px + 2) * v.d.ny;
    ir[(long)(2 * t.tx + 1) * v.d.ny + t.j0 + lane] = yf;
    ir[(long)(2 * t.tx + 2) * v.d.ny + t.j0 + lane] = yl;
  }
}

// ---------------------------------------------------------------------------------------------------------
// reduced banded systems: one thread per (field, line); unknown order F_0, E_0, F_1, E_1, ...
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) rect_reduced_kernel(RectView v, int dir) {
  const int nlines = dir == 0 ? v.d.ny : v.d.nx;
  const int m = 2 * (dir == 0 ? v.d.px : v.d.py);
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)nlines * v.d.nfield) return;
  const int b = (int)(gid / nlines);
  const int line = (int)(gid - (long)b * nlines);
  const double* lu = v.lu[dir] + (long)b * 5 * m;
  const double* rhs = v.iface[dir] + ((long)b * (m + 2) + 1) * nlines + line;
  double* z = v.z[dir] + (long)b * m * nlines + line;
  double y1 = 0.0, y2 = 0.0;  // y_{i-1}, y_{i-2}
  for (int i = 0; i < m; ++i) {
    const double y = rhs[(long)i * nlines] - lu[i] * y1 - lu[m + i] * y2;
    z[(long)i * nlines] = y;
    y2 = y1;
    y1 = y;
  }
  double z1 = 0.0, z2 = 0.0;  // z_{i+1}, z_{i+2}
  for (int i = m - 1; i >= 0; --i) {
    const double zi = (z[(long)i * nlines] - lu[3 * m + i] * z1 - lu[4 * m + i] * z2) * lu[2 * m + i];
    z[(long)i * nlines] = zi;
    z2 = z1;
    z1 = zi;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side: tables, plan
// ---------------------------------------------------------------------------------------------------------
static void chunk_diagonal(const DirSpec& s, double a, int p, int L, int len, std::vector<double>& bdiag) {
  bdiag.assign(len, 0.0);
  for (int k = 0; k < len; ++k) {
    const int gk = p * L + k;
    const double links = (gk > 0 ? 1.0 : 0.0) + (gk < s.n - 1 ? 1.0 : 0.0);
    const double e = (gk == 0 ? s.e_lo : 0.0) + (gk == s.n - 1 ? s.e_hi : 0.0);
    bdiag[k] = 1.0 + a * (links + e);
  }
}

static void solve_chunk(const std::vector<double>& bdiag, double a, std::vector<double>& rhs) {
  const int len = (int)bdiag.size();
  std::vector<double> w(len);
  double dp = 0.0;
  for (int k = 0; k < len; ++k) {
    w[k] = 1.0 / (bdiag[k] - (k > 0 ? a * a * w[k - 1] : 0.0));
    dp = (rhs[k] + (k > 0 ? a * dp : 0.0)) * w[k];
    rhs[k] = dp;
  }
  for (int k = len - 2; k >= 0; --k) rhs[k] += a * w[k] * rhs[k + 1];
}

// table of chunk p of a line cut into chunks of L cells (L = TS for the 64 x 64 tiles, FS for the fine tiles): [T_NSLOT][L]
void build_chunk_table_len(const DirSpec& s, double a, int p, int L, double* tab, double ends[4]) {
  const int len = std::min(L, s.n - p * L);
  std::vector<double> bd;
  chunk_diagonal(s, a, p, L, len, bd);
  for (int k = 0; k < L * T_NSLOT; ++k) tab[k] = 0.0;
  double wprev = 0.0;
  for (int k = 0; k < L; ++k) {
    if (k < len) {
      const double w = 1.0 / (bd[k] - (k > 0 ? a * a * wprev : 0.0));
      tab[T_W * L + k] = w;
      tab[T_AWF * L + k] = k > 0 ? a * w : 0.0;
      tab[T_AWB * L + k] = k < len - 1 ? a * w : 0.0;
      wprev = w;
      const int gk = p * L + k;
      const bool lm = gk > 0, lp = gk < s.n - 1;
      const double e = (gk == 0 ? s.e_lo : 0.0) + (gk == s.n - 1 ? s.e_hi : 0.0);
      tab[T_CM * L + k] = lm ? a : 0.0;
      tab[T_CP * L + k] = lp ? a : 0.0;
      tab[T_C0 * L + k] = 1.0 - a * ((lm ? 1.0 : 0.0) + (lp ? 1.0 : 0.0) + e) - s.c0_shift;
      tab[T_SRC * L + k] = a * ((gk == 0 ? s.s_lo : 0.0) + (gk == s.n - 1 ? s.s_hi : 0.0));
    } else {
      tab[T_W * L + k] = 1.0;
    }
  }
  // elimination slots: forward sweep = the LU pivots again, but padded entries pass the value through; backward sweep =
  // the UL pivots v_k = 1 / (b_k - a^2 v_{k+1}), zero on padded entries
  double vnext = 0.0;
  for (int k = L - 1; k >= 0; --k) {
    if (k < len) {
      const double vk = 1.0 / (bd[k] - (k < len - 1 ? a * a * vnext : 0.0));
      tab[T_EV * L + k] = vk;
      tab[T_EAV * L + k] = k < len - 1 ? a * vk : 0.0;
      vnext = vk;
      tab[T_EW * L + k] = tab[T_W * L + k];
      tab[T_EAWF * L + k] = tab[T_AWF * L + k];
    } else {
      tab[T_EW * L + k] = 0.0;
      tab[T_EAWF * L + k] = 1.0;
    }
  }
  std::vector<double> g(len, 0.0), h(len, 0.0);
  g[0] = 1.0;
  h[len - 1] = 1.0;
  solve_chunk(bd, a, g);
  solve_chunk(bd, a, h);
  ends[0] = g[0];
  ends[1] = g[len - 1];
  ends[2] = h[0];
  ends[3] = h[len - 1];
}

void build_chunk_table(const DirSpec& s, double a, int p, double* tab, double ends[4]) {
  build_chunk_table_len(s, a, p, TS, tab, ends);
}

// Reduced-system data of one (field, direction).  `s` describes the GLOBAL line; the local block holds chunks
// [p0, p0 + Ploc).  Fills icoef[Ploc+1][3] (interface q sits between local chunks q-1 and q; q = 0 / Ploc touch the
// neighbouring blocks) and, when `lu` is non-NULL (no decomposition: p0 == 0, Ploc == s.P), the banded LU factors
// [5][2P] (2 sub-, 2 super-diagonals, no pivoting: the matrix is strictly diagonally dominant).  Returns the largest
// far-coupling weight of the global reduced matrix.
static double reduced_tables(const DirSpec& s, double a, int p0, int Ploc, double* lu, double* icoef, int L = TS) {
  std::vector<double> tab(L * T_NSLOT);
  std::vector<double> near_g(s.P), near_h(s.P), far_g(s.P), far_h(s.P);
  double far = 0.0;
  for (int p = 0; p < s.P; ++p) {
    double ends[4];
    build_chunk_table_len(s, a, p, L, tab.data(), ends);
    near_g[p] = a * ends[0];   // weight of E_{p-1} in the F_p equation
    far_g[p] = a * ends[1];    // weight of E_{p-1} in the E_p equation
    far_h[p] = a * ends[2];    // weight of F_{p+1} in the F_p equation
    near_h[p] = a * ends[3];   // weight of F_{p+1} in the E_p equation
    if (p > 0) far = std::max(far, std::fabs(far_g[p]));
    if (p < s.P - 1) far = std::max(far, std::fabs(far_h[p]));
  }
  for (int q = 0; q <= Ploc; ++q) {
    const int pl = p0 + q - 1, pr = p0 + q;   // global chunks left / right of the interface
    const double sc = (pl >= 0 && pr < s.P) ? near_h[pl] : 0.0;
    const double tc = (pl >= 0 && pr < s.P) ? near_g[pr] : 0.0;
    icoef[q * 3] = sc;
    icoef[q * 3 + 1] = tc;
    icoef[q * 3 + 2] = 1.0 / (1.0 - sc * tc);
  }
  if (!lu) return far;
  const int m = 2 * s.P;
  std::vector<double> A((size_t)m * m, 0.0);
  for (int p = 0; p < s.P; ++p) {
    const int f = 2 * p, e = 2 * p + 1;
    A[(size_t)f * m + f] = 1.0;
    A[(size_t)e * m + e] = 1.0;
    if (p > 0) {
      A[(size_t)f * m + (2 * p - 1)] += -near_g[p];
      A[(size_t)e * m + (2 * p - 1)] += -far_g[p];
    }
    if (p < s.P - 1) {
      A[(size_t)f * m + (2 * p + 2)] += -far_h[p];
      A[(size_t)e * m + (2 * p + 2)] += -near_h[p];
    }
  }
  for (int k = 0; k < m; ++k) {
    for (int i = k + 1; i < std::min(m, k + 3); ++i) {
      const double l = A[(size_t)i * m + k] / A[(size_t)k * m + k];
      A[(size_t)i * m + k] = l;
      for (int j = k + 1; j < std::min(m, k + 5); ++j) A[(size_t)i * m + j] -= l * A[(size_t)k * m + j];
    }
  }
  for (int i = 0; i < m; ++i) {
    lu[i] = i >= 1 ? A[(size_t)i * m + i - 1] : 0.0;
    lu[m + i] = i >= 2 ? A[(size_t)i * m + i - 2] : 0.0;
    lu[2 * m + i] = 1.0 / A[(size_t)i * m + i];
    lu[3 * m + i] = i + 1 < m ? A[(size_t)i * m + i + 1] : 0.0;
    lu[4 * m + i] = i + 2 < m ? A[(size_t)i * m + i + 2] : 0.0;
  }
  return far;
}

#include "qp_adi_fine.inc"

}  // namespace qp

// launches NAME<ARG, STREAM> for the plan's stream mode (0 cached, 2 non-temporal stores, 3 non-temporal both)
#define QP_LAUNCH_STREAMED_C(mode, NAME, ARG, C, ...)                              \
  do {                                                                             \
    if ((mode) == 0) hipLaunchKernelGGL((NAME<ARG, 0, C>), __VA_ARGS__);           \
    else if ((mode) == 2) hipLaunchKernelGGL((NAME<ARG, 2, C>), __VA_ARGS__);      \
    else hipLaunchKernelGGL((NAME<ARG, 3, C>), __VA_ARGS__);                       \
  } while (0)
#define QP_LAUNCH_STREAMED_SRC(mode, compact, NAME, ARG, ...)                                              \
  do {                                                                                                   \
    if (compact) {                                                                                       \
      if ((mode) == 0) hipLaunchKernelGGL((NAME<ARG, 0, true, true>), __VA_ARGS__);                      \
      else if ((mode) == 2) hipLaunchKernelGGL((NAME<ARG, 2, true, true>), __VA_ARGS__);                 \
      else hipLaunchKernelGGL((NAME<ARG, 3, true, true>), __VA_ARGS__);                                  \
    } else {                                                                                             \
      if ((mode) == 0) hipLaunchKernelGGL((NAME<ARG, 0, false, true>), __VA_ARGS__);                     \
      else if ((mode) == 2) hipLaunchKernelGGL((NAME<ARG, 2, false, true>), __VA_ARGS__);                \
      else hipLaunchKernelGGL((NAME<ARG, 3, false, true>), __VA_ARGS__);                                 \
    }                                                                                                    \
  } while (0)
// ... and for the plan's table form (compact: constant middle of every slot; otherwise every entry is fetched)
#define QP_LAUNCH_STREAMED(mode, compact, NAME, ARG, ...)                          \
  do {                                                                             \
    if (compact) QP_LAUNCH_STREAMED_C(mode, NAME, ARG, true, __VA_ARGS__);         \
    else QP_LAUNCH_STREAMED_C(mode, NAME, ARG, false, __VA_ARGS__);                \
  } while (0)

#define QP_LAUNCH_FINE(mode, NAME, ARG, ...)                                       \
  do {                                                                             \
    if ((mode) == 0) hipLaunchKernelGGL((NAME<ARG, 0>), __VA_ARGS__);              \
    else if ((mode) == 2) hipLaunchKernelGGL((NAME<ARG, 2>), __VA_ARGS__);         \
    else hipLaunchKernelGGL((NAME<ARG, 3>), __VA_ARGS__);                          \
  } while (0)

#define QP_LAUNCH_FINE_SRC(mode, NAME, ARG, ...)                                   \
  do {                                                                             \
    if ((mode) == 0) hipLaunchKernelGGL((NAME<ARG, 0, true>), __VA_ARGS__);        \
    else if ((mode) == 2) hipLaunchKernelGGL((NAME<ARG, 2, true>), __VA_ARGS__);   \
    else hipLaunchKernelGGL((NAME<ARG, 3, true>), __VA_ARGS__);                    \
  } while (0)

struct qp_adi_rect_plan {
  qp::RectView view;
  // fine tiles (qp_adi_fine.inc): when `fine`, every pass of this plan runs the 32-cell-chunk kernels on their own
  // interface arrays (the 64 x 64 view above stays valid for qp_adi_rect_combine and the plan queries)
  bool fine = false;
  qp::FineView fview;
  double* d_fctab = nullptr;
  double* d_ficoef[2] = {nullptr, nullptr};
  double* d_fiface[2] = {nullptr, nullptr};
  double* d_alpha = nullptr;
  double* d_tab = nullptr;
  double* d_ctab = nullptr;
  double* d_lu[2] = {nullptr, nullptr};
  double* d_icoef[2] = {nullptr, nullptr};
  double* d_iface[2] = {nullptr, nullptr};
  double* d_z[2] = {nullptr, nullptr};
  double* d_uhalo[2] = {nullptr, nullptr};
  double* d_slab = nullptr;  // ONE device allocation behind every table / interface pointer above and below (SlabBuilder)
  double* d_work = nullptr;  // [nfield][ncell] carried right-hand side
  bool owns_work = true;     // false: d_work belongs to another plan (qp_adi_rect_plan_create_pr with `share`)
  double pr_scale = 0.0;     // != 0: Peaceman-Rachford iteration plan, 1 / (1/2 + p)
  long ncell = 0;
  bool decomposed = false;
  double bc_diag[4] = {0, 0, 0, 0};   // left, right, up, down (1/dx^2 units)
  double bc_src[4] = {0, 0, 0, 0};
};

namespace qp {

struct RectSides {
  double dl, dr, du, dd;     // boundary diagonal terms of the four sides
  double sl, sr, su, sd;     // boundary sources
};

// out = c0 u + cx (a Lx u) + cy (a Ly u) + cs a (sx + sy) + cr rin on a full rectangle with one boundary condition per side:
// the 5-point operator of qp_stencil_combine without its per-cell geometry arrays (33 B per cell) - positions decide.  With
// `part` non-NULL every block also leaves max |out| of its cells there (the convergence check of the exact-CN iteration
// then needs no pass of its own; NaN is propagated as +inf).
// STORE = false: only the norm is wanted (the residual CHECK of the exact-CN step: its plane is read again only when the
// check fails) - one plane transfer less.  RB: rows per band of the 4-cells-per-thread branch (two halo rows per band).
template <bool STORE, int RB>
__global__ void __launch_bounds__(256) rect_combine_kernel(int ny, int nx, int nfield, const double* __restrict__ alpha,
                                                           RectSides g, const double* __restrict__ u,
                                                           const double* __restrict__ rin, double* __restrict__ out,
                                                           double c0, double cx, double cy, double cs, double cr,
                                                           double* __restrict__ part) {
  const long ncell = (long)ny * nx;
  double m = 0.0;
  auto cell = [&](double a, int i, int j, double um, double up, double upp, double uu, double ud) {
    double lx = 0.0, ly = 0.0, src = 0.0;
    if (i > 0) lx += um - up; else { lx -= g.dl * up; src += g.sl; }
    if (i < nx - 1) lx += upp - up; else { lx -= g.dr * up; src += g.sr; }
    if (j > 0) ly += uu - up; else { ly -= g.du * up; src += g.su; }
    if (j < ny - 1) ly += ud - up; else { ly -= g.dd * up; src += g.sd; }
    return c0 * up + cx * (a * lx) + cy * (a * ly) + cs * (a * src);
  };
  if ((nx & 3) == 0) {
    // A block walks bands of RB rows of one field; a thread owns a strip 4 cells wide and walks down the band with the row
    // above, the row itself and the row below in registers: every row of u is loaded once per band (+2 halo rows per RB)
    // as two 16-byte loads per thread, the two cells beside the strip as 8-byte loads (cache hits) - no per-cell division.
    const int bands_per_field = (ny + RB - 1) / RB;
    const int chunk = 4 * blockDim.x, chunks = (nx + chunk - 1) / chunk;      // column chunks of one strip per thread
    for (long item = blockIdx.x; item < (long)nfield * bands_per_field * chunks; item += gridDim.x) {
      const int band = (int)(item / chunks), ch = (int)(item - (long)band * chunks);
      const int b = band / bands_per_field, j0 = (band - b * bands_per_field) * RB;
      const double a = alpha[b];
      const double* ub0 = u + (long)b * ncell;
      const int i = ch * chunk + 4 * threadIdx.x;
      if (i < nx) {
        auto load = [&](int j, double2& q0, double2& q1) {
          const double* p = ub0 + (long)j * nx + i;
          q0 = *reinterpret_cast<const double2*>(p);
          q1 = *reinterpret_cast<const double2*>(p + 2);
        };
        double2 c0v, c1v, u0, u1, d0, d1;
        load(j0, c0v, c1v);
        u0 = c0v; u1 = c1v;
        if (j0 > 0) load(j0 - 1, u0, u1);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int j = j0 + r;
          if (j >= ny) break;
          d0 = c0v; d1 = c1v;
          if (j < ny - 1) load(j + 1, d0, d1);
          const double* ub = ub0 + (long)j * nx;
          const double left = i > 0 ? ub[i - 1] : 0.0, right = i + 4 < nx ? ub[i + 4] : 0.0;
          double res[4] = {cell(a, i, j, left, c0v.x, c0v.y, u0.x, d0.x), cell(a, i + 1, j, c0v.x, c0v.y, c1v.x, u0.y, d0.y),
                           cell(a, i + 2, j, c0v.y, c1v.x, c1v.y, u1.x, d1.x), cell(a, i + 3, j, c1v.x, c1v.y, right, u1.y, d1.y)};
          const long t0 = (long)b * ncell + (long)j * nx + i;
          if (cr != 0.0) {
            const double2 q0 = *reinterpret_cast<const double2*>(rin + t0), q1 = *reinterpret_cast<const double2*>(rin + t0 + 2);
            res[0] += cr * q0.x; res[1] += cr * q0.y; res[2] += cr * q1.x; res[3] += cr * q1.y;
          }
          if (STORE) {
            *reinterpret_cast<double2*>(out + t0) = make_double2(res[0], res[1]);
            *reinterpret_cast<double2*>(out + t0 + 2) = make_double2(res[2], res[3]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double v = fabs(res[q]);
            m = (v != v) ? __builtin_huge_val() : fmax(m, v);
          }
          u0 = c0v; u1 = c1v;
          c0v = d0; c1v = d1;
        }
      }
    }
  } else {
    // a block walks (field, row) pairs, its threads the cells of the row
    for (int line = blockIdx.x; line < nfield * ny; line += gridDim.x) {
      const int b = line / ny, j = line - b * ny;
      const double a = alpha[b];
      const double* ub = u + (long)b * ncell + (long)j * nx;
      const long t0 = (long)b * ncell + (long)j * nx;
      for (int i = threadIdx.x; i < nx; i += blockDim.x) {
        double res = cell(a, i, j, i > 0 ? ub[i - 1] : 0.0, ub[i], i < nx - 1 ? ub[i + 1] : 0.0, j > 0 ? ub[i - nx] : 0.0,
                          j < ny - 1 ? ub[i + nx] : 0.0);
        if (cr != 0.0) res += cr * rin[t0 + i];
        if (STORE) out[t0 + i] = res;
        const double v = fabs(res);
        m = (v != v) ? __builtin_huge_val() : fmax(m, v);
      }
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

// Fine tiles are eligible on undecomposed grids whose extents are multiples of 64.  QPSIM_FINE_TILES=0 / 1 forces the
// choice (1: whenever the plan qualifies); the default is the size rule measured on MI355X.
// All small device arrays of a plan live in one allocation: a plan used to cost ~17 hipMalloc + as many blocking copies
// (~3 ms), which showed when the host builds a Peaceman-Rachford cycle of 7-8 plans for a 200-step run on a 64^2 grid.
// Uploads are staged in one host buffer and copied once; the zero-filled arrays follow and get one hipMemset.
struct SlabBuilder {
  struct Item {
    double** dst;
    size_t count;
    const double* host;      // nullptr: zero-filled
  };
  std::vector<Item> items;
  std::vector<std::vector<double>> owned;      // host arrays whose builders have returned
  static size_t pad(size_t n) { return (n + 31) & ~(size_t)31; }      // 256-byte granules
  void upload(const std::vector<double>& h, double** dst) { items.push_back({dst, h.size(), h.data()}); }
  void upload_owned(std::vector<double>&& h, double** dst) {
    owned.push_back(std::move(h));
    items.push_back({dst, owned.back().size(), owned.back().data()});
  }
  void zeros(size_t count, double** dst) { items.push_back({dst, count, nullptr}); }
  bool commit(double** slab) {
    size_t up = 0, total = 0;
    for (const Item& it : items) if (it.host) up += pad(it.count);
    total = up;
    for (const Item& it : items) if (!it.host) total += pad(it.count);
    if (hipMalloc((void**)slab, std::max<size_t>(total, 32) * sizeof(double)) != hipSuccess) return false;
    std::vector<double> stage(up, 0.0);
    size_t o = 0, z = up;
    for (const Item& it : items) {
      if (it.host) {
        std::copy(it.host, it.host + it.count, stage.begin() + o);
        *it.dst = *slab + o;
        o += pad(it.count);
      } else {
        *it.dst = *slab + z;
        z += pad(it.count);
      }
    }
    if (up && hipMemcpy(*slab, stage.data(), up * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (total > up && hipMemset(*slab + up, 0, (total - up) * sizeof(double)) != hipSuccess) return false;
    return true;
  }
};

static bool fine_tiles_allowed() {      // QPSIM_FINE_TILES=0 switches the fine kernels off everywhere
  const char* e = getenv("QPSIM_FINE_TILES");
  return !e || atoi(e) != 0;
}

static bool fine_tiles_wanted(int nfield, int ny, int nx) {
  if (ny % 64 != 0 || nx % 64 != 0) return false;
  if (const char* e = getenv("QPSIM_FINE_TILES")) return atoi(e) != 0;
  // sweep time in us, 64 x 64 tiles / fine tiles: 1024^2 7.55 / 5.91, 2048^2 15.1 / 13.2, 512^2 x 12 11.8 / 10.2,
  // 4160 x 2176 28.7 / 25.8, 4096^2 47.0 / 45.9; beyond the cached regime 5760^2 86.9 / 92.7, 8192^2 189 / 187
  return stream_mode((size_t)nfield * ny * nx * sizeof(double)) == 0;
}

// Tables, interface coefficients and interface arrays of the fine view; leaves plan->fine false (and no error) when the
// plan does not qualify: chunks of 32 cells not decoupled at this r D, or a table without the compact form.
static void fine_plan_prepare(qp_adi_rect_plan* plan, double r, const double* dcoef_host, const DirSpec (&coarse)[2],
                              SlabBuilder& slab) {
  const RectView& v = plan->view;
  const int nfield = v.d.nfield, ny = v.d.ny, nx = v.d.nx;
  FineView& f = plan->fview;
  f.ny = ny; f.nx = nx; f.nfield = nfield;
  f.py = ny / FS; f.px = nx / FS;
  f.stream = v.d.stream;
  for (int d = 0; d < 2; ++d)
    for (int k = 0; k < 2; ++k) f.other_src[d][k] = v.other_src[d][k];
  DirSpec spec[2] = {coarse[0], coarse[1]};
  spec[0].P = f.px;
  spec[1].P = f.py;
  std::vector<double> tab((size_t)T_NSLOT * FS);
  std::vector<double> ctab((size_t)2 * nfield * 4 * 2 * CT_PART, 0.0);
  std::vector<double> icoef[2];
  for (int d = 0; d < 2; ++d) icoef[d].assign((size_t)nfield * (spec[d].P + 1) * 3, 0.0);
  for (int b = 0; b < nfield; ++b) {
    const double a = r * dcoef_host[b];
    for (int d = 0; d < 2; ++d) {
      const int P = spec[d].P;       // >= 2
      for (int var = 0; var < 3; ++var) {
        int p;
        if (var == 0) { if (P < 3) continue; p = 1; }
        else if (var == 1) p = 0;
        else p = P - 1;
        double ends[4];
        build_chunk_table_len(spec[d], a, p, FS, tab.data(), ends);
        {
          // scaled eliminations (ends32): eawf'_k = eawf_k ew_{k-1} / ew_k, eav'_k = eav_k ev_{k+1} / ev_k; of EW / EV only
          // the last / first entry is read.  (Full chunks: no padded entries, every pivot is positive.)
          double* ew = &tab[T_EW * FS]; double* eawf = &tab[T_EAWF * FS];
          double* ev = &tab[T_EV * FS]; double* eav = &tab[T_EAV * FS];
          for (int k = FS - 1; k >= 1; --k) eawf[k] = eawf[k] * ew[k - 1] / ew[k];
          eawf[0] = 0.0;
          for (int k = 0; k < FS - 1; ++k) eav[k] = eav[k] * ev[k + 1] / ev[k];
          eav[FS - 1] = 0.0;
        }
        if (!table_is_compact_len(FS, tab.data())) return;
        build_compact_table_len(FS, tab.data(), &ctab[(((size_t)d * nfield + b) * 4 + var) * 2 * CT_PART]);
      }
      const double far = reduced_tables(spec[d], a, 0, P, nullptr, &icoef[d][(size_t)b * (P + 1) * 3], FS);
      if (!(far < kFarCouplingDrop)) return;
    }
  }
  slab.upload_owned(std::move(ctab), &plan->d_fctab);
  for (int d = 0; d < 2; ++d) {
    const size_t nlines = d == 0 ? ny : nx;
    slab.upload_owned(std::move(icoef[d]), &plan->d_ficoef[d]);
    slab.zeros((size_t)nfield * (2 * spec[d].P + 2) * nlines, &plan->d_fiface[d]);
  }
  f.bsrc = nullptr;
  f.bscale = 0.0;
  plan->fine = true;      // the view's pointers are bound after SlabBuilder::commit (rect_plan_create_impl)
}

// the passes of qp_adi_rect_phase on the fine view (interfaces always decoupled: the reduced phases are empty)
static int fine_phase(qp_adi_rect_plan* plan, int phase, double* u, hipStream_t stream) {
  const FineView& f = plan->fview;
  const unsigned tiles = (unsigned)((long)f.nfield * (f.ny / 64) * f.px);     // = nfield * py * (nx / 64)
  double* w = plan->d_work;
  switch (phase) {
    case QP_ADI_ENTRY:
      QP_LAUNCH_FINE(f.stream, fine_y_kernel, 0, dim3(tiles), dim3(64), 0, stream, f, (const double*)u, w);
      break;
    case QP_ADI_REDUCED_X:
    case QP_ADI_REDUCED_Y:
      break;
    case QP_ADI_SWEEP_X:
      QP_LAUNCH_FINE(f.stream, fine_x_kernel, true, dim3(tiles), dim3(64), 0, stream, f, w);
      break;
    case QP_ADI_SWEEP_Y_CARRY:
      QP_LAUNCH_FINE(f.stream, fine_y_kernel, 1, dim3(tiles), dim3(64), 0, stream, f, (const double*)w, w);
      break;
    case QP_ADI_SWEEP_Y_EXIT:
      QP_LAUNCH_FINE(f.stream, fine_y_kernel, 2, dim3(tiles), dim3(64), 0, stream, f, (const double*)w, u);
      break;
    default:
      set_error("qp_adi_rect_phase: unknown phase %d", phase);
      return QP_ERR_INVALID_ARGUMENT;
  }
  return check_launch("qp_adi_rect_phase (fine tiles)");
}

int rect_ablation_mask() { return QP_ABL; }

}  // namespace qp

extern "C" {

int qp_adi_rect_plan_destroy(qp_adi_rect_plan* plan) {
  if (!plan) return QP_OK;
  (void)hipFree(plan->d_slab);      // every table / interface array
  if (plan->owns_work) (void)hipFree(plan->d_work);
  delete plan;
  return QP_OK;
}

// pr_scale != 0: plan of one Peaceman-Rachford iteration (qp_adi_rect_plan_create_pr) - fine tiles wherever the grid
// qualifies (the 64 x 64 view then gets no interface arrays), the carried plane may be borrowed from `share`.
static int rect_plan_create_impl(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                                 const double* bc_diag, const double* bc_src, int32_t force_banded, int32_t gny,
                                 int32_t gnx, int32_t j0, int32_t i0, double pr_scale, qp_adi_rect_plan* share,
                                 qp_adi_rect_plan** out) {
  QP_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  QP_REQUIRE(ny > 0 && nx > 0 && nfield > 0, "ny, nx, nfield must be positive");
  QP_REQUIRE(r > 0.0 && dcoef_host && bc_diag && bc_src, "r must be positive; dcoef/bc arrays non-NULL");
  QP_REQUIRE(j0 >= 0 && i0 >= 0 && j0 + ny <= gny && i0 + nx <= gnx, "block must lie inside the global grid");
  using namespace qp;
  QP_REQUIRE(j0 % TS == 0 && i0 % TS == 0, "block offsets must be multiples of 64");
  QP_REQUIRE((j0 + ny == gny || ny % TS == 0) && (i0 + nx == gnx || nx % TS == 0),
             "interior block extents must be multiples of 64");
  for (int b = 0; b < nfield; ++b) QP_REQUIRE(dcoef_host[b] >= 0.0, "diffusion coefficients must be >= 0");
  const bool split[2] = {gnx != nx, gny != ny};   // direction d is cut across ranks
  const bool decomposed = split[0] || split[1];
  QP_REQUIRE(!(decomposed && force_banded), "the banded reduced solve is not available on decomposed grids");
  auto* plan = new qp_adi_rect_plan();
  plan->decomposed = decomposed;
  RectView& v = plan->view;
  v.d = RectDims{ny, nx, nfield, (ny + TS - 1) / TS, (nx + TS - 1) / TS, gny, gnx, j0, i0,
                 (gny + TS - 1) / TS, (gnx + TS - 1) / TS,
                 stream_mode((size_t)nfield * ny * nx * sizeof(double))};
  plan->ncell = (long)ny * nx;
  // bc_* order: left, right, up, down  (x-faces then y-faces); specs describe the GLOBAL lines
  DirSpec spec[2] = {{gnx, v.d.gpx, bc_diag[0], bc_diag[1], bc_src[0], bc_src[1], pr_scale},
                     {gny, v.d.gpy, bc_diag[2], bc_diag[3], bc_src[2], bc_src[3], pr_scale}};
  const int ploc[2] = {v.d.px, v.d.py};
  const int p0[2] = {i0 / TS, j0 / TS};
  for (int k = 0; k < 4; ++k) { plan->bc_diag[k] = bc_diag[k]; plan->bc_src[k] = bc_src[k]; }
  v.other_src[0][0] = bc_src[0];
  v.other_src[0][1] = bc_src[1];
  v.other_src[1][0] = bc_src[2];
  v.other_src[1][1] = bc_src[3];

  std::vector<double> alpha(nfield);
  std::vector<double> tab((size_t)2 * nfield * 4 * T_NSLOT * TS, 0.0);
  bool all_compact = true;
  std::vector<double> ctab((size_t)2 * nfield * 4 * 2 * CT_PART, 0.0);
  std::vector<double> lu[2], icoef[2];
  double far[2] = {0.0, 0.0};
  for (int d = 0; d < 2; ++d) {
    lu[d].assign(split[d] ? 1 : (size_t)nfield * 5 * 2 * spec[d].P, 0.0);
    icoef[d].assign((size_t)nfield * (ploc[d] + 1) * 3, 0.0);
  }
  for (int b = 0; b < nfield; ++b) {
    const double a = r * dcoef_host[b];
    alpha[b] = a;
    for (int d = 0; d < 2; ++d) {
      const int P = spec[d].P;
      // representative global chunk per variant: interior -> 1, first -> 0, last -> P-1, single -> 0
      for (int var = 0; var < 4; ++var) {
        int p;
        if (var == 0) { if (P < 3) continue; p = 1; }
        else if (var == 1) { if (P < 2) continue; p = 0; }
        else if (var == 2) { if (P < 2) continue; p = P - 1; }
        else { if (P != 1) continue; p = 0; }
        double ends[4];
        double* tb = &tab[((((size_t)d * nfield + b) * 4 + var) * T_NSLOT) * TS];
        build_chunk_table(spec[d], a, p, tb, ends);
        all_compact = all_compact && table_is_compact(tb);
        build_compact_table(tb, &ctab[(((size_t)d * nfield + b) * 4 + var) * 2 * CT_PART]);
      }
      far[d] = std::max(far[d], reduced_tables(spec[d], a, p0[d], ploc[d],
                                               split[d] ? nullptr : &lu[d][(size_t)b * 5 * 2 * P],
                                               &icoef[d][(size_t)b * (ploc[d] + 1) * 3]));
    }
  }
  for (int d = 0; d < 2; ++d) v.decoupled[d] = (force_banded == 0 && far[d] < kFarCouplingDrop) ? 1 : 0;
  if ((split[0] && !v.decoupled[0]) || (split[1] && !v.decoupled[1])) {
    delete plan;
    set_error("qp_adi_rect_plan_create_block: r*D too large for a decomposed grid (couplings between 64-cell chunks "
              "along x / y: %.3g / %.3g, limit %.0e; the banded reduced solve is not implemented across ranks; a short "
              "remainder chunk, n %% 64 < ~40, has the same effect)",
              far[0], far[1], kFarCouplingDrop);
    return QP_ERR_UNSUPPORTED;
  }
  SlabBuilder slab;
  slab.upload(alpha, &plan->d_alpha);
  slab.upload(tab, &plan->d_tab);
  slab.upload(ctab, &plan->d_ctab);
  v.compact = all_compact ? 1 : 0;
  if (const char* e = getenv("QPSIM_COMPACT_TABLES")) v.compact = v.compact && atoi(e) != 0;   // 0: force the full form
  v.bsrc = nullptr;
  v.bscale = 0.0;
  plan->pr_scale = pr_scale;
  // fine tiles first: a plan that runs them needs no interface arrays for the 64 x 64 kernels when it is the plan of a
  // Peaceman-Rachford iteration (nothing else ever runs on it)
  if (!decomposed && force_banded == 0 && ny % 64 == 0 && nx % 64 == 0 &&
      (pr_scale != 0.0 ? fine_tiles_allowed() : fine_tiles_wanted(nfield, ny, nx)))
    fine_plan_prepare(plan, r, dcoef_host, spec, slab);
  const bool lean = pr_scale != 0.0 && plan->fine;
  for (int d = 0; d < 2; ++d) {
    const size_t nlines = d == 0 ? ny : nx;
    slab.upload(lu[d], &plan->d_lu[d]);
    slab.upload(icoef[d], &plan->d_icoef[d]);
    slab.zeros(lean ? 1 : (size_t)nfield * (2 * ploc[d] + 2) * nlines, &plan->d_iface[d]);
    slab.zeros(lean ? 1 : (size_t)nfield * 2 * ploc[d] * nlines, &plan->d_z[d]);
    slab.zeros(lean ? 1 : (size_t)nfield * nx, &plan->d_uhalo[d]);
  }
  bool ok = slab.commit(&plan->d_slab);
  if (share) {
    plan->d_work = share->d_work;
    plan->owns_work = false;
  } else {
    ok = ok && hipMalloc((void**)&plan->d_work, (size_t)nfield * plan->ncell * sizeof(double)) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    qp_adi_rect_plan_destroy(plan);
    set_error("qp_adi_rect_plan_create: device allocation or upload failed");
    return QP_ERR_ALLOC;
  }
  v.alpha = plan->d_alpha;
  v.tab = plan->d_tab;
  v.ctab = plan->d_ctab;
  for (int d = 0; d < 2; ++d) {
    v.lu[d] = plan->d_lu[d];
    v.icoef[d] = plan->d_icoef[d];
    v.iface[d] = plan->d_iface[d];
    v.z[d] = plan->d_z[d];
    v.uhalo[d] = plan->d_uhalo[d];
  }
  if (plan->fine) {
    FineView& f = plan->fview;
    f.alpha = plan->d_alpha;
    f.ctab = plan->d_fctab;
    for (int d = 0; d < 2; ++d) {
      f.icoef[d] = plan->d_ficoef[d];
      f.iface[d] = plan->d_fiface[d];
    }
  }
  *out = plan;
  return QP_OK;
}

int qp_adi_rect_plan_create_block(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                                  const double* bc_diag, const double* bc_src, int32_t force_banded, int32_t gny,
                                  int32_t gnx, int32_t j0, int32_t i0, qp_adi_rect_plan** out) {
  return rect_plan_create_impl(ny, nx, nfield, r, dcoef_host, bc_diag, bc_src, force_banded, gny, gnx, j0, i0, 0.0, nullptr,
                               out);
}

// Plan of ONE Peaceman-Rachford iteration with parameter p > 0 for the unsplit Crank-Nicolson system
//   A u = b,  A = I - a (Lx + Ly) = H + V,  H = I/2 - a Lx,  V = I/2 - a Ly   (a = r D per field):
//   (H + p) u* = b - (V - p) u,   (V + p) u' = b - (H - p) u*.
// Divided by 1/2 + p =: 1/s these are the sweeps of an ADI step with a' = s a whose explicit operators are
// (I + a' L) - s I and whose right-hand sides receive s b: the plan holds the tables of a' with the shifted diagonal
// (boundary SOURCES belong to b, the plan has none); qp_adi_rect_pr_iteration runs the three passes.
int qp_adi_rect_plan_create_pr(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                               const double* bc_diag, double p, qp_adi_rect_plan* share, qp_adi_rect_plan** out) {
  QP_REQUIRE(p > 0.0, "the iteration parameter must be positive");
  QP_REQUIRE(!share || (share->ncell == (long)ny * nx && share->view.d.nfield == nfield),
             "the plan whose work plane is shared must have the same shape");
  const double s = 1.0 / (0.5 + p);
  const double zero[4] = {0.0, 0.0, 0.0, 0.0};
  return rect_plan_create_impl(ny, nx, nfield, r * s, dcoef_host, bc_diag, zero, 0, ny, nx, 0, 0, s, share, out);
}

// u <- one CYCLE of Peaceman-Rachford iterations, plans[0 .. nplans) in order.  When every plan runs the fine tiles the
// iterations are carried: the y-kernel of iteration j leaves the right-hand side of the x-solve of iteration j + 1
// (fine_y_next_kernel), two passes per iteration instead of three (6 instead of 8 plane transfers); otherwise one
// qp_adi_rect_pr_iteration per plan.
int qp_adi_rect_pr_cycle(qp_adi_rect_plan* const* plans, int32_t nplans, double* u, const double* b, void* stream_) {
  QP_REQUIRE(plans && nplans >= 1 && u && b, "plans, u, b must be non-NULL and nplans >= 1");
  using namespace qp;
  bool carried = true;
  for (int j = 0; j < nplans; ++j) {
    QP_REQUIRE(plans[j] && plans[j]->pr_scale != 0.0, "not a Peaceman-Rachford plan (qp_adi_rect_plan_create_pr)");
    QP_REQUIRE(plans[j]->ncell == plans[0]->ncell && plans[j]->view.d.nfield == plans[0]->view.d.nfield &&
                   plans[j]->view.d.nx == plans[0]->view.d.nx, "the plans of a cycle must have one shape");
    carried = carried && plans[j]->fine;
  }
  if (const char* e = getenv("QPSIM_PR_CARRIED")) carried = carried && atoi(e) != 0;
  if (!carried) {
    for (int j = 0; j < nplans; ++j) {
      const int rc = qp_adi_rect_pr_iteration(plans[j], u, b, stream_);
      if (rc) return rc;
    }
    return QP_OK;
  }
  hipStream_t stream = (hipStream_t)stream_;
  double* w = plans[0]->d_work;
  const FineView& f0 = plans[0]->fview;
  const unsigned tiles = (unsigned)((long)f0.nfield * (f0.ny / 64) * f0.px);
  auto view = [&](int j) {
    FineView f = plans[j]->fview;
    f.bsrc = b;
    f.bscale = plans[j]->pr_scale;
    return f;
  };
  FineView cur = view(0);
  QP_LAUNCH_FINE_SRC(cur.stream, fine_y_kernel, 0, dim3(tiles), dim3(64), 0, stream, cur, (const double*)u, w);
  for (int j = 0; j < nplans; ++j) {
    QP_LAUNCH_FINE_SRC(cur.stream, fine_x_kernel, true, dim3(tiles), dim3(64), 0, stream, cur, w);
    if (j + 1 < nplans) {
      const FineView nxt = view(j + 1);
      if (cur.stream == 0) hipLaunchKernelGGL((fine_y_next_kernel<0>), dim3(tiles), dim3(64), 0, stream, cur, nxt, w);
      else if (cur.stream == 2) hipLaunchKernelGGL((fine_y_next_kernel<2>), dim3(tiles), dim3(64), 0, stream, cur, nxt, w);
      else hipLaunchKernelGGL((fine_y_next_kernel<3>), dim3(tiles), dim3(64), 0, stream, cur, nxt, w);
      cur = nxt;
    } else {
      QP_LAUNCH_FINE(cur.stream, fine_y_kernel, 2, dim3(tiles), dim3(64), 0, stream, cur, (const double*)w, u);
    }
  }
  return check_launch("qp_adi_rect_pr_cycle");
}

// u <- one Peaceman-Rachford iteration towards A u = b (see qp_adi_rect_plan_create_pr), u and b [nfield][ny*nx].
int qp_adi_rect_pr_iteration(qp_adi_rect_plan* plan, double* u, const double* b, void* stream_) {
  QP_REQUIRE(plan && u && b, "plan, u, b must be non-NULL");
  QP_REQUIRE(plan->pr_scale != 0.0, "not a Peaceman-Rachford plan (qp_adi_rect_plan_create_pr)");
  using namespace qp;
  hipStream_t stream = (hipStream_t)stream_;
  if (!plan->fine) {      // 64 x 64 tiles: any extents, banded reduced systems where the chunks do not decouple
    RectView v = plan->view;
    v.bsrc = b;
    v.bscale = plan->pr_scale;
    const unsigned ctiles = (unsigned)((long)v.d.nfield * v.d.py * v.d.px);
    double* cw = plan->d_work;
    QP_LAUNCH_STREAMED_SRC(v.d.stream, v.compact, rect_y_kernel, 0, dim3(ctiles), dim3(64), 0, stream, v, (const double*)u, cw);
    int rc = qp_adi_rect_phase(plan, QP_ADI_REDUCED_X, u, stream_);
    if (rc) return rc;
    QP_LAUNCH_STREAMED_SRC(v.d.stream, v.compact, rect_x_kernel, true, dim3(ctiles), dim3(64), 0, stream, v, cw);
    rc = qp_adi_rect_phase(plan, QP_ADI_REDUCED_Y, u, stream_);
    if (rc) return rc;
    return qp_adi_rect_phase(plan, QP_ADI_SWEEP_Y_EXIT, u, stream_);
  }
  FineView f = plan->fview;
  f.bsrc = b;
  f.bscale = plan->pr_scale;
  const unsigned tiles = (unsigned)((long)f.nfield * (f.ny / 64) * f.px);
  double* w = plan->d_work;
  QP_LAUNCH_FINE_SRC(f.stream, fine_y_kernel, 0, dim3(tiles), dim3(64), 0, stream, f, (const double*)u, w);
  QP_LAUNCH_FINE_SRC(f.stream, fine_x_kernel, true, dim3(tiles), dim3(64), 0, stream, f, w);
  QP_LAUNCH_FINE(f.stream, fine_y_kernel, 2, dim3(tiles), dim3(64), 0, stream, f, (const double*)w, u);
  return check_launch("qp_adi_rect_pr_iteration");
}

int qp_adi_rect_plan_create(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                            const double* bc_diag, const double* bc_src, int32_t force_banded,
                            qp_adi_rect_plan** out) {
  return qp_adi_rect_plan_create_block(ny, nx, nfield, r, dcoef_host, bc_diag, bc_src, force_banded, ny, nx, 0, 0, out);
}

int qp_adi_rect_plan_decoupled(const qp_adi_rect_plan* plan, int32_t dir) {
  if (!plan || dir < 0 || dir > 1) return -1;
  return plan->view.decoupled[dir];
}

int qp_adi_rect_plan_fine(const qp_adi_rect_plan* plan) { return plan && plan->fine ? 1 : 0; }

int qp_adi_rect_phase(qp_adi_rect_plan* plan, int32_t phase, double* u, void* stream_) {
  QP_REQUIRE(plan != nullptr, "plan is NULL");
  using namespace qp;
  hipStream_t stream = (hipStream_t)stream_;
  if (plan->fine) {
    QP_REQUIRE(u != nullptr || (phase != QP_ADI_ENTRY && phase != QP_ADI_SWEEP_Y_EXIT), "u is NULL");
    return fine_phase(plan, phase, u, stream);
  }
  const RectView& v = plan->view;
  const unsigned tiles = (unsigned)((long)v.d.nfield * v.d.py * v.d.px);
  double* w = plan->d_work;
  switch (phase) {
    case QP_ADI_ENTRY:
      QP_REQUIRE(u != nullptr, "u is NULL");
      QP_LAUNCH_STREAMED(v.d.stream, v.compact, rect_y_kernel, 0, dim3(tiles), dim3(64), 0, stream, v, (const double*)u, w);
      break;
    case QP_ADI_REDUCED_X:
      if (!v.decoupled[0])
        hipLaunchKernelGGL(rect_reduced_kernel, dim3((unsigned)(((long)v.d.ny * v.d.nfield + 63) / 64)), dim3(64), 0,
                           stream, v, 0);
      break;
    case QP_ADI_SWEEP_X:
      QP_LAUNCH_STREAMED(v.d.stream, v.compact, rect_x_kernel, true, dim3(tiles), dim3(64), 0, stream, v, w);
      break;
    case QP_ADI_REDUCED_Y:
      if (!v.decoupled[1])
        hipLaunchKernelGGL(rect_reduced_kernel, dim3((unsigned)(((long)v.d.nx * v.d.nfield + 63) / 64)), dim3(64), 0,
                           stream, v, 1);
      break;
    case QP_ADI_SWEEP_Y_CARRY:
      QP_LAUNCH_STREAMED(v.d.stream, v.compact, rect_y_kernel, 1, dim3(tiles), dim3(64), 0, stream, v, (const double*)w, w);
      break;
    case QP_ADI_SWEEP_Y_EXIT:
      QP_REQUIRE(u != nullptr, "u is NULL");
      QP_LAUNCH_STREAMED(v.d.stream, v.compact, rect_y_kernel, 2, dim3(tiles), dim3(64), 0, stream, v, (const double*)w, u);
      break;
    default:
      set_error("qp_adi_rect_phase: unknown phase %d", phase);
      return QP_ERR_INVALID_ARGUMENT;
  }
  return check_launch("qp_adi_rect_phase");
}

int qp_adi_rect_steps(qp_adi_rect_plan* plan, double* u, int32_t nsteps, void* stream) {
  QP_REQUIRE(plan && u, "plan and u must be non-NULL");
  QP_REQUIRE(nsteps >= 1, "nsteps must be >= 1");
  QP_REQUIRE(!plan->decomposed, "a decomposed plan needs halo exchanges between phases: drive it with qp_adi_rect_phase");
  int rc = qp_adi_rect_phase(plan, QP_ADI_ENTRY, u, stream);
  for (int s = 0; s < nsteps && rc == QP_OK; ++s) {
    rc = qp_adi_rect_phase(plan, QP_ADI_REDUCED_X, u, stream);
    if (rc == QP_OK) rc = qp_adi_rect_phase(plan, QP_ADI_SWEEP_X, u, stream);
    if (rc == QP_OK) rc = qp_adi_rect_phase(plan, QP_ADI_REDUCED_Y, u, stream);
    if (rc == QP_OK) rc = qp_adi_rect_phase(plan, s + 1 < nsteps ? QP_ADI_SWEEP_Y_CARRY : QP_ADI_SWEEP_Y_EXIT, u, stream);
  }
  return rc;
}

// x <- (I - a Ly)^-1 (I - a Lx)^-1 x in place, per field: the ADI factorisation M of the Crank-Nicolson matrix
// A = I - a (Lx + Ly), applied as a preconditioner (no explicit operators, no boundary sources).  Three passes.
int qp_adi_rect_solve(qp_adi_rect_plan* plan, double* x, void* stream_) {
  QP_REQUIRE(plan && x, "plan and x must be non-NULL");
  QP_REQUIRE(!plan->decomposed, "qp_adi_rect_solve is not available on decomposed plans");
  using namespace qp;
  hipStream_t stream = (hipStream_t)stream_;
  if (plan->fine) {
    const FineView& f = plan->fview;
    const unsigned ftiles = (unsigned)((long)f.nfield * (f.ny / 64) * f.px);
    QP_LAUNCH_FINE(f.stream, fine_y_kernel, 3, dim3(ftiles), dim3(64), 0, stream, f, (const double*)x, x);
    QP_LAUNCH_FINE(f.stream, fine_x_kernel, false, dim3(ftiles), dim3(64), 0, stream, f, x);
    QP_LAUNCH_FINE(f.stream, fine_y_kernel, 2, dim3(ftiles), dim3(64), 0, stream, f, (const double*)x, x);
    return check_launch("qp_adi_rect_solve (fine tiles)");
  }
  const RectView& v = plan->view;
  const unsigned tiles = (unsigned)((long)v.d.nfield * v.d.py * v.d.px);
  QP_LAUNCH_STREAMED(v.d.stream, v.compact, rect_y_kernel, 3, dim3(tiles), dim3(64), 0, stream, v, (const double*)x, x);
  int rc = qp_adi_rect_phase(plan, QP_ADI_REDUCED_X, x, stream_);
  if (rc) return rc;
  QP_LAUNCH_STREAMED(v.d.stream, v.compact, rect_x_kernel, false, dim3(tiles), dim3(64), 0, stream, v, x);
  rc = qp_adi_rect_phase(plan, QP_ADI_REDUCED_Y, x, stream_);
  if (rc) return rc;
  QP_LAUNCH_STREAMED(v.d.stream, v.compact, rect_y_kernel, 2, dim3(tiles), dim3(64), 0, stream, v, (const double*)x, x);
  return check_launch("qp_adi_rect_solve");
}

// out = c0 u + cx (a Lx u) + cy (a Ly u) + cs a S + cr rin with the plan's operator (see rect_combine_kernel); with
// norm_out non-NULL also norm_out[0] = max |out| (workspace: qp_pauli_workspace_bytes() bytes).
int qp_adi_rect_combine(qp_adi_rect_plan* plan, const double* u, const double* rin, double* out, double c0, double cx,
                        double cy, double cs, double cr, void* workspace, double* norm_out, void* stream_) {
  QP_REQUIRE(plan && u, "plan and u must be non-NULL");
  QP_REQUIRE(out || norm_out, "out may be NULL only when the norm is wanted (norm_out)");
  QP_REQUIRE(cr == 0.0 || rin, "rin is required when cr != 0");
  QP_REQUIRE(!plan->decomposed, "qp_adi_rect_combine is not available on decomposed plans");
  QP_REQUIRE((norm_out == nullptr) || workspace, "workspace is required with norm_out");
  using namespace qp;
  hipStream_t stream = (hipStream_t)stream_;
  const RectView& v = plan->view;
  // one band of 8 rows (row lengths that are multiples of 4) or one row per block and trip
  // rows per band: 8 (two halo rows per band) on large grids; 4 on small ones, where 8-row bands leave most CUs without a
  // block (1024^2: 128 blocks).  16-row bands were measured at 4096^2: no gain (the kernel is not bound by the halo rows).
  const bool small = (long)v.d.nfield * plan->ncell < (1L << 22);
  const int rb = small ? 4 : 8;
  long blocks = (v.d.nx & 3) == 0 ? (long)v.d.nfield * ((v.d.ny + rb - 1) / rb) * ((v.d.nx + 1023) / 1024)
                                  : (long)v.d.nfield * v.d.ny;
  if (blocks > 1024) blocks = 1024;         // = the partial slots of the reduction workspace
  QP_REQUIRE((v.d.nx & 3) != 0 || ((uintptr_t)u | (uintptr_t)out | (uintptr_t)rin) % 16 == 0,
             "u, rin, out must be 16-byte aligned when nx is a multiple of 4");
  RectSides g{plan->bc_diag[0], plan->bc_diag[1], plan->bc_diag[2], plan->bc_diag[3],
              plan->bc_src[0], plan->bc_src[1], plan->bc_src[2], plan->bc_src[3]};
  if (v.d.nx == 1) { g.dr = 0.0; g.sr = 0.0; }   // one column: rect_side_terms folds both x-faces into the "left" slot
  if (v.d.ny == 1) { g.dd = 0.0; g.sd = 0.0; }
  double* part = (out == nullptr || norm_out) ? (double*)workspace : nullptr;
#define QP_COMBINE(STORE, RB)                                                                                              \
  hipLaunchKernelGGL((rect_combine_kernel<STORE, RB>), dim3((unsigned)blocks), dim3(256), 0, stream, v.d.ny, v.d.nx,      \
                     v.d.nfield, (const double*)plan->d_alpha, g, u, rin, out, c0, cx, cy, cs, cr, part)
  if (out) { if (small) QP_COMBINE(true, 4); else QP_COMBINE(true, 8); }
  else { if (small) QP_COMBINE(false, 4); else QP_COMBINE(false, 8); }
#undef QP_COMBINE
  if (norm_out) absmax_finish((const double*)workspace, (int)blocks, norm_out, stream);
  return check_launch("qp_adi_rect_combine");
}

// Boundary rows of the reduced right-hand sides <-> contiguous [nfield][nlines] buffers (domain decomposition).
// dir 0: lines are rows, neighbours are left (side 0) / right (side 1); dir 1: columns, up (0) / down (1).
// op 0 (pack): the row the neighbour on `side` needs (side 0: y[0] of my first chunk; side 1: y[last] of my last chunk).
// op 1 (unpack): store the neighbour's row into my halo slot on `side`.
int qp_adi_rect_iface_halo(qp_adi_rect_plan* plan, int32_t dir, int32_t side, int32_t op, double* buf, void* stream) {
  QP_REQUIRE(plan && buf, "plan and buf must be non-NULL");
  QP_REQUIRE((dir == 0 || dir == 1) && (side == 0 || side == 1) && (op == 0 || op == 1), "dir, side, op must be 0 or 1");
  // only block plans of a decomposed grid own interface rows (Peaceman-Rachford plans on fine tiles allocate none)
  QP_REQUIRE(plan->decomposed && plan->pr_scale == 0.0, "only plans of qp_adi_rect_plan_create_block on a decomposed grid exchange interface rows");
  const qp::RectView& v = plan->view;
  const size_t nlines = dir == 0 ? v.d.ny : v.d.nx;
  const int P = dir == 0 ? v.d.px : v.d.py;
  const size_t rows = 2 * (size_t)P + 2;
  size_t row;
  if (op == 0) row = side == 0 ? 1 : 2 * (size_t)P;
  else row = side == 0 ? 0 : 2 * (size_t)P + 1;
  double* strided = plan->d_iface[dir] + row * nlines;
  const size_t w = nlines * sizeof(double), pitch = rows * nlines * sizeof(double);
  hipError_t e = op == 0
                     ? hipMemcpy2DAsync(buf, w, strided, pitch, w, v.d.nfield, hipMemcpyDeviceToDevice, (hipStream_t)stream)
                     : hipMemcpy2DAsync(strided, pitch, buf, w, w, v.d.nfield, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) {
    qp::set_error("qp_adi_rect_iface_halo: %s", hipGetErrorString(e));
    return QP_ERR_LAUNCH;
  }
  return QP_OK;
}

// Field rows just above (side 0) / below (side 1) the local block, [nfield][nx], for the entry pass of a decomposed plan.
int qp_adi_rect_set_field_halo(qp_adi_rect_plan* plan, int32_t side, const double* rows, void* stream) {
  QP_REQUIRE(plan && rows && (side == 0 || side == 1), "bad arguments");
  QP_REQUIRE(plan->decomposed && plan->pr_scale == 0.0, "only plans of qp_adi_rect_plan_create_block on a decomposed grid take field halo rows");
  const qp::RectView& v = plan->view;
  hipError_t e = hipMemcpyAsync(plan->d_uhalo[side], rows, (size_t)v.d.nfield * v.d.nx * sizeof(double),
                                hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) {
    qp::set_error("qp_adi_rect_set_field_halo: %s", hipGetErrorString(e));
    return QP_ERR_LAUNCH;
  }
  return QP_OK;
}

}  // extern "C"
