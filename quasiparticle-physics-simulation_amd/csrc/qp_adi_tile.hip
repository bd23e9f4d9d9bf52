// Tiled CN-ADI path for masked grids (any mask, any per-face boundary condition, one diffusivity per field).
//
// Same scheme as qp_adi_rect.hip - 64 x 64 tiles, one wave per tile, partition method along each grid line, the
// carried right-hand side read and written once per sweep - but the lines of a masked grid are all different, so
// nothing can be tabulated per chunk position.  Instead every cell carries a 16-bit code
//     bit 0..3  links to the x-, x+, y-, y+ neighbour     bit 4  cell is inside the mask
//     bit 5..15 index into a small table of boundary terms (ex, ey, sx + sy); 0 = no boundary face
// stored twice (row-major for x-direction work, column-major for y-direction work: a lane always reads the 128
// contiguous bytes of its own line), and the tiles are sorted into three classes at plan creation:
//     empty    no cell inside the mask: never touched (the carried planes stay 0 there)
//     clean    all 4096 cells inside, all links present, no boundary face: the interior-chunk tables of the
//              rectangle path apply (wave-uniform scalars), same cost as the rectangle kernels
//     general  everything else: pivots, spikes and the explicit operator are computed per lane from the codes
//              (one fp64 division per cell and pass; these kernels keep the Thomas coefficients of a whole line
//              in registers and run at one wave per SIMD)
// Clean and general tiles are launched as separate kernels over their tile lists.
//
// Coupling between the chunks of a line: as in the rectangle path's "decoupled" regime.  With
//     x_p = y_p + wm_first E_{p-1} g_p + wp_last F_{p+1} h_p,        g_p, h_p = first / last column of A_p^-1
// the far corners g_p[last], h_p[first] are checked at plan creation to be below 1e-22 (else the plan is refused and
// the caller falls back to the per-line kernels), so each interface is the 2 x 2 system
//     E_p - s_p F_{p+1} = y_p[last],   F_{p+1} - t_{p+1} E_p = y_{p+1}[first],
//     s_p = wp_last h_p[last],  t_p = wm_first g_p[first]
// whose coefficients are per (field, line, chunk) constants computed once on the device (`coef`), and whose
// right-hand sides (`iface`) are produced by the preceding sweep kernel as a by-product.
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "qp_tile_common.h"

namespace qp {

constexpr int kCodeIdxShift = 5;
constexpr int kMaxBc = 1024;            // boundary-term table entries (24 B each, staged in LDS by general tiles)

struct TileView {
  int ny, nx, nfield;
  int py, px;                   // tiles per column / per row
  int pny, pnx;                 // padded extents of the code planes (py * 64, px * 64)
  const double* alpha;          // [nfield] r D
  const double* tab;            // [nfield][T_NSLOT][TS] interior-chunk tables (clean tiles)
  int compact;                  // every field's table has the compact form (qp_tile_common.h): clean kernels <..., true>
  const double* ctab;           // [nfield][2 parts][CT_PART] compact interior-chunk tables (valid when `compact`)
  const uint16_t* code_r;       // [pny][pnx]
  const uint16_t* code_c;       // [pnx][pny]
  const double* bct;            // [nbc][3] (ex, ey, sx + sy)
  int nbc;
  double* iface[2];             // per dir [nfield][2P+2][nlines]: row 2p+1 = y_p[first], row 2p+2 = y_p[last]
  double* coef[2];              // same shape: row 2p+1 = t_p, row 2p+2 = s_p
  const int32_t* tiles[2];      // [0] clean, [1] general: (has_bc << 30) | (ty << 16) | tx
  int ntiles[2];
  // variable-D plans (var != 0): per field, padded [pny][pnx] row-major (x) and [pnx][pny] column-major (y)
  int var;
  int stream;                   // non-temporal plane accesses (working set beyond the Infinity Cache)
  const double* wx;             // weight of the face between (j, i) and (j, i+1)
  const double* wy;             // weight of the face between (j, i) and (j+1, i)
  const double* rdx;            // r D of the cell, row-major
  const double* rdy;            // r D of the cell, column-major
};

template <int DIR> struct Bits {
  static constexpr unsigned LM = DIR == 0 ? QP_FLAG_LINK_XM : QP_FLAG_LINK_YM;
  static constexpr unsigned LP = DIR == 0 ? QP_FLAG_LINK_XP : QP_FLAG_LINK_YP;
};

// code of cell k of the lane's line (two codes per dword)
__device__ __forceinline__ unsigned code_at(const unsigned (&cw)[TS / 2], int k) {
  return (k & 1) ? (cw[k >> 1] >> 16) : (cw[k >> 1] & 0xffffu);
}

__device__ __forceinline__ void load_codes(const uint16_t* line, unsigned (&cw)[TS / 2]) {
  const uint4* p = reinterpret_cast<const uint4*>(line);      // 128-byte aligned by construction
#pragma unroll
  for (int q = 0; q < TS / 8; ++q) {
    const uint4 w = p[q];
    cw[4 * q] = w.x; cw[4 * q + 1] = w.y; cw[4 * q + 2] = w.z; cw[4 * q + 3] = w.w;
  }
}

// Makes every code word "depend" on `x`: decoding them (pure arithmetic, no order of its own inside the unrolled basic
// block) cannot be hoisted above the point where x is produced.  Without it the decoding of the explicit pass, 3 x 64
// doubles, is scheduled into the solve that precedes it, where 256 registers are already live.
__device__ __forceinline__ void codes_after(unsigned (&cw)[TS / 2], double& x) {
#pragma unroll
  for (int q = 0; q < TS / 2; ++q) asm volatile("" : "+v"(cw[q]), "+v"(x));
}

// 1 / x for a pivot (1 <= x, far from overflow): v_rcp_f64 + two Newton steps, 5 dependent instructions instead of the
// ~12 of the IEEE division sequence; the general kernels are issue-bound and pivots are their inner loop.
__device__ __forceinline__ double pivot_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

// Per-cell coefficients of the lane's line, uniform diffusivity: decoded from the codes.
//   wm(k) / wp(k)  weights of the faces towards k-1 / k+1 (0 without a link)      bd(k)  boundary diagonal term
//   src(k)         boundary source of both directions
template <int DIR>
struct UniCoef {
  const unsigned (&cw)[TS / 2];
  double a;
  const double* bct;
  __device__ __forceinline__ double wm(int k) const { return (code_at(cw, k) & Bits<DIR>::LM) ? a : 0.0; }
  __device__ __forceinline__ double wp(int k) const { return (code_at(cw, k) & Bits<DIR>::LP) ? a : 0.0; }
  __device__ __forceinline__ double bd(int k) const { return a * bct[3 * (code_at(cw, k) >> kCodeIdxShift) + DIR]; }
  __device__ __forceinline__ double src(int k) const { return a * bct[3 * (code_at(cw, k) >> kCodeIdxShift) + 2]; }
  __device__ __forceinline__ void tie(double&) const {}
};

// Spatially varying diffusivity: face weights r * harmonic mean come from a per-field plane (w[k] = face k | k+1 of the
// lane's line, wfirst = face towards the previous chunk).  Tiles without boundary faces (almost all) hold the line's
// weights in registers.
template <int DIR>
struct VarCoef {
  const double (&w)[TS];
  double wfirst;
  __device__ __forceinline__ double wm(int k) const { return k > 0 ? w[k - 1] : wfirst; }
  __device__ __forceinline__ double wp(int k) const { return w[k]; }
  __device__ __forceinline__ double bd(int) const { return 0.0; }
  __device__ __forceinline__ double src(int) const { return 0.0; }
  __device__ __forceinline__ void tie(double&) const {}
};

// Variable-D tiles WITH boundary faces: boundary terms scale with r D of the cell (solver.py:296-311), so the cell's r D is
// needed next to its weights.  Holding both lines in registers does not fit beside e and c; these (rare) tiles stream
// weights and r D from the lane's contiguous lines cell by cell instead.  `zero` is an opaque 0 added to every index and
// re-defined together with the recurrence value at each cell, which keeps the loads of later cells from being hoisted.
template <int DIR>
struct StreamCoef {
  const double* wl;
  const double* rd;
  const unsigned (&cw)[TS / 2];
  const double* bct;
  double wfirst;
  mutable unsigned zero;
  __device__ __forceinline__ double wm(int k) const { return k > 0 ? wl[k - 1 + zero] : wfirst; }
  __device__ __forceinline__ double wp(int k) const { return wl[k + zero]; }
  __device__ __forceinline__ double bd(int k) const {
    return rd[k + zero] * bct[3 * (code_at(cw, k) >> kCodeIdxShift) + DIR];
  }
  __device__ __forceinline__ double src(int k) const {
    return rd[k + zero] * bct[3 * (code_at(cw, k) >> kCodeIdxShift) + 2];
  }
  __device__ __forceinline__ void tie(double& chain) const { asm volatile("" : "+v"(zero), "+v"(chain)); }
};

// (I - L_w) x = e along the lane's chunk with neighbour values gl / gr beyond its ends.
template <class Coef>
__device__ __forceinline__ void solve_general(double (&e)[TS], const Coef& co, double gl, double gr) {
  double c[TS];
  double wm = co.wm(0);
  e[0] = fma(wm, gl, e[0]);
  double cprev = 0.0, dprev = 0.0;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    co.tie(cprev);
    const double wp = co.wp(k);
    const double bd = co.bd(k);
    if (k == TS - 1) e[k] = fma(wp, gr, e[k]);
    const double inv = pivot_rcp(fma(-wm, cprev, 1.0 + wm + wp + bd));
    cprev = wp * inv;
    c[k] = cprev;
    dprev = fma(wm, dprev, e[k]) * inv;
    e[k] = dprev;
    wm = wp;
  }
  double x = 0.0;
#pragma unroll
  for (int k = TS - 1; k >= 0; --k) {
    x = fma(c[k], x, e[k]);
    e[k] = x;
  }
}

// e <- (I + L_w) e + boundary sources along the lane's chunk
template <class Coef>
__device__ __forceinline__ void explicit_general(double (&e)[TS], const Coef& co, double gl, double gr) {
  double prev = gl;
  double wm = co.wm(0);
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    co.tie(prev);
    const double wp = co.wp(k);
    const double bd = co.bd(k), src = co.src(k);
    const double cur = e[k];
    const double nxt = (k + 1 < TS) ? e[k + 1] : gr;
    e[k] = fma(wm, prev - cur, fma(wp, nxt - cur, fma(-bd, cur, cur + src)));
    prev = cur;
    wm = wp;
  }
}

// First and last entry of A_p^-1 e (chunk-local system, couplings to the neighbouring chunks dropped), plus the
// interface coefficients s = wp_last h[last], t = wm_first g[first] (by-products of the two eliminations).
template <class Coef>
__device__ __forceinline__ void ends_general(const double (&e)[TS], const Coef& co, double& yf, double& yl, double& s,
                                             double& t) {
  double wm = co.wm(0);
  double wq = co.wp(TS - 1);
  double cf = 0.0, df = 0.0, cb = 0.0, db = 0.0;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    co.tie(cf);
    {  // forward elimination, cell k
      const double wp = co.wp(k);
      const double inv = pivot_rcp(fma(-wm, cf, 1.0 + wm + wp + co.bd(k)));
      cf = wp * inv;
      df = fma(wm, df, e[k]) * inv;
      wm = wp;
    }
    {  // backward elimination, cell TS-1-k
      const int kb = TS - 1 - k;
      const double wl = co.wm(kb);
      const double inv = pivot_rcp(fma(-wq, cb, 1.0 + wl + wq + co.bd(kb)));
      cb = wl * inv;
      db = fma(wq, db, e[kb]) * inv;
      wq = wl;
    }
  }
  yl = df;
  yf = db;
  s = cf;
  t = cb;
}

// Neighbour values of the solved line just outside chunk p of `line`: gl = E_{p-1}, gr = F_{p+1}.  The eight loads are issued
// by `tile_ghosts_prefetch` at the very start of a kernel (always in bounds: the interface arrays have 2P + 2 rows), the
// arithmetic happens where the values are needed - see ghost_prefetch in qp_adi_rect.hip.
struct TileGhostRaw {
  double y[4], c[4];
};

template <int DIR>
__device__ __forceinline__ TileGhostRaw tile_ghosts_prefetch(const TileView& v, int b, int p, long line) {
  const int P = DIR == 0 ? v.px : v.py;
  const long nl = DIR == 0 ? v.ny : v.nx;
  if (line > nl - 1) line = nl - 1;
  const long base = (long)b * (2 * P + 2) * nl + line;
  const double* ir = v.iface[DIR] + base;
  const double* cf = v.coef[DIR] + base;
  TileGhostRaw g;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    g.y[q] = ir[(long)(2 * p + q) * nl];
    g.c[q] = cf[(long)(2 * p + q) * nl];
  }
  return g;
}

template <int DIR>
__device__ __forceinline__ void tile_ghosts_finish(const TileView& v, int p, bool on, const TileGhostRaw& g, double& gl,
                                                   double& gr) {
  const int P = DIR == 0 ? v.px : v.py;
  gl = 0.0;
  gr = 0.0;
  if (p > 0) gl = fma(g.c[0], g.y[1], g.y[0]) / fma(-g.c[0], g.c[1], 1.0);          // (yl + s yf) / (1 - s t)
  if (p < P - 1) gr = fma(g.c[3], g.y[2], g.y[3]) / fma(-g.c[2], g.c[3], 1.0);      // (yf + t yl) / (1 - s t)
  if (!on) gl = gr = 0.0;
}

constexpr int kTileHasBc = 1 << 30;      // tile-list entry: (has_bc << 30) | (ty << 16) | tx

// CLS 0: clean tile, 1: general tile with uniform D, 2: tile of a variable-D plan
// STREAM >= 0: stream mode known at compile time (clean tiles: the bandwidth-bound kernels keep branch-free accesses);
// STREAM < 0: taken from the plan at run time (general tiles are issue-bound, one variant suffices)
template <int CLS, int STREAM>
__device__ __forceinline__ TileCoord tile_of(const TileView& v, int id, bool& has_bc) {
  TileCoord t;
  t.b = id % v.nfield;                          // fields of one tile are neighbours in launch order: shared codes hit L2
  // the list entry is wave-uniform; say so, or every row address of the tile becomes a per-lane 64-bit value
  const int packed = __builtin_amdgcn_readfirstlane(v.tiles[CLS == 0 ? 0 : 1][id / v.nfield]);
  has_bc = (packed & kTileHasBc) != 0;
  t.ty = (packed >> 16) & 0x3fff;
  t.tx = packed & 0xffff;
  t.j0 = t.ty * TS;
  t.i0 = t.tx * TS;
  t.nr = min(TS, v.ny - t.j0);
  t.nc = min(TS, v.nx - t.i0);
  t.stream = STREAM >= 0 ? STREAM : v.stream;
  return t;
}

__device__ __forceinline__ void stage_bct(const TileView& v, double* dst, int lane) {
  for (int q = lane; q < 3 * v.nbc; q += 64) dst[q] = v.bct[q];
  __syncthreads();
}

__device__ __forceinline__ void load_line(const double* line, double (&w)[TS]) {
  const double2* p = reinterpret_cast<const double2*>(line);      // 512-byte aligned by construction
#pragma unroll
  for (int q = 0; q < TS / 2; ++q) {
    const double2 v2 = p[q];
    w[2 * q] = v2.x;
    w[2 * q + 1] = v2.y;
  }
}

// Lane's line of the row-major (x-direction work, lane = row) / column-major (y-direction work, lane = column) planes.
template <int DIR>
__device__ __forceinline__ long line_offset(const TileView& v, const TileCoord& t, int lane) {
  return DIR == 0 ? (long)(t.j0 + lane) * v.pnx + t.i0 : (long)(t.i0 + lane) * v.pny + t.j0;
}

// solve [+ explicit operator] along DIR on a general / variable-D tile
template <int CLS, int DIR, bool SOLVE, bool EXPLICIT>
__device__ __forceinline__ void line_work(const TileView& v, const TileCoord& t, int lane, bool has_bc, const double* bct,
                                          double a, double (&e)[TS], double gl, double gr) {
  const long off = line_offset<DIR>(v, t, lane);
  unsigned cw[TS / 2];
  load_codes((DIR == 0 ? v.code_r : v.code_c) + off, cw);
  if (CLS == 1) {
    const UniCoef<DIR> co{cw, a, bct};
    if (SOLVE) solve_general(e, co, gl, gr);
    if (SOLVE && EXPLICIT) codes_after(cw, e[0]);
    if (EXPLICIT) explicit_general(e, co, gl, gr);
  } else {
    const long poff = (long)t.b * v.pny * v.pnx + off;
    const double* wl = (DIR == 0 ? v.wx : v.wy) + poff;
    const double wfirst = (DIR == 0 ? t.tx : t.ty) > 0 ? wl[-1] : 0.0;
    if (has_bc) {     // wave-uniform
      const StreamCoef<DIR> co{wl, (DIR == 0 ? v.rdx : v.rdy) + poff, cw, bct, wfirst, 0u};
      if (SOLVE) solve_general(e, co, gl, gr);
      if (EXPLICIT) explicit_general(e, co, gl, gr);
    } else {
      double w[TS];
      load_line(wl, w);
      const VarCoef<DIR> co{w, wfirst};
      if (SOLVE) solve_general(e, co, gl, gr);
      if (EXPLICIT) explicit_general(e, co, gl, gr);
    }
  }
}

// chunk-local eliminations along DIR: first / last entry of A_p^-1 e and the interface coefficients
template <int CLS, int DIR>
__device__ __forceinline__ void line_ends(const TileView& v, const TileCoord& t, int lane, bool has_bc, const double* bct,
                                          double a, const double (&e)[TS], double& yf, double& yl, double& s, double& tt,
                                          double& wm0, double& wp1) {
  const long off = line_offset<DIR>(v, t, lane);
  unsigned cw[TS / 2];
  load_codes((DIR == 0 ? v.code_r : v.code_c) + off, cw);
  if (CLS != 2) {
    const UniCoef<DIR> co{cw, a, bct};
    ends_general(e, co, yf, yl, s, tt);
    wm0 = co.wm(0);
    wp1 = co.wp(TS - 1);
  } else {
    const long poff = (long)t.b * v.pny * v.pnx + off;
    const double* wl = (DIR == 0 ? v.wx : v.wy) + poff;
    const double wfirst = (DIR == 0 ? t.tx : t.ty) > 0 ? wl[-1] : 0.0;
    if (has_bc) {
      const StreamCoef<DIR> co{wl, (DIR == 0 ? v.rdx : v.rdy) + poff, cw, bct, wfirst, 0u};
      ends_general(e, co, yf, yl, s, tt);
    } else {
      double w[TS];
      load_line(wl, w);
      const VarCoef<DIR> co{w, wfirst};
      ends_general(e, co, yf, yl, s, tt);
    }
    wm0 = wfirst;
    wp1 = wl[TS - 1];
  }
}

// coefficient source of the clean tiles: the field's interior-chunk table in registers, or a part of its compact form
template <bool COMPACT, int PART>
__device__ __forceinline__ auto clean_coefs(const TileView& v, int b, int lane, bool clean) {
  if constexpr (COMPACT) {
    return CoefCompact{as_const(v.ctab + ((long)b * 2 + PART) * CT_PART)};
  } else {
    CoefFull c;
    if (clean) {
      if (PART == 0) load_tab<T_W, T_EW - T_W>(v.tab + (long)b * T_NSLOT * TS, lane, c.r);
      else load_tab<T_EW, T_NSLOT - T_EW>(v.tab + (long)b * T_NSLOT * TS, lane, c.r);
    }
    return c;
  }
}

// x-kernel: finish the x-solve, [apply (I + a Lx) . + a S], store, chunk-local eliminations along y -> iface[1]
template <int CLS, bool EXPLICIT, int STREAM, bool COMPACT>
__device__ __forceinline__ void tile_x_body(const TileView& v, double* __restrict__ buf, double* lds, double* bct,
                                            int block) {
  const int lane = threadIdx.x;
  bool has_bc;
  const TileCoord t = tile_of<CLS, STREAM>(v, block, has_bc);
  if (CLS != 0) stage_bct(v, bct, lane);
  const long ncell = (long)v.ny * v.nx;
  double* plane = buf + (long)t.b * ncell;
  const double a = as_const(v.alpha)[t.b];
  const TileGhostRaw graw = tile_ghosts_prefetch<0>(v, t.b, t.tx, t.j0 + lane);
  const auto csolve = clean_coefs<COMPACT, 0>(v, t.b, lane, CLS == 0);
  const auto cends = clean_coefs<COMPACT, 1>(v, t.b, lane, CLS == 0);
  double e[TS];
  load_cols(plane, t, v.nx, lane, e);
  if constexpr (COMPACT && CLS == 0) {
    const CoefCompact parts[2] = {csolve, cends};
    warm_scalar_cache(parts);
  }
  transpose64(e, lds, lane);
  double gl, gr;
  tile_ghosts_finish<0>(v, t.tx, lane < t.nr, graw, gl, gr);
  if (CLS != 0) {
    line_work<CLS, 0, true, EXPLICIT>(v, t, lane, has_bc, bct, a, e, gl, gr);
  } else {
    e[0] = fma(a, gl, e[0]);
    e[TS - 1] = fma(a, gr, e[TS - 1]);
    thomas64(e, csolve);
    if (EXPLICIT) explicit64(e, gl, gr, csolve, 0.0);
  }
  transpose64(e, lds, lane);
  store_cols(plane, t, v.nx, lane, e);
  double yf, yl;
  if (CLS != 0) {
    double s, tt, w0, w1;
    line_ends<CLS, 1>(v, t, lane, has_bc, bct, a, e, yf, yl, s, tt, w0, w1);
  } else {
    ends64(e, cends, yf, yl);
  }
  if (lane < t.nc) {
    double* ir = v.iface[1] + (long)t.b * (2 * v.py + 2) * v.nx;
    ir[(long)(2 * t.ty + 1) * v.nx + t.i0 + lane] = yf;
    ir[(long)(2 * t.ty + 2) * v.nx + t.i0 + lane] = yl;
  }
}

template <int CLS, bool EXPLICIT, int STREAM, bool COMPACT = false>
__global__ void __launch_bounds__(64) tile_x_kernel(TileView v, double* __restrict__ buf) {
  __shared__ double lds[LDS_DOUBLES];
  extern __shared__ double bct[];
  tile_x_body<CLS, EXPLICIT, STREAM, COMPACT>(v, buf, lds, bct, blockIdx.x);
}

// One launch for both classes: the general tiles first (they take longest), the clean tiles behind them.  For plans with
// FEW general tiles (a ring in 4096^2: 328 single-wave blocks for 1024 SIMDs) two launches in a row leave the chip
// two-thirds idle for the whole general launch; merged, the general path's register footprint (one wave per SIMD) applies
// to the clean tiles as well, which costs them less than the idle launch did (ring in 4096^2: 55 -> 45 us per sweep).
template <bool EXPLICIT>
__global__ void __launch_bounds__(64) tile_x_merged_kernel(TileView v, double* __restrict__ buf) {
  __shared__ double lds[LDS_DOUBLES];
  extern __shared__ double bct[];
  const int ngen = v.ntiles[1] * v.nfield;
  if ((int)blockIdx.x < ngen) tile_x_body<1, EXPLICIT, -1, false>(v, buf, lds, bct, blockIdx.x);
  else tile_x_body<0, EXPLICIT, 0, true>(v, buf, lds, bct, blockIdx.x - ngen);
}

// y-kernel.  MODE 0 (entry): src = u -> rhs1 = (I + a Ly) u + a S;  MODE 1 (carry): y-solve, rhs1' of the next step;
//            MODE 2 (exit): y-solve, dst = u';  MODE 3 (reduce): only the x-eliminations of src (nothing stored).
//            MODE 0, 1, 3 end with the chunk-local eliminations along x -> iface[0]
template <int CLS, int MODE, int STREAM, bool COMPACT>
__device__ __forceinline__ void tile_y_body(const TileView& v, const double* src, double* dst, double* lds, double* bct,
                                            int block) {   // src may alias dst
  const int lane = threadIdx.x;
  bool has_bc;
  const TileCoord t = tile_of<CLS, STREAM>(v, block, has_bc);
  if (CLS != 0) stage_bct(v, bct, lane);
  const long ncell = (long)v.ny * v.nx;
  const double* splane = src + (long)t.b * ncell;
  double* dplane = dst + (long)t.b * ncell;
  const double a = as_const(v.alpha)[t.b];
  const int col = t.i0 + lane;
  const bool col_on = lane < t.nc;
  TileGhostRaw graw;
  if (MODE == 1 || MODE == 2) graw = tile_ghosts_prefetch<1>(v, t.b, t.ty, col);
  const auto csolve = clean_coefs<COMPACT, 0>(v, t.b, lane, CLS == 0 && MODE != 3);
  const auto cends = clean_coefs<COMPACT, 1>(v, t.b, lane, CLS == 0 && MODE != 2);
  double gu = 0.0, gd = 0.0;
  if (MODE == 0) {
    if (col_on && t.ty > 0) gu = splane[(long)(t.j0 - 1) * v.nx + col];
    if (col_on && t.j0 + TS < v.ny) gd = splane[(long)(t.j0 + TS) * v.nx + col];
  }
  double e[TS];
  load_cols(splane, t, v.nx, lane, e);
  if constexpr (COMPACT && CLS == 0) {
    const CoefCompact parts[2] = {csolve, cends};
    warm_scalar_cache(parts);
  }
  if (MODE == 1 || MODE == 2) tile_ghosts_finish<1>(v, t.ty, col_on, graw, gu, gd);
  if (CLS != 0) {
    if (MODE == 0) line_work<CLS, 1, false, true>(v, t, lane, has_bc, bct, a, e, gu, gd);
    if (MODE == 1) line_work<CLS, 1, true, true>(v, t, lane, has_bc, bct, a, e, gu, gd);
    if (MODE == 2) line_work<CLS, 1, true, false>(v, t, lane, has_bc, bct, a, e, gu, gd);
  } else {
    if (MODE == 1 || MODE == 2) {
      e[0] = fma(a, gu, e[0]);
      e[TS - 1] = fma(a, gd, e[TS - 1]);
      thomas64(e, csolve);
    }
    if (MODE == 0 || MODE == 1) explicit64(e, gu, gd, csolve, 0.0);
  }
  if (MODE != 3) store_cols(dplane, t, v.nx, lane, e);
  if (MODE == 2) return;
  transpose64(e, lds, lane);
  double yf, yl;
  if (CLS != 0) {
    double s, tt, w0, w1;
    line_ends<CLS, 0>(v, t, lane, has_bc, bct, a, e, yf, yl, s, tt, w0, w1);
  } else {
    ends64(e, cends, yf, yl);
  }
  if (lane < t.nr) {
    double* ir = v.iface[0] + (long)t.b * (2 * v.px + 2) * v.ny;
    ir[(long)(2 * t.tx + 1) * v.ny + t.j0 + lane] = yf;
    ir[(long)(2 * t.tx + 2) * v.ny + t.j0 + lane] = yl;
  }
}

template <int CLS, int MODE, int STREAM, bool COMPACT = false>
__global__ void __launch_bounds__(64) tile_y_kernel(TileView v, const double* src, double* dst) {
  __shared__ double lds[LDS_DOUBLES];
  extern __shared__ double bct[];
  tile_y_body<CLS, MODE, STREAM, COMPACT>(v, src, dst, lds, bct, blockIdx.x);
}

template <int MODE>
__global__ void __launch_bounds__(64) tile_y_merged_kernel(TileView v, const double* src, double* dst) {   // see tile_x_merged_kernel
  __shared__ double lds[LDS_DOUBLES];
  extern __shared__ double bct[];
  const int ngen = v.ntiles[1] * v.nfield;
  if ((int)blockIdx.x < ngen) tile_y_body<1, MODE, -1, false>(v, src, dst, lds, bct, blockIdx.x);
  else tile_y_body<0, MODE, 0, true>(v, src, dst, lds, bct, blockIdx.x - ngen);
}

// Plan creation: interface coefficients (s, t) of every chunk of every non-empty tile in both directions, and the
// largest far coupling (atomic max over the bit pattern of a non-negative double).  CLS 1 (codes; also right for clean
// tiles) or 2 (variable D); `list` selects the tile list.
template <int CLS>
__global__ void __launch_bounds__(64) tile_setup_kernel(TileView v, int list, unsigned long long* far_bits) {
  extern __shared__ double bct[];
  const int lane = threadIdx.x;
  TileCoord t;
  const int id = blockIdx.x;
  t.b = id % v.nfield;
  const int packed = __builtin_amdgcn_readfirstlane(v.tiles[list][id / v.nfield]);
  const bool has_bc = (packed & kTileHasBc) != 0;
  t.ty = (packed >> 16) & 0x3fff;
  t.tx = packed & 0xffff;
  t.j0 = t.ty * TS;
  t.i0 = t.tx * TS;
  t.nr = min(TS, v.ny - t.j0);
  t.nc = min(TS, v.nx - t.i0);
  t.stream = 0;
  stage_bct(v, bct, lane);
  const double a = v.alpha[t.b];
  double far = 0.0;
  double e[TS];
  double yf, yl, s, tt, yf2, yl2, s2, t2, wm0, wp1;
  // direction x (lane = row)
#pragma unroll
  for (int k = 0; k < TS; ++k) e[k] = k == 0 ? 1.0 : 0.0;
  line_ends<CLS, 0>(v, t, lane, has_bc, bct, a, e, yf, yl, s, tt, wm0, wp1);
  e[0] = 0.0;
  e[TS - 1] = 1.0;
  line_ends<CLS, 0>(v, t, lane, has_bc, bct, a, e, yf2, yl2, s2, t2, wm0, wp1);
  far = fmax(far, fmax(fabs(wm0 * yl), fabs(wp1 * yf2)));      // wm_first g[last], wp_last h[first]
  if (lane < t.nr) {
    double* cf = v.coef[0] + (long)t.b * (2 * v.px + 2) * v.ny + t.j0 + lane;
    cf[(long)(2 * t.tx + 1) * v.ny] = tt;
    cf[(long)(2 * t.tx + 2) * v.ny] = s;
  }
  // direction y (lane = column)
  e[TS - 1] = 0.0;
  e[0] = 1.0;
  line_ends<CLS, 1>(v, t, lane, has_bc, bct, a, e, yf, yl, s, tt, wm0, wp1);
  e[0] = 0.0;
  e[TS - 1] = 1.0;
  line_ends<CLS, 1>(v, t, lane, has_bc, bct, a, e, yf2, yl2, s2, t2, wm0, wp1);
  far = fmax(far, fmax(fabs(wm0 * yl), fabs(wp1 * yf2)));
  if (lane < t.nc) {
    double* cf = v.coef[1] + (long)t.b * (2 * v.py + 2) * v.nx + t.i0 + lane;
    cf[(long)(2 * t.ty + 1) * v.nx] = tt;
    cf[(long)(2 * t.ty + 2) * v.nx] = s;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) far = fmax(far, __shfl_xor(far, off));
  if (lane == 0) atomicMax(far_bits, (unsigned long long)__double_as_longlong(far));
}

// Variable-D plans: face weights r * harmonic mean (solver.py:283) and r D per cell, in the row-major and column-major
// padded layouts the sweeps read; one thread per padded cell and field.
__global__ void __launch_bounds__(256) tile_var_weights_kernel(TileView v, const double* __restrict__ dfield, double r,
                                                               double* wx, double* wy, double* rdx, double* rdy) {
  const long pcell = (long)v.pny * v.pnx;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= pcell * v.nfield) return;
  const int b = (int)(gid / pcell);
  const long q = gid - (long)b * pcell;
  const int j = (int)(q / v.pnx), i = (int)(q - (long)j * v.pnx);
  double d = 0.0, fx = 0.0, fy = 0.0;
  if (j < v.ny && i < v.nx) {
    const unsigned code = v.code_r[q];
    const double* D = dfield + (long)b * v.ny * v.nx;
    const long p = (long)j * v.nx + i;
    if (code & QP_FLAG_ACTIVE) d = D[p];
    if (code & QP_FLAG_LINK_XP) fx = r * face_mean(d, D[p + 1]);
    if (code & QP_FLAG_LINK_YP) fy = r * face_mean(d, D[p + v.nx]);
  }
  const long qc = (long)i * v.pny + j;
  wx[gid] = fx;
  rdx[gid] = r * d;
  wy[(long)b * pcell + qc] = fy;
  rdy[(long)b * pcell + qc] = r * d;
}

int tile_ablation_mask() { return QP_ABL; }

}  // namespace qp

struct qp_adi_tile_plan {
  qp::TileView view;
  std::vector<void*> allocs;
  double* d_work = nullptr;
  long ncell = 0;
  int counts[3] = {0, 0, 0};       // empty, clean, general tiles
  size_t bct_bytes = 0;
  double far = 0.0;
  // Tuning knob QPSIM_TILE_FORK=1: the general tiles (a ring in 4096^2: 328 single-wave blocks for 1024 SIMDs, launched after
  // the clean tiles) go to a side stream, forked from and joined to the caller's stream with events around every sweep.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool merged = false;             // few general tiles: one launch for both classes (tile_x_merged_kernel)
};

extern "C" {

int qp_adi_tile_plan_destroy(qp_adi_tile_plan* plan) {
  if (!plan) return QP_OK;
  for (void* p : plan->allocs) (void)hipFree(p);
  if (plan->side) (void)hipStreamDestroy(plan->side);
  if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
  if (plan->ev_join) (void)hipEventDestroy(plan->ev_join);
  delete plan;
  return QP_OK;
}

static int tile_plan_create(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                            const double* dfield_dev, const uint8_t* flags, const double* ex, const double* ey,
                            const double* sx, const double* sy, qp_adi_tile_plan** out) {
  QP_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  QP_REQUIRE(ny > 0 && nx > 0 && nfield > 0, "ny, nx, nfield must be positive");
  QP_REQUIRE(r > 0.0 && flags && ex && ey && sx && sy, "r must be positive; host arrays non-NULL");
  QP_REQUIRE((dcoef_host != nullptr) != (dfield_dev != nullptr), "exactly one of dcoef / dfield must be given");
  using namespace qp;
  const bool var = dfield_dev != nullptr;
  const int py = (ny + TS - 1) / TS, px = (nx + TS - 1) / TS;
  QP_REQUIRE(py < 16384 && px < 65536, "grid too large for the packed tile index");
  const int pny = py * TS, pnx = px * TS;
  if (!var)
    for (int b = 0; b < nfield; ++b) QP_REQUIRE(dcoef_host[b] >= 0.0, "diffusion coefficients must be >= 0");

  // codes and the boundary-term table
  std::map<std::tuple<double, double, double>, int> bc_index;
  std::vector<double> bct = {0.0, 0.0, 0.0};
  bc_index[std::make_tuple(0.0, 0.0, 0.0)] = 0;
  std::vector<uint16_t> code_r((size_t)pny * pnx, 0), code_c((size_t)pnx * pny, 0);
  for (int j = 0; j < ny; ++j) {
    for (int i = 0; i < nx; ++i) {
      const size_t p = (size_t)j * nx + i;
      const unsigned f = flags[p];
      if (!(f & QP_FLAG_ACTIVE)) continue;
      // links must be mutual and stay inside the mask: the sweeps use wm_k = wp_{k-1}
      const bool ok = (!(f & QP_FLAG_LINK_XM) || (i > 0 && (flags[p - 1] & QP_FLAG_LINK_XP))) &&
                      (!(f & QP_FLAG_LINK_XP) || (i < nx - 1 && (flags[p + 1] & QP_FLAG_LINK_XM))) &&
                      (!(f & QP_FLAG_LINK_YM) || (j > 0 && (flags[p - nx] & QP_FLAG_LINK_YP))) &&
                      (!(f & QP_FLAG_LINK_YP) || (j < ny - 1 && (flags[p + nx] & QP_FLAG_LINK_YM)));
      QP_REQUIRE(ok, "link flags are not mutual");
      int idx = 0;
      const double s = sx[p] + sy[p];
      if (ex[p] != 0.0 || ey[p] != 0.0 || s != 0.0) {
        const auto key = std::make_tuple(ex[p], ey[p], s);
        auto it = bc_index.find(key);
        if (it == bc_index.end()) {
          idx = (int)bc_index.size();
          if (idx >= kMaxBc) {
            set_error("qp_adi_tile_plan_create: more than %d distinct boundary-term combinations", kMaxBc);
            return QP_ERR_UNSUPPORTED;
          }
          bc_index[key] = idx;
          bct.push_back(ex[p]);
          bct.push_back(ey[p]);
          bct.push_back(s);
        } else {
          idx = it->second;
        }
      }
      const uint16_t code = (uint16_t)((f & 31u) | ((unsigned)idx << kCodeIdxShift));
      code_r[(size_t)j * pnx + i] = code;
      code_c[(size_t)i * pny + j] = code;
    }
  }
  // tile classes
  std::vector<int32_t> lists[2];
  int counts[3] = {0, 0, 0};
  for (int ty = 0; ty < py; ++ty) {
    for (int tx = 0; tx < px; ++tx) {
      bool any = false, clean = true, has_bc = false;
      for (int j = ty * TS; j < ty * TS + TS; ++j) {
        const uint16_t* row = &code_r[(size_t)j * pnx + tx * TS];
        for (int k = 0; k < TS; ++k) {
          any = any || row[k] != 0;
          clean = clean && row[k] == 31u;
          has_bc = has_bc || (row[k] >> kCodeIdxShift) != 0;
        }
      }
      if (!any) { counts[0]++; continue; }
      const int cls = (clean && !var) ? 0 : 1;      // variable D: no tile shares coefficients with another
      counts[1 + cls]++;
      lists[cls].push_back((has_bc ? kTileHasBc : 0) | (ty << 16) | tx);
    }
  }
  // interior-chunk tables of the rectangle path (chunk 1 of a 3-chunk line; end faces irrelevant)
  std::vector<double> alpha(nfield), tab((size_t)nfield * T_NSLOT * TS);
  bool all_compact = true;
  std::vector<double> ctab((size_t)nfield * 2 * CT_PART, 0.0);
  const DirSpec spec{3 * TS, 3, 0.0, 0.0, 0.0, 0.0};
  for (int b = 0; b < nfield && !var; ++b) {
    alpha[b] = r * dcoef_host[b];
    double ends[4];
    build_chunk_table(spec, alpha[b], 1, &tab[(size_t)b * T_NSLOT * TS], ends);
    all_compact = all_compact && table_is_compact(&tab[(size_t)b * T_NSLOT * TS]);
    build_compact_table(&tab[(size_t)b * T_NSLOT * TS], &ctab[(size_t)b * 2 * CT_PART]);
  }

  auto* plan = new qp_adi_tile_plan();
  TileView& v = plan->view;
  v.ny = ny; v.nx = nx; v.nfield = nfield; v.py = py; v.px = px; v.pny = pny; v.pnx = pnx;
  v.nbc = (int)bc_index.size();
  v.var = var ? 1 : 0;
  v.stream = stream_mode((size_t)nfield * ny * nx * sizeof(double));
  v.wx = v.wy = v.rdx = v.rdy = nullptr;
  plan->ncell = (long)ny * nx;
  plan->bct_bytes = bct.size() * sizeof(double);
  for (int k = 0; k < 3; ++k) plan->counts[k] = counts[k];
  bool ok = true;
  auto upload = [&](const void* h, size_t bytes) -> void* {
    void* d = nullptr;
    if (!ok) return nullptr;
    if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) { ok = false; return nullptr; }
    plan->allocs.push_back(d);
    if (bytes && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
    return d;
  };
  auto zalloc = [&](size_t bytes) -> void* {
    void* d = nullptr;
    if (!ok) return nullptr;
    if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) { ok = false; return nullptr; }
    plan->allocs.push_back(d);
    if (hipMemset(d, 0, bytes ? bytes : 8) != hipSuccess) ok = false;
    return d;
  };
  v.alpha = (const double*)upload(alpha.data(), alpha.size() * sizeof(double));
  v.tab = (const double*)upload(tab.data(), tab.size() * sizeof(double));
  v.ctab = (const double*)upload(ctab.data(), ctab.size() * sizeof(double));
  v.compact = (all_compact && !var) ? 1 : 0;
  if (const char* e = getenv("QPSIM_COMPACT_TABLES")) v.compact = v.compact && atoi(e) != 0;
  v.code_r = (const uint16_t*)upload(code_r.data(), code_r.size() * sizeof(uint16_t));
  v.code_c = (const uint16_t*)upload(code_c.data(), code_c.size() * sizeof(uint16_t));
  v.bct = (const double*)upload(bct.data(), bct.size() * sizeof(double));
  for (int c = 0; c < 2; ++c) {
    v.tiles[c] = (const int32_t*)upload(lists[c].data(), lists[c].size() * sizeof(int32_t));
    v.ntiles[c] = (int)lists[c].size();
  }
  for (int d = 0; d < 2; ++d) {
    const size_t nl = d == 0 ? ny : nx, P = d == 0 ? px : py;
    v.iface[d] = (double*)zalloc((size_t)nfield * (2 * P + 2) * nl * sizeof(double));
    v.coef[d] = (double*)zalloc((size_t)nfield * (2 * P + 2) * nl * sizeof(double));
  }
  plan->d_work = (double*)zalloc((size_t)nfield * plan->ncell * sizeof(double));
  unsigned long long* d_far = (unsigned long long*)zalloc(sizeof(unsigned long long));
  double* wplanes[4] = {nullptr, nullptr, nullptr, nullptr};
  const size_t pplane = (size_t)nfield * pny * pnx;
  if (var) {
    // one leading pad element per array: the first line reads w[-1] only when a previous chunk exists, but keep the
    // address valid for the hardware prefetch of partial tiles anyway
    for (int k = 0; k < 4; ++k) wplanes[k] = (double*)zalloc(pplane * sizeof(double));
    v.wx = wplanes[0]; v.wy = wplanes[1]; v.rdx = wplanes[2]; v.rdy = wplanes[3];
  }
  if (!ok) {
    (void)hipGetLastError();
    qp_adi_tile_plan_destroy(plan);
    set_error("qp_adi_tile_plan_create: device allocation or upload failed");
    return QP_ERR_ALLOC;
  }
  if (var) {
    hipLaunchKernelGGL(tile_var_weights_kernel, dim3((unsigned)((pplane + 255) / 256)), dim3(256), 0, 0, v, dfield_dev, r,
                       wplanes[0], wplanes[1], wplanes[2], wplanes[3]);
    if (v.ntiles[1] > 0)
      hipLaunchKernelGGL(tile_setup_kernel<2>, dim3((unsigned)((long)v.ntiles[1] * nfield)), dim3(64), plan->bct_bytes, 0,
                         v, 1, d_far);
  } else {
    for (int c = 0; c < 2; ++c)
      if (v.ntiles[c] > 0)
        hipLaunchKernelGGL(tile_setup_kernel<1>, dim3((unsigned)((long)v.ntiles[c] * nfield)), dim3(64), plan->bct_bytes,
                           0, v, c, d_far);
  }
  unsigned long long far_bits = 0;
  if (hipMemcpy(&far_bits, d_far, sizeof(far_bits), hipMemcpyDeviceToHost) != hipSuccess || hipGetLastError() != hipSuccess) {
    qp_adi_tile_plan_destroy(plan);
    set_error("qp_adi_tile_plan_create: setup kernel failed");
    return QP_ERR_LAUNCH;
  }
  double far;
  memcpy(&far, &far_bits, sizeof(far));
  plan->far = far;
  if (!(far < kFarCouplingDrop)) {
    qp_adi_tile_plan_destroy(plan);
    set_error("qp_adi_tile_plan_create: r*D too large for the tiled path (coupling across a 64-cell chunk %.3g, limit %.0e)",
              far, kFarCouplingDrop);
    return QP_ERR_UNSUPPORTED;
  }
  {
    const char* m = getenv("QPSIM_TILE_MERGE");
    const bool few = (long)v.ntiles[1] * nfield <= 1024;
    plan->merged = v.ntiles[0] > 0 && v.ntiles[1] > 0 && !var && v.stream == 0 && v.compact && (m ? atoi(m) != 0 : few);
  }
  {
    // opt-in (QPSIM_TILE_FORK=1): measured on MI355X the event fork / join costs more than the overlap gains for one field
    // (ring in 4096^2: 0.113 -> 0.131 ms per step) and gains 4 % for four fields
    const char* e = getenv("QPSIM_TILE_FORK");
    if (v.ntiles[0] > 0 && v.ntiles[1] > 0 && !var && e && atoi(e) != 0) {
      if (hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (plan->side) (void)hipStreamDestroy(plan->side);
        plan->side = nullptr;      // no side stream: the launches stay in order on the caller's stream
      }
    }
  }
  *out = plan;
  return QP_OK;
}

int qp_adi_tile_plan_create(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dcoef_host,
                            const uint8_t* flags, const double* ex, const double* ey, const double* sx, const double* sy,
                            qp_adi_tile_plan** out) {
  QP_REQUIRE(dcoef_host != nullptr, "dcoef_host is NULL");
  return tile_plan_create(ny, nx, nfield, r, dcoef_host, nullptr, flags, ex, ey, sx, sy, out);
}

int qp_adi_tile_plan_create_var(int32_t ny, int32_t nx, int32_t nfield, double r, const double* dfield,
                                const uint8_t* flags, const double* ex, const double* ey, const double* sx,
                                const double* sy, qp_adi_tile_plan** out) {
  QP_REQUIRE(dfield != nullptr, "dfield is NULL");
  return tile_plan_create(ny, nx, nfield, r, nullptr, dfield, flags, ex, ey, sx, sy, out);
}

int qp_adi_tile_plan_info(const qp_adi_tile_plan* plan, int32_t* counts, double* far) {
  QP_REQUIRE(plan && counts, "plan and counts must be non-NULL");
  for (int k = 0; k < 3; ++k) counts[k] = plan->counts[k];
  if (far) *far = plan->far;
  return QP_OK;
}

}  // extern "C"

namespace qp {

// general tiles on the side stream: fork before the sweep's launches, join after them
static hipStream_t fork_general(const qp_adi_tile_plan* plan, hipStream_t stream) {
  if (!plan->side) return stream;
  (void)hipEventRecord(plan->ev_fork, stream);
  (void)hipStreamWaitEvent(plan->side, plan->ev_fork, 0);
  return plan->side;
}
static void join_general(const qp_adi_tile_plan* plan, hipStream_t stream) {
  if (!plan->side) return;
  (void)hipEventRecord(plan->ev_join, plan->side);
  (void)hipStreamWaitEvent(stream, plan->ev_join, 0);
}

template <bool EXPLICIT>
static void launch_x(const qp_adi_tile_plan* plan, double* buf, hipStream_t stream) {
  const TileView& v = plan->view;
  if (plan->merged) {
    hipLaunchKernelGGL((tile_x_merged_kernel<EXPLICIT>), dim3((unsigned)((long)(v.ntiles[0] + v.ntiles[1]) * v.nfield)),
                       dim3(64), plan->bct_bytes, stream, v, buf);
    return;
  }
  const hipStream_t gstream = fork_general(plan, stream);
  if (v.ntiles[0] > 0) {
    const dim3 grid((unsigned)((long)v.ntiles[0] * v.nfield));
    if (v.compact) {
      if (v.stream == 0) hipLaunchKernelGGL((tile_x_kernel<0, EXPLICIT, 0, true>), grid, dim3(64), 0, stream, v, buf);
      else if (v.stream == 2) hipLaunchKernelGGL((tile_x_kernel<0, EXPLICIT, 2, true>), grid, dim3(64), 0, stream, v, buf);
      else hipLaunchKernelGGL((tile_x_kernel<0, EXPLICIT, 3, true>), grid, dim3(64), 0, stream, v, buf);
    } else {
      if (v.stream == 0) hipLaunchKernelGGL((tile_x_kernel<0, EXPLICIT, 0>), grid, dim3(64), 0, stream, v, buf);
      else if (v.stream == 2) hipLaunchKernelGGL((tile_x_kernel<0, EXPLICIT, 2>), grid, dim3(64), 0, stream, v, buf);
      else hipLaunchKernelGGL((tile_x_kernel<0, EXPLICIT, 3>), grid, dim3(64), 0, stream, v, buf);
    }
  }
  if (v.ntiles[1] > 0 && !v.var)
    hipLaunchKernelGGL((tile_x_kernel<1, EXPLICIT, -1>), dim3((unsigned)((long)v.ntiles[1] * v.nfield)), dim3(64),
                       plan->bct_bytes, gstream, v, buf);
  if (v.ntiles[1] > 0 && v.var)
    hipLaunchKernelGGL((tile_x_kernel<2, EXPLICIT, -1>), dim3((unsigned)((long)v.ntiles[1] * v.nfield)), dim3(64),
                       plan->bct_bytes, stream, v, buf);
  join_general(plan, stream);
}

template <int MODE>
static void launch_y(const qp_adi_tile_plan* plan, const double* src, double* dst, hipStream_t stream) {
  const TileView& v = plan->view;
  if (plan->merged) {
    hipLaunchKernelGGL((tile_y_merged_kernel<MODE>), dim3((unsigned)((long)(v.ntiles[0] + v.ntiles[1]) * v.nfield)),
                       dim3(64), plan->bct_bytes, stream, v, src, dst);
    return;
  }
  const hipStream_t gstream = fork_general(plan, stream);
  if (v.ntiles[0] > 0) {
    const dim3 grid((unsigned)((long)v.ntiles[0] * v.nfield));
    if (v.compact) {
      if (v.stream == 0) hipLaunchKernelGGL((tile_y_kernel<0, MODE, 0, true>), grid, dim3(64), 0, stream, v, src, dst);
      else if (v.stream == 2) hipLaunchKernelGGL((tile_y_kernel<0, MODE, 2, true>), grid, dim3(64), 0, stream, v, src, dst);
      else hipLaunchKernelGGL((tile_y_kernel<0, MODE, 3, true>), grid, dim3(64), 0, stream, v, src, dst);
    } else {
      if (v.stream == 0) hipLaunchKernelGGL((tile_y_kernel<0, MODE, 0>), grid, dim3(64), 0, stream, v, src, dst);
      else if (v.stream == 2) hipLaunchKernelGGL((tile_y_kernel<0, MODE, 2>), grid, dim3(64), 0, stream, v, src, dst);
      else hipLaunchKernelGGL((tile_y_kernel<0, MODE, 3>), grid, dim3(64), 0, stream, v, src, dst);
    }
  }
  if (v.ntiles[1] > 0 && !v.var)
    hipLaunchKernelGGL((tile_y_kernel<1, MODE, -1>), dim3((unsigned)((long)v.ntiles[1] * v.nfield)), dim3(64),
                       plan->bct_bytes, gstream, v, src, dst);
  if (v.ntiles[1] > 0 && v.var)
    hipLaunchKernelGGL((tile_y_kernel<2, MODE, -1>), dim3((unsigned)((long)v.ntiles[1] * v.nfield)), dim3(64),
                       plan->bct_bytes, stream, v, src, dst);
  join_general(plan, stream);
}

}  // namespace qp

extern "C" {

// `nsteps` Peaceman-Rachford steps in place on u[nfield][ny*nx] (cells outside the mask must be, and stay, 0).
int qp_adi_tile_steps(qp_adi_tile_plan* plan, double* u, int32_t nsteps, void* stream_) {
  QP_REQUIRE(plan && u, "plan and u must be non-NULL");
  QP_REQUIRE(nsteps >= 1, "nsteps must be >= 1");
  using namespace qp;
  hipStream_t stream = (hipStream_t)stream_;
  double* w = plan->d_work;
  launch_y<0>(plan, u, w, stream);
  for (int s = 0; s < nsteps; ++s) {
    launch_x<true>(plan, w, stream);
    if (s + 1 < nsteps) launch_y<1>(plan, w, w, stream);
    else launch_y<2>(plan, w, u, stream);
  }
  return check_launch("qp_adi_tile_steps");
}

// x <- (I - a Ly)^-1 (I - a Lx)^-1 x in place (ADI preconditioner of the unsplit CN matrix); x = 0 outside the mask.
int qp_adi_tile_solve(qp_adi_tile_plan* plan, double* x, void* stream_) {
  QP_REQUIRE(plan && x, "plan and x must be non-NULL");
  using namespace qp;
  hipStream_t stream = (hipStream_t)stream_;
  launch_y<3>(plan, x, x, stream);
  launch_x<false>(plan, x, stream);
  launch_y<2>(plan, x, x, stream);
  return check_launch("qp_adi_tile_solve");
}

}  // extern "C"
