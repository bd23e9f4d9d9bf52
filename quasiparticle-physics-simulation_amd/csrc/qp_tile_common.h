// Device helpers shared by the tiled ADI kernels (qp_adi_rect.hip: full rectangles, qp_adi_tile.hip: masked grids).
// One wave owns one 64 x 64 tile held as 64 doubles per lane.
#pragma once

#include <stdlib.h>

#include "qp_common.h"

namespace qp {

constexpr int TS = 64;          // tile edge = chunk length
// far-corner weights of the reduced system below this (relative to its unit diagonal) are dropped: six orders of
// magnitude under the fp64 rounding of the retained terms
constexpr double kFarCouplingDrop = 1e-22;
// carried plane sets above this size bypass the caches (see TileCoord::stream)
constexpr size_t kStreamBytes = (size_t)192 << 20;
// bit 0: non-temporal loads, bit 1: non-temporal stores; QPSIM_STREAM_MODE=0..3 overrides the size rule (tuning knob)
inline int stream_mode(size_t plane_set_bytes) {
  if (const char* e = getenv("QPSIM_STREAM_MODE")) return atoi(e) & 3;
  // measured on MI355X (fraction of 8 TB/s, cached / NT stores / NT both): 4096^2 (128 MiB) 0.62 / 0.59 / 0.60;
  // 8192^2 (512 MiB) 0.61 / 0.735 / 0.69; 16384^2 (2 GiB) 0.64 / 0.67 / 0.71
  if (plane_set_bytes > 4 * kStreamBytes) return 3;
  return plane_set_bytes > kStreamBytes ? 2 : 0;      // (an environment override of 1 is served by the kernels of 3)
}

// table slots per (direction, field, chunk variant); each slot is TS doubles
enum { T_W = 0, T_AWF, T_AWB, T_CM, T_C0, T_CP, T_SRC, T_G, T_H, T_NSLOT };

// Tables are written once at plan creation and never by a kernel: read them through the constant address space so
// that wave-uniform accesses become scalar loads (s_load) and the values feed the FMAs straight from SGPRs.
typedef const double __attribute__((address_space(4))) * ctab_t;

__device__ __forceinline__ ctab_t as_const(const double* p) {
  // the address is wave-uniform by construction (kernel arguments and blockIdx only): say so explicitly
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (ctab_t)(((unsigned long long)hi << 32) | lo);
}

// Thomas solve of one chunk held in registers; padded entries (k >= chunk length) carry w = 1, aw = 0.
__device__ __forceinline__ void thomas64(double (&e)[TS], ctab_t t) {
  double dp = 0.0;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    dp = fma(t[T_AWF * TS + k], dp, e[k] * t[T_W * TS + k]);
    e[k] = dp;
  }
  double x = 0.0;
#pragma unroll
  for (int k = TS - 1; k >= 0; --k) {
    x = fma(t[T_AWB * TS + k], x, e[k]);
    e[k] = x;
  }
}

// e <- (I + a L) e + a s along the chunk, with neighbour values gl / gr beyond its ends, plus `extra` on valid cells.
__device__ __forceinline__ void explicit64(double (&e)[TS], double gl, double gr, ctab_t t, double extra) {
  double prev = gl;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    const double cur = e[k];
    const double nxt = (k + 1 < TS) ? e[k + 1] : gr;
    // `extra` also lands on padded cells (k >= chunk length); those are never stored and meet zero weights in dots64
    e[k] = fma(t[T_CM * TS + k], prev, fma(t[T_CP * TS + k], nxt, fma(t[T_C0 * TS + k], cur, t[T_SRC * TS + k] + extra)));
    prev = cur;
  }
}

__device__ __forceinline__ void dots64(const double (&e)[TS], ctab_t t, double& yf, double& yl) {
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    a0 = fma(t[T_G * TS + k], e[k], a0);
    a1 = fma(t[T_H * TS + k], e[k], a1);
  }
  yf = a0;
  yl = a1;
}

// In-place transpose of the 64 x 64 tile distributed as v[j] on lane i  ->  v[i] on lane j (its own inverse:
// lane = column / index = row  <->  lane = row / index = column).  Viewed as 2 x 2 blocks of 32 x 32:
//   1. v_permlane32_swap exchanges the off-diagonal blocks between the half-waves (registers 0..31 of lanes 32..63
//      <-> registers 32..63 of lanes 0..31), no memory involved;
//   2. every block is then transposed inside its own half-wave through a 32 x 33 LDS block, registers 0..31 first,
//      32..63 second, so only 2 x 32 x 33 doubles (16.5 KiB per wave instead of 33 KiB) are live and no lane-dependent
//      register index appears.  Pitch 33 keeps the column-wise writes and row-wise reads conflict-free.
constexpr int HB = 32;
constexpr int HP = HB + 1;
constexpr int LDS_DOUBLES = 2 * HB * HP;

__device__ __forceinline__ void swap_half_waves(double& lo_reg, double& hi_reg) {
  // lanes 32..63 of lo_reg <-> lanes 0..31 of hi_reg
  const unsigned long long a = __double_as_longlong(lo_reg), b = __double_as_longlong(hi_reg);
  const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
  lo_reg = __longlong_as_double(((unsigned long long)r1[0] << 32) | r0[0]);
  hi_reg = __longlong_as_double(((unsigned long long)r1[1] << 32) | r0[1]);
}

__device__ __forceinline__ void transpose64(double (&v)[TS], double* lds, int lane) {
  const int l = lane & 31;
  double* blk = lds + (lane >> 5) * (HB * HP);
#pragma unroll
  for (int k = 0; k < HB; ++k) swap_half_waves(v[k], v[HB + k]);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int k = 0; k < HB; ++k) blk[k * HP + l] = v[half * HB + k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HB; ++k) v[half * HB + k] = blk[l * HP + k];
    __syncthreads();
  }
}

struct TileCoord {
  int b, ty, tx, j0, i0, nr, nc;
  int stream;     // planes larger than the 256 MiB Infinity Cache: non-temporal loads / stores (measured +13 % at 8192^2,
                  // -5 % at 4096^2 where the carried plane survives in the cache between sweeps; set per plan)
};

// Row r of the tile starts at a wave-uniform address and the lane adds a 32-bit offset.  (hipcc 7.2 still forms a 64-bit
// per-lane address with one v_lshl_add_u64 per row instead of the SGPR-base addressing mode; the kernels are
// bandwidth-bound, so this costs issue slots only.  Buffer instructions - one resource descriptor per tile, lane offset
// in one VGPR, row offset as scalar soffset - remove that arithmetic but measured 2-4 % slower on MI355X.)
__device__ __forceinline__ void load_cols(const double* __restrict__ base, const TileCoord& t, int nx, int lane,
                                          double (&v)[TS]) {
  const double* p = base + (long)t.j0 * nx + t.i0;
  const unsigned l = (unsigned)lane;
  if (t.nr == TS && t.nc == TS) {      // interior tile (wave-uniform test): 64 unconditional row-segment loads
    if (t.stream & 1) {
      // the empty asm keeps the two branches distinct: without it the optimiser merges their (otherwise identical)
      // instructions and drops the non-temporal hint
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < TS; ++r) v[r] = __builtin_nontemporal_load(&(p + (long)r * nx)[l]);
      asm volatile("" ::: "memory");
    } else {
#pragma unroll
      for (int r = 0; r < TS; ++r) v[r] = (p + (long)r * nx)[l];
    }
  } else {
    const bool on = lane < t.nc;
#pragma unroll
    for (int r = 0; r < TS; ++r) v[r] = (on && r < t.nr) ? (p + (long)r * nx)[l] : 0.0;
  }
}

__device__ __forceinline__ void store_cols(double* __restrict__ base, const TileCoord& t, int nx, int lane,
                                           const double (&v)[TS]) {
  double* p = base + (long)t.j0 * nx + t.i0;
  const unsigned l = (unsigned)lane;
  if (t.nr == TS && t.nc == TS) {
    if (t.stream & 2) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < TS; ++r) __builtin_nontemporal_store(v[r], &(p + (long)r * nx)[l]);
      asm volatile("" ::: "memory");
    } else {
#pragma unroll
      for (int r = 0; r < TS; ++r) (p + (long)r * nx)[l] = v[r];
    }
  } else if (lane < t.nc) {
#pragma unroll
    for (int r = 0; r < TS; ++r)
      if (r < t.nr) (p + (long)r * nx)[l] = v[r];
  }
}

// host: fills the T_NSLOT x TS table of chunk `p` of a line described by (n, P, end-face terms); returns
// g[0], g[last], h[0], h[last] through `ends`
struct DirSpec {
  int n;          // line length
  int P;          // chunks
  double e_lo, e_hi;   // BC diagonal terms of the two end faces (1/dx^2 units)
  double s_lo, s_hi;   // BC sources of the two end faces
};
void build_chunk_table(const DirSpec& s, double a, int p, double* tab, double ends[4]);

}  // namespace qp
