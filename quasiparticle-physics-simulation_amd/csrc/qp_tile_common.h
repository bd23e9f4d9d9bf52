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
  // measured on MI355X, round 2 kernels (sweep time in us, cached / NT stores / NT both): 5760^2 (253 MiB) 91.7 / 89.8 / 97.2;
  // 4096^2 x 2 (256 MiB) 88.5 / 89.5 / 93.5; 8192^2 (512 MiB) 211 / 206 / 188; 4096^2 x 4 (512 MiB) 200 / 190 / 177;
  // 11520^2 (1 GiB) 407 / 410 / 376: once the carried planes no longer fit the 256 MiB Infinity Cache both directions
  // should bypass it
  if (plane_set_bytes > 2 * kStreamBytes) return 3;
  return plane_set_bytes > kStreamBytes ? 2 : 0;      // (an environment override of 1 is served by the kernels of 3)
}

// table slots per (direction, field, chunk variant); each slot is TS doubles.  T_W..T_SRC drive the solve and the explicit
// operator along the chunk; T_EW..T_EAV the two eliminations that give the first / last entry of A_p^-1 d (forward sweep
// with the LU pivots, backward sweep with the UL pivots) for the reduced right-hand sides of the other direction.
enum { T_W = 0, T_AWF, T_AWB, T_CM, T_C0, T_CP, T_SRC, T_EW, T_EAWF, T_EV, T_EAV, T_NSLOT };

// Compact form of a table: the pivots of a chunk converge to their fixed point within a dozen cells of its end (rate rho^2
// per cell) and the explicit-operator coefficients differ from (a, 1 - 2a, a, 0) only in the cells that touch a wall, so for
// every full-length chunk each slot is CONSTANT on [pre, TS - suf): a short prefix / suffix of distinct entries and one
// value for the middle.  The plan checks that bit for bit per table (`table_is_compact`); kernels then feed the middle of
// every dependent chain from one SGPR pair and fetch only the prefix / suffix entries.
// The same holds for chunks of FS = 32 cells (the fine tiles of qp_adi_fine.inc): the functions take the chunk length L.
constexpr int FS = 32;
__host__ __device__ constexpr int slot_pre(int slot) {
  return (slot == T_W || slot == T_AWF || slot == T_AWB || slot == T_EW || slot == T_EAWF) ? 16 : 1;
}
__host__ __device__ constexpr int slot_suf(int slot) { return (slot == T_EV || slot == T_EAV) ? 16 : 1; }
// an entry of the constant stretch [pre, L - suf) of a slot
__host__ __device__ constexpr int slot_mid(int L, int slot) { return (slot == T_EV || slot == T_EAV) ? L - 17 : 16; }
inline bool table_is_compact_len(int L, const double* tab) {
  for (int q = 0; q < T_NSLOT; ++q)
    for (int k = slot_pre(q); k < L - slot_suf(q); ++k)
      if (tab[q * L + k] != tab[q * L + slot_mid(L, q)]) return false;
  return true;
}
inline bool table_is_compact(const double* tab) { return table_is_compact_len(TS, tab); }
// Compact table = two parts of CT_PART doubles (solve slots T_W..T_SRC, elimination slots T_EW..T_EAV): the 16-entry
// prefixes / suffixes, then per slot its middle value, its last (or first) entry.  cidx maps (slot, k) into its part.
constexpr int CT_PART = 72;
__host__ __device__ constexpr int cidx_len(int L, int slot, int k) {
  if (slot < T_EW) {
    if (slot <= T_AWB && k < 16) return slot * 16 + k;                  // W, AWF, AWB prefixes: 0..47
    if (k == L - 1) return 56 + slot;                                   // last entry of every solve slot: 56..62
    if (slot >= T_CM && k == 0) return 64 + (slot - T_CM);              // first entry of the explicit slots: 64..67
    return 48 + slot;                                                   // middle: 48..54
  }
  const int q = slot - T_EW;                                            // 0 EW, 1 EAWF, 2 EV, 3 EAV
  if (q < 2 && k < 16) return q * 16 + k;                               // forward prefixes: 0..31
  if (q >= 2 && k >= L - 16) return q * 16 + (k - (L - 16));            // backward suffixes: 32..63
  if (q < 2 && k == L - 1) return 68 + q;                               // EW, EAWF last: 68, 69
  if (q >= 2 && k == 0) return 68 + q;                                  // EV, EAV first: 70, 71
  return 64 + q;                                                        // middle: 64..67
}
__host__ __device__ constexpr int cidx(int slot, int k) { return cidx_len(TS, slot, k); }
inline void build_compact_table_len(int L, const double* tab, double* ct) {       // ct[2][CT_PART]
  for (int i = 0; i < 2 * CT_PART; ++i) ct[i] = 0.0;
  for (int q = 0; q < T_NSLOT; ++q)
    for (int k = 0; k < L; ++k) ct[(q < T_EW ? 0 : CT_PART) + cidx_len(L, q, k)] = tab[q * L + k];
}
inline void build_compact_table(const double* tab, double* ct) { build_compact_table_len(TS, tab, ct); }

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

#ifndef QP_ABL
#define QP_ABL 0      // timing-only ablation builds (tools/ablate.sh): results are wrong, only the clock matters
#endif

// Where the solves get coefficient k of a slot from.  The chains need it as a wave-uniform operand at step k.
//  * CoefCompact (every table of the plan has the compact form - all grids whose extents are multiples of 64 with
//    r D <~ 1.5): a 576-byte scalar table per part; the middle of every chain runs on one SGPR pair per slot and only the
//    16-entry prefixes / suffixes are fetched (a handful of s_load_dwordx16, issued while the tile rows are in flight).
//  * CoefFull: lane k loads entry k of every slot once (one coalesced 512 B load per slot, ahead of the tile rows) and step
//    k fetches it with v_readlane into an SGPR pair - two VALU instructions per coefficient, no memory latency in the
//    chain.  (Streaming full tables through scalar loads put a scalar-memory round trip in front of every few steps:
//    16 % of the 4096^2 sweep and 34 % of the 2048^2 sweep were spent in those waits.)
struct TabRegs {
  double s[T_NSLOT];
};

template <int FIRST, int COUNT>
__device__ __forceinline__ void load_tab(const double* __restrict__ tab, int lane, TabRegs& r) {
#pragma unroll
  for (int q = FIRST; q < FIRST + COUNT; ++q) r.s[q] = tab[q * TS + lane];
}

__device__ __forceinline__ double tab_at(const TabRegs& r, int slot, int k) {
  // The empty volatile asm "redefines" the source at this point of the instruction stream: readlanes are pure, and
  // without it the scheduler hoists hundreds of them to the top of the unrolled chains, runs out of SGPRs and spills
  // them into VGPR lanes.  Volatile asms keep their relative order.
  double src = r.s[slot];
  asm volatile("" : "+v"(src));
  const unsigned long long b = (unsigned long long)__double_as_longlong(src);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

struct CoefFull {
  TabRegs r;
  __device__ __forceinline__ double at(int slot, int k) const {
    if (QP_ABL & 2) return 0.1 + 0.001 * slot;
    return tab_at(r, slot, k);
  }
};

struct CoefCompact {
  ctab_t part;      // the part (solve or elimination) of the compact table this kernel phase reads
  __device__ __forceinline__ double at(int slot, int k) const {
    if (QP_ABL & 2) return 0.1 + 0.001 * slot;
    return part[cidx(slot, k)];
  }
};

// Touches the nine 64-byte lines of a compact-table part so that the chains' scalar loads hit the scalar cache: the loads
// the compiler places inside the dependent chains (a dozen s_waitcnt per tile) otherwise each pay a trip to L2 whenever
// the line has left the scalar cache - always on small grids, where every wave is the first of its launch on its CU
// (~3 us per tile at 1024^2), and often enough on large ones (4096^2 fine tiles: 48.4 -> 45.8 us per sweep).  Called right
// after the tile's row loads have been issued, so the one wait here overlaps their latency.
template <class Coef, int NPART>
__device__ __forceinline__ void warm_scalar_cache(const Coef (&t)[NPART]) {
  if (QP_ABL & 64) return;
  double w[NPART][9];
#pragma unroll
  for (int q = 0; q < NPART; ++q)
#pragma unroll
    for (int i = 0; i < 9; ++i) w[q][i] = t[q].part[8 * i];
  // all loads above, one consumer per part below: a single wait covers them
#pragma unroll
  for (int q = 0; q < NPART; ++q)
    asm volatile("" ::"s"(w[q][0]), "s"(w[q][1]), "s"(w[q][2]), "s"(w[q][3]), "s"(w[q][4]), "s"(w[q][5]), "s"(w[q][6]),
                 "s"(w[q][7]), "s"(w[q][8]));
}

// Thomas solve of one chunk held in registers; padded entries (k >= chunk length) carry w = 1, aw = 0.
template <class Coef>
__device__ __forceinline__ void thomas64(double (&e)[TS], const Coef& t) {
  if (QP_ABL & 16) return;
  double dp = 0.0;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    dp = fma(t.at(T_AWF, k), dp, e[k] * t.at(T_W, k));
    e[k] = dp;
  }
  double x = 0.0;
#pragma unroll
  for (int k = TS - 1; k >= 0; --k) {
    x = fma(t.at(T_AWB, k), x, e[k]);
    e[k] = x;
  }
}

// e <- (I + a L) e + a s along the chunk, with neighbour values gl / gr beyond its ends, plus `extra` on valid cells.
template <class Coef>
__device__ __forceinline__ void explicit64(double (&e)[TS], double gl, double gr, const Coef& t, double extra) {
  if (QP_ABL & 32) return;
  double prev = gl;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    const double cur = e[k];
    const double nxt = (k + 1 < TS) ? e[k + 1] : gr;
    // `extra` also lands on padded cells (k >= chunk length); those are never stored and are skipped by ends64
    e[k] = fma(t.at(T_CM, k), prev, fma(t.at(T_CP, k), nxt, fma(t.at(T_C0, k), cur, t.at(T_SRC, k) + extra)));
    prev = cur;
  }
}

// First / last entry of A_p^-1 e (the reduced right-hand sides of the chunk): yl is where the forward elimination ends,
// yf where the backward one does.  Padded entries pass the forward value through (ew = 0, eawf = 1) and keep the backward
// one at zero until the last valid cell (ev = eav = 0), so whatever sits in padded cells is never looked at.
template <class Coef>
__device__ __forceinline__ void ends64(const double (&e)[TS], const Coef& t, double& yf, double& yl) {
  double dp = 0.0, bp = 0.0;
#pragma unroll
  for (int k = 0; k < TS; ++k) {
    const int j = TS - 1 - k;
    dp = fma(t.at(T_EAWF, k), dp, e[k] * t.at(T_EW, k));
    bp = fma(t.at(T_EAV, j), bp, e[j] * t.at(T_EV, j));
  }
  yf = bp;
  yl = dp;
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

// The workgroup of every tile kernel is ONE wave (they are launched with 64 threads) and a wave's LDS instructions execute in order, so
// no s_barrier is needed between the writes and the reads of a patch - only a fence that keeps the compiler from moving
// LDS accesses across it.
__device__ __forceinline__ void wave_lds_fence() {
#ifdef QP_TILE_BARRIERS
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

__device__ __forceinline__ void swap_half_waves(double& lo_reg, double& hi_reg) {
  // lanes 32..63 of lo_reg <-> lanes 0..31 of hi_reg
  const unsigned long long a = __double_as_longlong(lo_reg), b = __double_as_longlong(hi_reg);
  const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
  lo_reg = __longlong_as_double(((unsigned long long)r1[0] << 32) | r0[0]);
  hi_reg = __longlong_as_double(((unsigned long long)r1[1] << 32) | r0[1]);
}

__device__ __forceinline__ void transpose64(double (&v)[TS], double* lds, int lane) {
  if (QP_ABL & 4) return;
  const int l = lane & 31;
  double* blk = lds + (lane >> 5) * (HB * HP);
#pragma unroll
  for (int k = 0; k < HB; ++k) swap_half_waves(v[k], v[HB + k]);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int k = 0; k < HB; ++k) blk[k * HP + l] = v[half * HB + k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < HB; ++k) v[half * HB + k] = blk[l * HP + k];
    wave_lds_fence();
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

// v += scale * (tile of `base`), same tile / bounds as load_cols (source plane of the Peaceman-Rachford passes)
__device__ __forceinline__ void add_source_cols(const double* __restrict__ base, const TileCoord& t, int nx, int lane,
                                                double scale, double (&v)[TS]) {
  const double* p = base + (long)t.j0 * nx + t.i0;
  const unsigned l = (unsigned)lane;
  if (t.nr == TS && t.nc == TS) {
#pragma unroll
    for (int r = 0; r < TS; ++r) v[r] = fma(scale, (p + (long)r * nx)[l], v[r]);
  } else if (lane < t.nc) {
#pragma unroll
    for (int r = 0; r < TS; ++r)
      if (r < t.nr) v[r] = fma(scale, (p + (long)r * nx)[l], v[r]);
  }
}

// host: fills the T_NSLOT x TS table of chunk `p` of a line described by (n, P, end-face terms); returns
// g[0], g[last], h[0], h[last] through `ends`
struct DirSpec {
  int n;          // line length
  int P;          // chunks
  double e_lo, e_hi;   // BC diagonal terms of the two end faces (1/dx^2 units)
  double s_lo, s_hi;   // BC sources of the two end faces
  double c0_shift = 0.0;   // explicit operator (I + a L) - c0_shift I (Peaceman-Rachford iteration plans)
};
void build_chunk_table(const DirSpec& s, double a, int p, double* tab, double ends[4]);

}  // namespace qp
