// Collision update with one wave per pixel, lane <-> energy bin (NE <= 64, any gap-class map, any phonon-bin map).
//
// The register kernel of qp_collision_fast.hip keeps 4 NE doubles per thread and unrolls NE^2 pairs; beyond NE = 16 (the
// reference's default is NE = 50) that neither fits nor compiles in reasonable time.  Here a wave works on one pixel at a
// time: lane i owns n_i, q_i and the gain / loss sums of bin i; the loop over j broadcasts n_j, q_j with v_readlane
// (scalar operands), lane i does the pair (i, j).  The pixel's phonon occupations and the per-bin sums live in LDS
// (3 NW doubles), so the relaxation / growth updates cost one exp per lane and per 64 bins instead of NE + NW per thread.
// A wave takes 8 consecutive pixels per block so that every lane reads / writes 64 contiguous bytes of its planes.
//
// K^s_0, K^r_0, idx_diff and idx_sum are symmetric and sign is antisymmetric, so row j of each table (contiguous over
// lanes) serves the pair (lane, j).  Per-bin sums: for a fixed j the lanes of one instruction hit distinct bins when the
// maps have the |i-j| / i+j structure (emission and absorption go to different arrays), so a plain read-modify-write by
// the single wave is race-free and deterministic; otherwise (unstructured maps) LDS atomics are used.
#include <algorithm>

#include "qp_common.h"

namespace qp {

struct CollView;  // defined in qp_collision.hip

struct WaveCollView {
  int ne, nw, nclass;
  const double* kr0;
  const double* ks0;
  const double* rho;
  const int32_t* idx_diff;
  const int32_t* idx_sum;
  const int8_t* sign;
  const int32_t* cls;
  const int32_t* diag_bin;   // [ne]     bin of |Ei-Ej| for |i-j| = k, or NULL (unstructured maps)
  const int32_t* anti_bin;   // [2ne-1]  bin of Ei+Ej for i+j = m
};

__device__ __forceinline__ double relax_update_w(double n, double gain, double loss, double dt) {
  const double mu = fmax(loss, 0.0);
  const double P = fmax(gain + (mu - loss) * n, 0.0);
  const double decay = exp(-mu * dt);
  const double coeff = (mu < 1e-14) ? dt : (1.0 - decay) / mu;
  return fmax(decay * n + coeff * P, 0.0);
}

__device__ __forceinline__ double affine_update_w(double y, double a, double b, double dt) {
  const double xx = fmin(fmax(b * dt, -80.0), 80.0);
  const double ex = exp(xx);
  const double coeff = (fabs(b) < 1e-14) ? dt : (ex - 1.0) / b;
  return fmax(ex * y + coeff * a, 0.0);
}

__device__ __forceinline__ double bcast(double x, int srclane) {
  const unsigned long long u = __double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, srclane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), srclane);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

constexpr int PB = 8;        // pixels per wave and group
constexpr int MAXBINS = 3;   // phonon bins per lane: NW <= 3*64 - 1 for NE <= 64
constexpr int WAVES = 4;
constexpr int PG = 4;        // pixels advanced together through the j-loop (structured variant)

// STAGED (one gap class): the block's 4 waves first copy K^s_0, K^r_0, idx_diff, idx_sum and sign into LDS and then read
// row j from there; otherwise the rows come from global memory (L1/L2) with the class offset of the pixel.  Blocks are
// persistent (each wave strides over groups of 8 pixels), so the staging cost is paid once per block.
// Waves only synchronise once (after staging); inside a wave LDS accesses execute in order, so wave_barrier() (a
// code-motion barrier) is all that is needed between the phases of a pixel.
// Every shared access is written as lds[integer index]: pointer selects or integer round-trips of the base make the
// compiler fall back to flat (generic address space) accesses, which are several times slower than ds_* instructions.
__device__ __forceinline__ double gather(double x, int srclane) {   // value of x on lane (srclane & 63)
  const unsigned long long u = __double_as_longlong(x);
  const int a = (srclane & 63) << 2;
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(a, (int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(a, (int)(unsigned)(u >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// STRUCT: the bin maps are idx_diff[i][j] = D[|i-j|], idx_sum[i][j] = S[i+j] (bins may be shared between a diagonal and
// an anti-diagonal).  Then nothing in the j-loop has to touch LDS memory: lane l keeps the occupation of bins D[l], S[l],
// S[l+64] and the running sums of diagonal l and anti-diagonals l, l+64; the occupation a pair needs and the pair's
// contribution to those sums travel between lanes with ds_bpermute (a crossbar gather, no dependent read-modify-write).
// The generic variant (ATOMIC or plain RMW on per-bin LDS slots) serves unstructured maps.
template <bool ATOMIC, bool STAGED, bool STRUCT>
__global__ void __launch_bounds__(64 * WAVES) collision_wave_kernel(WaveCollView t, const uint8_t* __restrict__ flags,
                                                                   long ncell, const double* __restrict__ sin_,
                                                                   double* __restrict__ sout, double* __restrict__ ph,
                                                                   double dE, double dt, int en_r, int en_s, int upd_ph) {
  extern __shared__ double lds[];
  int* ilds = (int*)lds;
  signed char* blds = (signed char*)lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NE = t.ne, NW = t.nw, NN = NE * NE;
  const bool use_s = en_s && t.ks0, use_r = en_r && t.kr0;
  const bool do_ph = upd_ph && (use_s || use_r);
  // carve-up (STAGED): doubles [0, NN) K^s, [NN, 2NN) K^r; ints [4NN, 5NN) idx_diff, [5NN, 6NN) idx_sum; bytes [24NN, 25NN) sign
  const int o_kr = NN, o_idd = 4 * NN, o_ids = 5 * NN, o_sg = 24 * NN;
  const int o_wave = (STAGED ? (25 * NN + 7) / 8 : 0) + wave * 3 * NW;   // per-wave: pP | A | Bm
  const int o_A = o_wave + NW, o_B = o_wave + 2 * NW;
  if (STAGED) {
    for (int q = threadIdx.x; q < NN; q += 64 * WAVES) {
      lds[q] = use_s ? t.ks0[q] : 0.0;
      lds[o_kr + q] = use_r ? t.kr0[q] : 0.0;
      ilds[o_idd + q] = t.idx_diff[q];
      ilds[o_ids + q] = t.idx_sum[q];
      blds[o_sg + q] = t.sign[q];
    }
    __syncthreads();
  }
  const bool on = lane < NE;
  int binD = 0, binS0 = 0, binS1 = 0;
  if (STRUCT) {
    binD = on ? t.diag_bin[lane] : 0;
    binS0 = lane < 2 * NE - 1 ? t.anti_bin[lane] : 0;
    binS1 = lane + 64 < 2 * NE - 1 ? t.anti_bin[lane + 64] : 0;
  }
  const long ngroups = (ncell + PB - 1) / PB;
  for (long grp = (long)blockIdx.x * WAVES + wave; grp < ngroups; grp += (long)gridDim.x * WAVES) {
    const long p0 = grp * PB;
    const int npx = (int)min((long)PB, ncell - p0);
    double n[PB], pb[MAXBINS][PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) n[k] = (on && k < npx) ? sin_[(long)lane * ncell + p0 + k] : 0.0;
#pragma unroll
    for (int s = 0; s < MAXBINS; ++s) {
      const int w = lane + 64 * s;
#pragma unroll
      for (int k = 0; k < PB; ++k) pb[s][k] = (w < NW && k < npx) ? ph[(long)w * ncell + p0 + k] : 0.0;
    }

    if (STRUCT) {
      // PG pixels advance through the j-loop together: PG independent gather/accumulate chains per LDS wait, and (one
      // gap class) one table row read for all of them.
#pragma unroll
      for (int k0 = 0; k0 < PB; k0 += PG) {
        if (k0 >= npx) break;
        bool act[PG];
        double ni[PG], qi[PG], Pd[PG], Ps0[PG], Ps1[PG];
        const double* ksq[PG];
        const double* krq[PG];
#pragma unroll
        for (int q = 0; q < PG; ++q) {
          const int k = k0 + q;
          act[q] = k < npx && (flags[p0 + min(k, npx - 1)] & QP_FLAG_ACTIVE);
          const int c = (!STAGED && t.cls && k < npx) ? t.cls[p0 + k] : 0;
          const double rho_i = on ? t.rho[(long)c * NE + lane] : 0.0;
          ksq[q] = use_s ? t.ks0 + (long)c * NN : nullptr;
          krq[q] = use_r ? t.kr0 + (long)c * NN : nullptr;
          ni[q] = n[k];
          qi[q] = rho_i * fmax(1.0 - ni[q] / fmax(rho_i, 1e-30), 0.0);
#pragma unroll
          for (int s = 0; s < MAXBINS; ++s) {
            const int w = lane + 64 * s;
            if (w < NW) lds[o_wave + w] = pb[s][k];
          }
          __builtin_amdgcn_wave_barrier();
          Pd[q] = lds[o_wave + binD];
          Ps0[q] = lds[o_wave + binS0];
          Ps1[q] = lds[o_wave + binS1];
          __builtin_amdgcn_wave_barrier();
        }
        double g_s[PG], l_s[PG], g_r[PG], l_r[PG], em[PG], ab[PG], rec0[PG], rec1[PG], pb0[PG], pb1[PG];
#pragma unroll
        for (int q = 0; q < PG; ++q)
          g_s[q] = l_s[q] = g_r[q] = l_r[q] = em[q] = ab[q] = rec0[q] = rec1[q] = pb0[q] = pb1[q] = 0.0;
        for (int j = 0; j < NE; ++j) {
          const int row = on ? j * NE + lane : 0;
          const int dlt = lane - j, m = lane + j;
          const bool em_ok = lane >= 1 && m < NE, ab_ok = lane >= 1 && dlt <= 0;
          const bool r0_ok = dlt >= 0 && dlt < NE, r1_ok = dlt + 64 < NE;
          double Ks_shared = 0.0, Kr_shared = 0.0;
          if (STAGED) {
            Ks_shared = (on && use_s) ? lds[row] : 0.0;
            Kr_shared = (on && use_r) ? lds[o_kr + row] : 0.0;
          }
#pragma unroll
          for (int q = 0; q < PG; ++q) {
            const double nj = bcast(ni[q], j), qj = bcast(qi[q], j);
            if (use_s) {
              const double K = STAGED ? Ks_shared : (on ? ksq[q][row] : 0.0);
              const double P = gather(Pd[q], dlt < 0 ? -dlt : dlt);
              g_s[q] = fma(K * (dlt < 0 ? 1.0 + P : P), nj, g_s[q]);
              l_s[q] = fma(K * (dlt > 0 ? 1.0 + P : P), qj, l_s[q]);
              if (do_ph) {
                const double v = dE * (ni[q] * K * qj);            // pair (lane, j); zero on the diagonal (K = 0)
                const double ve = gather(v, m);                     // from lane l + j > j: emission into diagonal l
                const double va = gather(v, -dlt);                  // from lane j - l < j: absorption
                em[q] += em_ok ? ve : 0.0;
                ab[q] += ab_ok ? va : 0.0;
              }
            }
            if (use_r) {
              const double K = STAGED ? Kr_shared : (on ? krq[q][row] : 0.0);
              const double Pa = gather(Ps0[q], m), Pb = gather(Ps1[q], m);
              const double P = m < 64 ? Pa : Pb;
              l_r[q] = fma(K * (1.0 + P), nj, l_r[q]);
              g_r[q] = fma(K * P, qj, g_r[q]);
              if (do_ph) {
                const double vr = dE * (ni[q] * K * nj), vp = dE * (qi[q] * K * qj);
                const double r0 = gather(vr, dlt), p0v = gather(vp, dlt);            // anti-diagonal l: source lane l - j
                const double r1 = gather(vr, dlt + 64), p1v = gather(vp, dlt + 64);  // l + 64: source lane l + 64 - j
                rec0[q] += r0_ok ? r0 : 0.0;
                pb0[q] += r0_ok ? p0v : 0.0;
                rec1[q] += r1_ok ? r1 : 0.0;
                pb1[q] += r1_ok ? p1v : 0.0;
              }
            }
          }
        }
#pragma unroll
        for (int q = 0; q < PG; ++q) {
          const int k = k0 + q;
          if (!act[q]) continue;                                   // wave-uniform: holes / tail pass through unchanged
          n[k] = relax_update_w(ni[q], dE * qi[q] * g_s[q] + 2.0 * dE * qi[q] * g_r[q], dE * l_s[q] + 2.0 * dE * l_r[q], dt);
          if (do_ph) {   // per-bin sums through LDS: diagonals first, then the anti-diagonals on top (bins may be shared)
#pragma unroll
            for (int s = 0; s < MAXBINS; ++s) {
              const int w = lane + 64 * s;
              if (w < NW) { lds[o_A + w] = 0.0; lds[o_B + w] = 0.0; }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane >= 1 && on) { lds[o_A + binD] = em[q]; lds[o_B + binD] = ab[q]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < 2 * NE - 1) { lds[o_A + binS0] += rec0[q]; lds[o_B + binS0] += pb0[q]; }
            __builtin_amdgcn_wave_barrier();
            if (lane + 64 < 2 * NE - 1) { lds[o_A + binS1] += rec1[q]; lds[o_B + binS1] += pb1[q]; }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < MAXBINS; ++s) {
              const int w = lane + 64 * s;
              if (w < NW) pb[s][k] = affine_update_w(pb[s][k], lds[o_A + w], lds[o_A + w] - lds[o_B + w], dt);
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      if (k >= npx) break;
      if (!(flags[p0 + k] & QP_FLAG_ACTIVE)) continue;      // wave-uniform: holes pass through unchanged
      const int c = (!STAGED && t.cls) ? t.cls[p0 + k] : 0;
      const double rho_i = on ? t.rho[(long)c * NE + lane] : 0.0;
      const double* ks = use_s ? t.ks0 + (long)c * NN : nullptr;
      const double* kr = use_r ? t.kr0 + (long)c * NN : nullptr;
      const double ni = n[k];
      const double qi = rho_i * fmax(1.0 - ni / fmax(rho_i, 1e-30), 0.0);
#pragma unroll
      for (int s = 0; s < MAXBINS; ++s) {
        const int w = lane + 64 * s;
        if (w < NW) { lds[o_wave + w] = pb[s][k]; lds[o_A + w] = 0.0; lds[o_B + w] = 0.0; }
      }
      __builtin_amdgcn_wave_barrier();
      double g_s = 0.0, l_s = 0.0, g_r = 0.0, l_r = 0.0;
      {
      for (int j = 0; j < NE; ++j) {
        const double nj = bcast(ni, j), qj = bcast(qi, j);
        const int row = on ? j * NE + lane : 0;
        if (use_s) {
          const double K = on ? (STAGED ? lds[row] : ks[row]) : 0.0;
          const int d = STAGED ? ilds[o_idd + row] : t.idx_diff[row];
          const int sg = on ? -(int)(STAGED ? blds[o_sg + row] : t.sign[row]) : 0;     // sign(E_lane - E_j)
          const double P = lds[o_wave + d];
          g_s = fma(K * (sg < 0 ? 1.0 + P : P), nj, g_s);          // K^s_eff[j][i] n_j
          l_s = fma(K * (sg > 0 ? 1.0 + P : P), qj, l_s);          // K^s_eff[i][j] q_j
          if (do_ph && sg != 0) {
            const int slot = (sg > 0 ? o_A : o_B) + d;
            const double v = dE * (ni * K * qj);
            if (ATOMIC) atomicAdd(&lds[slot], v); else lds[slot] += v;
          }
        }
        if (use_r) {
          const double K = on ? (STAGED ? lds[o_kr + row] : kr[row]) : 0.0;
          const int s = STAGED ? ilds[o_ids + row] : t.idx_sum[row];
          const double P = lds[o_wave + s];
          l_r = fma(K * (1.0 + P), nj, l_r);
          g_r = fma(K * P, qj, g_r);
          if (do_ph && on) {
            const double va = dE * (ni * K * nj), vb = dE * (qi * K * qj);
            if (ATOMIC) { atomicAdd(&lds[o_A + s], va); atomicAdd(&lds[o_B + s], vb); }
            else { lds[o_A + s] += va; lds[o_B + s] += vb; }
          }
        }
      }
      }
      n[k] = relax_update_w(ni, dE * qi * g_s + 2.0 * dE * qi * g_r, dE * l_s + 2.0 * dE * l_r, dt);
      __builtin_amdgcn_wave_barrier();
      if (do_ph) {
#pragma unroll
        for (int s = 0; s < MAXBINS; ++s) {
          const int w = lane + 64 * s;
          if (w < NW) pb[s][k] = affine_update_w(pb[s][k], lds[o_A + w], lds[o_A + w] - lds[o_B + w], dt);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    }
    if (on) {
#pragma unroll
      for (int k = 0; k < PB; ++k)
        if (k < npx) sout[(long)lane * ncell + p0 + k] = n[k];
    }
    if (do_ph) {
#pragma unroll
      for (int s = 0; s < MAXBINS; ++s) {
        const int w = lane + 64 * s;
        if (w < NW) {
#pragma unroll
          for (int k = 0; k < PB; ++k)
            if (k < npx) ph[(long)w * ncell + p0 + k] = pb[s][k];
        }
      }
    }
  }
}

// returns false when the shape is outside this kernel's range
bool collision_wave_dispatch(const WaveCollView& v, bool structured, const uint8_t* flags, long ncell, const double* sin_,
                             double* sout, double* ph, double dE, double dt, int en_r, int en_s, int upd,
                             hipStream_t stream) {
  if (v.ne > 64 || v.nw > 64 * MAXBINS) return false;
  const long need = (ncell + (long)PB * WAVES - 1) / ((long)PB * WAVES);
  const size_t per_wave = (size_t)3 * v.nw * sizeof(double);
  const size_t table_bytes = (((size_t)25 * v.ne * v.ne + 7) / 8) * 8;
  const bool staged = v.nclass == 1 && table_bytes + WAVES * per_wave <= 150 * 1024;
  const size_t shmem = (staged ? table_bytes : 0) + WAVES * per_wave;
  // persistent blocks: enough to fill 256 CUs at the LDS-limited residency, never more than the work
  const long resident = 256L * std::max<long>(1, (long)(160 * 1024 / std::max<size_t>(shmem, 1)));
  const unsigned blocks = (unsigned)std::min<long>(need, std::min<long>(resident, 256L * 8));
  // more than 64 KiB of dynamic LDS has to be allowed explicitly (gfx950 has 160 KiB per CU)
#define QP_LAUNCH(AT, ST, SR)                                                                                          \
  do {                                                                                                                 \
    if (shmem > 48 * 1024)                                                                                             \
      (void)hipFuncSetAttribute((const void*)collision_wave_kernel<AT, ST, SR>,                                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                               \
    hipLaunchKernelGGL((collision_wave_kernel<AT, ST, SR>), dim3(blocks), dim3(64 * WAVES), shmem, stream, v, flags,   \
                       ncell, sin_, sout, ph, dE, dt, en_r, en_s, upd);                                                \
  } while (0)
  if (structured) { if (staged) QP_LAUNCH(false, true, true); else QP_LAUNCH(false, false, true); }
  else { if (staged) QP_LAUNCH(true, true, false); else QP_LAUNCH(true, false, false); }
#undef QP_LAUNCH
  return true;
}

}  // namespace qp
