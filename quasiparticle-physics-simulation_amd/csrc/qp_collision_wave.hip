// Collision update with one wave per pixel, lane <-> energy bin (NE <= 64, any gap-class map, any phonon-bin map).
//
// The register kernel of qp_collision_fast.hip keeps 4 NE doubles per thread and unrolls NE^2 pairs; beyond NE = 16 (the
// reference's default is NE = 50) that neither fits nor compiles in reasonable time.  Here a wave works on one pixel at a
// time: lane i owns n_i, q_i and the gain / loss sums of bin i; the loop over j broadcasts n_j, q_j with v_readlane
// (scalar operands), lane i does the pair (i, j).  The pixel's phonon occupations and the per-bin sums live in LDS
// (3 NW doubles per wave), so the relaxation / growth updates cost one exp per lane and per 64 bins instead of NE + NW
// per thread.  A wave takes 8 consecutive pixels so that every lane reads / writes 64 contiguous bytes of its planes.
//
// K^s_0, K^r_0, idx_diff and idx_sum are symmetric and sign is antisymmetric, so row j of each table (contiguous over
// lanes) serves the pair (lane, j); the rows are read from global memory (texture path), which keeps them off the LDS
// instruction pipe - that pipe (~8 cycles per ds instruction per CU) is what bounds this kernel.  Measured alternatives
// that were slower on MI355X, NE = 50: tables staged in LDS (+18 %), cross-lane ds_bpermute gathers instead of the
// per-bin read-modify-write (+25 %), four pixels interleaved in the j-loop (+75 %, one wave per SIMD).
// Per-bin sums: for a fixed j the lanes of one instruction hit distinct bins when the maps have the |i-j| / i+j structure
// (emission and absorption go to different arrays), so a plain read-modify-write by the single wave is race-free and
// deterministic; otherwise (unstructured maps) LDS atomics (ds_add_f64) are used.
// Every shared access is written as lds[integer index]: pointer selects or integer round-trips of the base make the
// compiler fall back to flat (generic address space) accesses.
#include "qp_common.h"

namespace qp {

struct WaveCollView {
  int ne, nw, nclass;
  const double* kr0;
  const double* ks0;
  const double* rho;
  const int32_t* idx_diff;
  const int32_t* idx_sum;
  const int8_t* sign;
  const int32_t* cls;
  const int32_t* diag_bin;   // non-NULL: the host vouches for the |i-j| / i+j structure of the maps
  const int32_t* anti_bin;
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

constexpr int PB = 8;        // pixels per wave
constexpr int MAXBINS = 3;   // phonon bins per lane: NW <= 3*64 - 1 for NE <= 64
constexpr int WAVES = 4;     // waves per block; they never synchronise with each other

template <bool ATOMIC>
__global__ void __launch_bounds__(64 * WAVES) collision_wave_kernel(WaveCollView t, const uint8_t* __restrict__ flags,
                                                                   long ncell, const double* __restrict__ sin_,
                                                                   double* __restrict__ sout, double* __restrict__ ph,
                                                                   double dE, double dt, int en_r, int en_s, int upd_ph) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NE = t.ne, NW = t.nw, NN = NE * NE;
  const bool use_s = en_s && t.ks0, use_r = en_r && t.kr0;
  const bool do_ph = upd_ph && (use_s || use_r);
  const int o_P = wave * 3 * NW;        // phonon occupations of the current pixel
  const int o_A = o_P + NW;             // sum of "a" terms (emission, recombination)
  const int o_B = o_P + 2 * NW;         // sum of the negative "b" terms (absorption, pair breaking): b = A - B
  const long p0 = ((long)blockIdx.x * WAVES + wave) * PB;
  if (p0 >= ncell) return;
  const int npx = (int)min((long)PB, ncell - p0);
  const bool on = lane < NE;

  double n[PB], pb[MAXBINS][PB];
#pragma unroll
  for (int k = 0; k < PB; ++k) n[k] = (on && k < npx) ? sin_[(long)lane * ncell + p0 + k] : 0.0;
#pragma unroll
  for (int s = 0; s < MAXBINS; ++s) {
    const int w = lane + 64 * s;
#pragma unroll
    for (int k = 0; k < PB; ++k) pb[s][k] = (w < NW && k < npx) ? ph[(long)w * ncell + p0 + k] : 0.0;
  }

#pragma unroll
  for (int k = 0; k < PB; ++k) {
    if (k >= npx) break;
    if (!(flags[p0 + k] & QP_FLAG_ACTIVE)) continue;      // wave-uniform: holes pass through unchanged
    const int c = t.cls ? t.cls[p0 + k] : 0;
    const double rho_i = on ? t.rho[(long)c * NE + lane] : 0.0;
    const double* ks = use_s ? t.ks0 + (long)c * NN : nullptr;
    const double* kr = use_r ? t.kr0 + (long)c * NN : nullptr;
    const double ni = n[k];
    const double qi = rho_i * fmax(1.0 - ni / fmax(rho_i, 1e-30), 0.0);
#pragma unroll
    for (int s = 0; s < MAXBINS; ++s) {
      const int w = lane + 64 * s;
      if (w < NW) { lds[o_P + w] = pb[s][k]; lds[o_A + w] = 0.0; lds[o_B + w] = 0.0; }
    }
    __builtin_amdgcn_wave_barrier();      // LDS accesses of one wave execute in order: only code motion must be fenced
    double g_s = 0.0, l_s = 0.0, g_r = 0.0, l_r = 0.0;
    for (int j = 0; j < NE; ++j) {
      const double nj = bcast(ni, j), qj = bcast(qi, j);
      const int row = on ? j * NE + lane : 0;
      if (use_s) {
        const double K = on ? ks[row] : 0.0;
        const int d = t.idx_diff[row];
        const int sg = on ? -(int)t.sign[row] : 0;                 // sign(E_lane - E_j)
        const double P = lds[o_P + d];
        g_s = fma(K * (sg < 0 ? 1.0 + P : P), nj, g_s);            // K^s_eff[j][i] n_j
        l_s = fma(K * (sg > 0 ? 1.0 + P : P), qj, l_s);            // K^s_eff[i][j] q_j
        if (do_ph && sg != 0) {
          const int slot = (sg > 0 ? o_A : o_B) + d;
          const double v = dE * (ni * K * qj);
          if (ATOMIC) atomicAdd(&lds[slot], v); else lds[slot] += v;
        }
      }
      if (use_r) {
        const double K = on ? kr[row] : 0.0;
        const int s = t.idx_sum[row];
        const double P = lds[o_P + s];
        l_r = fma(K * (1.0 + P), nj, l_r);
        g_r = fma(K * P, qj, g_r);
        if (do_ph && on) {
          const double va = dE * (ni * K * nj), vb = dE * (qi * K * qj);
          if (ATOMIC) { atomicAdd(&lds[o_A + s], va); atomicAdd(&lds[o_B + s], vb); }
          else { lds[o_A + s] += va; lds[o_B + s] += vb; }
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

// returns false when the shape is outside this kernel's range
bool collision_wave_dispatch(const WaveCollView& v, bool structured, const uint8_t* flags, long ncell, const double* sin_,
                             double* sout, double* ph, double dE, double dt, int en_r, int en_s, int upd,
                             hipStream_t stream) {
  if (v.ne > 64 || v.nw > 64 * MAXBINS) return false;
  const unsigned blocks = (unsigned)((ncell + (long)PB * WAVES - 1) / ((long)PB * WAVES));
  const size_t shmem = (size_t)WAVES * 3 * v.nw * sizeof(double);
  if (structured)
    hipLaunchKernelGGL(collision_wave_kernel<false>, dim3(blocks), dim3(64 * WAVES), shmem, stream, v, flags, ncell, sin_,
                       sout, ph, dE, dt, en_r, en_s, upd);
  else
    hipLaunchKernelGGL(collision_wave_kernel<true>, dim3(blocks), dim3(64 * WAVES), shmem, stream, v, flags, ncell, sin_,
                       sout, ph, dE, dt, en_r, en_s, upd);
  return true;
}

}  // namespace qp
