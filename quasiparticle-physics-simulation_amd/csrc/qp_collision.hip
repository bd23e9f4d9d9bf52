// Local coupled quasiparticle-phonon collision update (reference solver.py:703-791), one thread per cell.
//
// Generic table-driven kernel: any NE, any phonon-bin map, per-cell gap classes.  State planes are
// [bin][cell], so every load/store of a plane element is coalesced across the wave.  The per-cell phonon
// accumulators a, b live in a scratch plane set [2][nw][ncell] (each thread only touches its own column).
#include "qp_common.h"

namespace qp {

struct CollView {
  int ne, nw, nclass;
  const double* kr0;
  const double* ks0;
  const double* rho;
  const int32_t* idx_diff;
  const int32_t* idx_sum;
  const int8_t* sign;
  const int32_t* cls;
};

// n(t+dt) for dn/dt = gain - loss n with frozen coefficients (solver.py:640-665)
__device__ __forceinline__ double relax_update(double n, double gain, double loss, double dt) {
  const double mu = fmax(loss, 0.0);
  const double P = fmax(gain + (mu - loss) * n, 0.0);
  const double decay = exp(-mu * dt);
  const double coeff = (mu < 1e-14) ? dt : (1.0 - decay) / mu;
  return fmax(decay * n + coeff * P, 0.0);
}

// y(t+dt) for y' = a + b y with frozen coefficients (solver.py:686-700)
__device__ __forceinline__ double affine_update(double y, double a, double b, double dt) {
  const double xx = fmin(fmax(b * dt, -80.0), 80.0);
  const double ex = exp(xx);
  const double coeff = (fabs(b) < 1e-14) ? dt : (ex - 1.0) / b;
  return fmax(ex * y + coeff * a, 0.0);
}

__global__ void __launch_bounds__(256) collision_generic_kernel(CollView t, const uint8_t* __restrict__ flags,
                                                                long ncell, const double* __restrict__ sin_,
                                                                double* __restrict__ sout, double* __restrict__ ph,
                                                                double* __restrict__ acc, double dE, double dt,
                                                                int en_r, int en_s, int upd_ph) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ncell) return;
  const int NE = t.ne, NW = t.nw;
  if (!(flags[p] & QP_FLAG_ACTIVE)) {
    for (int i = 0; i < NE; ++i) sout[(long)i * ncell + p] = sin_[(long)i * ncell + p];
    return;
  }
  const int c = t.cls ? t.cls[p] : 0;
  const double* rho = t.rho + (long)c * NE;
  const double* kr = t.kr0 ? t.kr0 + (long)c * NE * NE : nullptr;
  const double* ks = t.ks0 ? t.ks0 + (long)c * NE * NE : nullptr;
  const bool use_r = en_r && kr;
  const bool use_s = en_s && ks;

  // ---- quasiparticle update: gain / loss rates from the OLD n and p ----
  for (int i = 0; i < NE; ++i) {
    const double ni = sin_[(long)i * ncell + p];
    const double rho_i = rho[i];
    const double qi = rho_i * fmax(1.0 - ni / fmax(rho_i, 1e-30), 0.0);
    double g_s = 0.0, l_s = 0.0, g_r = 0.0, l_r = 0.0;
    for (int j = 0; j < NE; ++j) {
      const double nj = sin_[(long)j * ncell + p];
      const double rho_j = rho[j];
      const double qj = rho_j * fmax(1.0 - nj / fmax(rho_j, 1e-30), 0.0);
      if (use_s && j != i) {
        // the reference's own maps are symmetric in (i, j); caller-supplied tables of the step API need not be
        const double pd_ij = ph[(long)t.idx_diff[i * NE + j] * ncell + p];
        const double pd_ji = ph[(long)t.idx_diff[j * NE + i] * ncell + p];
        const double np_ij = t.sign[i * NE + j] > 0 ? 1.0 + pd_ij : pd_ij;
        const double np_ji = t.sign[j * NE + i] > 0 ? 1.0 + pd_ji : pd_ji;
        g_s += ks[j * NE + i] * np_ji * nj;
        l_s += ks[i * NE + j] * np_ij * qj;
      }
      if (use_r) {
        const double ps = ph[(long)t.idx_sum[i * NE + j] * ncell + p];
        const double k = kr[i * NE + j];
        l_r += k * (1.0 + ps) * nj;
        g_r += k * ps * qj;
      }
    }
    const double gain = dE * qi * g_s + 2.0 * dE * qi * g_r;
    const double loss = dE * l_s + 2.0 * dE * l_r;
    sout[(long)i * ncell + p] = relax_update(ni, gain, loss, dt);
  }
  if (!upd_ph || !(en_r || en_s)) return;

  // ---- phonon update: bin the pair rates (bincount of solver.py:758-788) ----
  double* A = acc;
  double* B = acc + (long)NW * ncell;
  for (int w = 0; w < NW; ++w) {
    A[(long)w * ncell + p] = 0.0;
    B[(long)w * ncell + p] = 0.0;
  }
  for (int i = 0; i < NE; ++i) {
    const double ni = sin_[(long)i * ncell + p];
    const double rho_i = rho[i];
    const double qi = rho_i * fmax(1.0 - ni / fmax(rho_i, 1e-30), 0.0);
    for (int j = 0; j < NE; ++j) {
      const double nj = sin_[(long)j * ncell + p];
      const double rho_j = rho[j];
      const double qj = rho_j * fmax(1.0 - nj / fmax(rho_j, 1e-30), 0.0);
      if (use_s && j != i) {
        const long w = (long)t.idx_diff[i * NE + j] * ncell + p;
        const double base = dE * (ni * ks[i * NE + j] * qj);
        const int sg = t.sign[i * NE + j];
        if (sg > 0) {
          A[w] += base;
          B[w] += base;
        } else if (sg < 0) {
          B[w] -= base;
        }
      }
      if (use_r) {
        const long w = (long)t.idx_sum[i * NE + j] * ncell + p;
        const double k = kr[i * NE + j];
        const double rec = dE * (ni * k * nj);
        const double pb = dE * (qi * k * qj);
        A[w] += rec;
        B[w] += rec - pb;
      }
    }
  }
  for (int w = 0; w < NW; ++w) {
    const long o = (long)w * ncell + p;
    ph[o] = affine_update(ph[o], A[o], B[o], dt);
  }
}

bool collision_fast_dispatch(int ne, const double* kr0, const double* ks0, const double* rho, const int* diag_bin,
                             const int* anti_bin, double* stash, const uint8_t* flags, long ncell, const double* sin_,
                             double* sout, double* ph, double dE, double dt, int en_r, int en_s, int upd,
                             PauliPartial* guard, double guard_floor, bool* guard_done, hipStream_t stream);

bool collision_onepass_dispatch(const qp_collision_tables& tb, double* stash, const uint8_t* flags, long ncell,
                                const double* sin_, double* sout, double* ph, double dE, double dt, bool s, bool r, bool u,
                                hipStream_t stream);

bool collision_onepass_dispatch_classes(const qp_collision_tables& tb, double* stash, const uint8_t* flags, long ncell,
                                        const double* sin_, double* sout, double* ph, double dE, double dt, bool s, bool r,
                                        bool u, hipStream_t stream);
int collision_onepass_supported(int ne);
int collision_pair_supported(int ne);
bool collision_pair_dispatch(const qp_collision_tables& tb, const uint8_t* flags, long ncell, const double* sin_, double* sout,
                             double* ph, double dE, double dt_a, double dt_b, double gen, bool s, bool r, bool u,
                             PauliPartial* guard, double guard_floor, hipStream_t stream);
int collision_fast_supported(int ne);
int collision_fast_classes_supported(int ne);
bool collision_fast_dispatch_classes(int ne, const double* rho, const int* cls, const double* gap_sq, const double* kr_amp,
                                     const double* ks_amp, const double* pair_inv, const int* diag_bin, const int* anti_bin,
                                     double* stash, const uint8_t* flags, long ncell, const double* sin_, double* sout,
                                     double* ph, double dE, double dt, int en_r, int en_s, int upd, PauliPartial* guard,
                                     double guard_floor, bool* guard_done, hipStream_t stream);

struct WaveCollView {
  int ne, nw, nclass;
  const double* kr0;
  const double* ks0;
  const double* rho;
  const int32_t* idx_diff;
  const int32_t* idx_sum;
  const int8_t* sign;
  const int32_t* cls;
  const int32_t* diag_bin;
  const int32_t* anti_bin;
};
bool collision_wave_dispatch(const WaveCollView& v, bool structured, const uint8_t* flags, long ncell, const double* sin_,
                             double* sout, double* ph, double dE, double dt, int en_r, int en_s, int upd,
                             hipStream_t stream);

}  // namespace qp

static int collision_step_impl(const qp_collision_tables* t, const uint8_t* flags, int64_t ncell,
                               const double* state_in, double* state_out, double* phonon, double* ph_scratch,
                               double dE, double dt, int enable_recombination, int enable_scattering,
                               int update_phonons, qp::PauliPartial* guard, double guard_floor, bool* guard_done,
                               void* stream) {
  if (guard_done) *guard_done = false;
  QP_REQUIRE(t != nullptr, "tables are NULL");
  if (t->struct_size != sizeof(qp_collision_tables)) {
    qp::set_error("qp_collision_step: qp_collision_tables.struct_size is %u, this library expects %zu (binding built against "
                  "another header revision)", t->struct_size, sizeof(qp_collision_tables));
    return QP_ERR_INVALID_ARGUMENT;
  }
  QP_REQUIRE(t->ne > 0 && t->nw > 0 && t->nclass > 0, "ne, nw, nclass must be positive");
  QP_REQUIRE(t->rho && t->idx_diff && t->idx_sum && t->sign, "rho / idx maps / sign must be non-NULL");
  QP_REQUIRE(t->nclass == 1 || t->cls, "cls is required when nclass > 1");
  QP_REQUIRE(flags && state_in && state_out && phonon, "flags, state_in, state_out, phonon must be non-NULL");
  QP_REQUIRE(state_in != state_out, "state_in and state_out must not alias");
  QP_REQUIRE(ncell > 0, "ncell must be positive");
  const bool no_scratch_ok = !(t->flags & QP_COLL_FORCE_GENERIC) && t->ne <= 64 && t->nw <= 192;
  QP_REQUIRE(!(update_phonons && (enable_recombination || enable_scattering)) || ph_scratch || no_scratch_ok,
             "ph_scratch is required when phonons are updated by the generic kernel");
  QP_REQUIRE((t->diag_bin == nullptr) == (t->anti_bin == nullptr), "diag_bin and anti_bin come together");
  // merged bins (QP_COLL_SHARED_BINS): the register kernels park per-diagonal sums in ph_scratch (2 planes per merged bin);
  // without scratch only the variants that never write phonons qualify
  const bool shared_ok = !(t->flags & QP_COLL_SHARED_BINS) || ph_scratch ||
                         !(update_phonons && enable_recombination && enable_scattering && t->kr0 && t->ks0);
  // 32 <= ne <= 64 with the diagonal-major tables: one launch instead of the three of the split path
  if (t->diag_bin && t->nclass == 1 && qp::collision_onepass_supported(t->ne) && !(t->flags & (QP_COLL_FORCE_GENERIC | QP_COLL_FORCE_WAVE)) &&
      shared_ok && ncell < (1L << 28) &&
      qp::collision_onepass_dispatch(*t, ph_scratch, flags, (long)ncell, state_in, state_out, phonon, dE, dt,
                                     enable_scattering && t->ks0, enable_recombination && t->kr0,
                                     update_phonons && ((enable_scattering && t->ks0) || (enable_recombination && t->kr0)),
                                     (hipStream_t)stream))
    return qp::check_launch("qp_collision_step(one pass)");
  if (t->diag_bin && t->nclass == 1 && !(t->flags & (QP_COLL_FORCE_GENERIC | QP_COLL_FORCE_WAVE)) && shared_ok &&
      qp::collision_fast_dispatch(t->ne, t->kr0, t->ks0, t->rho, t->diag_bin, t->anti_bin, ph_scratch, flags, (long)ncell,
                                  state_in, state_out, phonon, dE, dt, enable_recombination, enable_scattering,
                                  update_phonons, guard, guard_floor, guard_done, (hipStream_t)stream))
    return qp::check_launch("qp_collision_step(fast)");
  // gap classes with the separable kernel tables: the one-pass kernel where it exists ...
  if (t->diag_bin && t->nclass > 1 && qp::collision_onepass_supported(t->ne) &&
      !(t->flags & (QP_COLL_FORCE_GENERIC | QP_COLL_FORCE_WAVE)) && shared_ok && ncell < (1L << 28) &&
      qp::collision_onepass_dispatch_classes(*t, ph_scratch, flags, (long)ncell, state_in, state_out, phonon, dE, dt,
                                             enable_scattering && t->ks0, enable_recombination && t->kr0,
                                             update_phonons && ((enable_scattering && t->ks0) || (enable_recombination && t->kr0)),
                                             (hipStream_t)stream))
    return qp::check_launch("qp_collision_step(one pass, gap classes)");
  // ... else the register kernels that form K per pixel
  if (t->diag_bin && t->nclass > 1 && t->gap_sq && t->pair_inv && t->cls &&
      (!(enable_recombination && t->kr0) || t->kr_amp) && (!(enable_scattering && t->ks0) || t->ks_amp) &&
      !(t->flags & (QP_COLL_FORCE_GENERIC | QP_COLL_FORCE_WAVE)) && shared_ok &&
      qp::collision_fast_dispatch_classes(t->ne, t->rho, t->cls, t->gap_sq, enable_recombination ? t->kr_amp : nullptr,
                                          enable_scattering ? t->ks_amp : nullptr, t->pair_inv, t->diag_bin, t->anti_bin,
                                          ph_scratch, flags, (long)ncell, state_in, state_out, phonon, dE, dt,
                                          enable_recombination, enable_scattering, update_phonons, guard, guard_floor,
                                          guard_done, (hipStream_t)stream))
    return qp::check_launch("qp_collision_step(fast, gap classes)");
  // NE <= 64: one wave per pixel (any class map; LDS atomics unless the host vouched for the bin-map structure)
  if (!(t->flags & QP_COLL_FORCE_GENERIC)) {
    qp::WaveCollView wv{t->ne, t->nw, t->nclass, t->kr0, t->ks0, t->rho, t->idx_diff, t->idx_sum, t->sign, t->cls,
                        t->diag_bin, t->anti_bin};
    if (qp::collision_wave_dispatch(wv, t->diag_bin != nullptr, flags, (long)ncell, state_in, state_out, phonon, dE, dt,
                                    enable_recombination, enable_scattering, update_phonons, (hipStream_t)stream))
      return qp::check_launch("qp_collision_step(wave)");
  }
  qp::CollView v{t->ne, t->nw, t->nclass, t->kr0, t->ks0, t->rho, t->idx_diff, t->idx_sum, t->sign, t->cls};
  const unsigned blocks = (unsigned)((ncell + 255) / 256);
  hipLaunchKernelGGL(qp::collision_generic_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, flags,
                     (long)ncell, state_in, state_out, phonon, ph_scratch, dE, dt, enable_recombination,
                     enable_scattering, update_phonons);
  return qp::check_launch("qp_collision_step");
}

extern "C" int qp_collision_step(const qp_collision_tables* t, const uint8_t* flags, int64_t ncell,
                                 const double* state_in, double* state_out, double* phonon, double* ph_scratch,
                                 double dE, double dt, int enable_recombination, int enable_scattering,
                                 int update_phonons, void* stream) {
  return collision_step_impl(t, flags, ncell, state_in, state_out, phonon, ph_scratch, dE, dt, enable_recombination,
                             enable_scattering, update_phonons, nullptr, 0.0, nullptr, stream);
}

extern "C" int64_t qp_collision_guard_workspace_bytes(int64_t ncell) {
  // the register kernels launch ceil(ncell / 128) blocks of two waves and EVERY wave writes a partial (also the second wave
  // of a last block that holds <= 64 cells), so the count follows the launch geometry, not ceil(ncell / 64)
  const int64_t waves = 2 * ((ncell + 127) / 128) + qp::kGuardMergeBlocks;
  const int64_t fused = waves * (int64_t)sizeof(qp::PauliPartial);
  const int64_t plain = qp_pauli_workspace_bytes();
  return fused > plain ? fused : plain;
}

extern "C" int qp_collision_step_guarded(const qp_collision_tables* t, const uint8_t* flags, int64_t ncell,
                                         const double* state_in, double* state_out, double* phonon, double* ph_scratch,
                                         double dE, double dt, int enable_recombination, int enable_scattering,
                                         int update_phonons, double density_floor, void* guard_workspace,
                                         double* out_vals, int64_t* out_idx, void* stream) {
  QP_REQUIRE(guard_workspace && out_vals && out_idx, "guard_workspace, out_vals, out_idx must be non-NULL");
  auto* parts = (qp::PauliPartial*)guard_workspace;
  bool done = false;
  const int rc = collision_step_impl(t, flags, ncell, state_in, state_out, phonon, ph_scratch, dE, dt, enable_recombination,
                                     enable_scattering, update_phonons, parts + qp::kGuardMergeBlocks, density_floor, &done,
                                     stream);
  if (rc != QP_OK) return rc;
  if (done) {      // one partial per wave of the register kernel (128-thread blocks): finish the reduction
    const long nparts = ((long)ncell + 127) / 128 * 2;
    qp::pauli_finish(parts + qp::kGuardMergeBlocks, nparts, parts, out_vals, (long*)out_idx, (hipStream_t)stream);
    return qp::check_launch("qp_collision_step_guarded");
  }
  // kernels without the fused epilogue (split kernels of NE >= 32, wave and generic kernels): separate pass
  return qp_pauli_stats(state_out, t->rho, t->cls, flags, t->ne, t->nclass, ncell, density_floor, guard_workspace,
                        out_vals, out_idx, stream);
}

// ---------------------------------------------------------------------------------------------------------
// Explicit (forward-Euler) fixed-bath collision helpers of the reference's public step API
// (solver.py:551-580 apply_scattering_step, :583-605 apply_recombination_step, :608-637 _collision_rhs).
// They are not called by the reference's time loop; kept for API parity.  One thread per cell, generic NE.
// ---------------------------------------------------------------------------------------------------------
namespace qp {

__global__ void __launch_bounds__(256) euler_collision_kernel(int ne, long ncell, const double* __restrict__ sin_,
                                                              double* __restrict__ out, const double* __restrict__ kr,
                                                              const double* __restrict__ g_therm,
                                                              const double* __restrict__ ks,
                                                              const double* __restrict__ rho, double dE, double dt,
                                                              int rhs_only) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ncell) return;
  for (int i = 0; i < ne; ++i) {
    const double ni = sin_[(long)i * ncell + p];
    double rhs = 0.0;
    if (kr && g_therm) {
      double acc = 0.0;
      for (int j = 0; j < ne; ++j) acc += kr[i * ne + j] * sin_[(long)j * ncell + p];
      rhs += g_therm[i] - 2.0 * ni * dE * acc;
    }
    if (ks && rho) {
      const double bi = fmax(1.0 - ni / fmax(rho[i], 1e-30), 0.0);
      double gin = 0.0, gout = 0.0;
      for (int j = 0; j < ne; ++j) {
        const double nj = sin_[(long)j * ncell + p];
        const double bj = fmax(1.0 - nj / fmax(rho[j], 1e-30), 0.0);
        gin += ks[j * ne + i] * nj;
        gout += ks[i * ne + j] * rho[j] * bj;
      }
      rhs += dE * rho[i] * bi * gin - ni * dE * gout;
    }
    out[(long)i * ncell + p] = rhs_only ? rhs : fmax(ni + dt * rhs, 0.0);
  }
}

}  // namespace qp

extern "C" int qp_euler_collision(int32_t ne, int64_t ncell, const double* state_in, double* out, const double* kr,
                                  const double* g_therm, const double* ks, const double* rho, double dE, double dt,
                                  int rhs_only, void* stream) {
  QP_REQUIRE(ne > 0 && ncell > 0 && state_in && out && state_in != out, "bad arguments (state_in and out must differ)");
  QP_REQUIRE((kr == nullptr) == (g_therm == nullptr), "kr and g_therm come together");
  QP_REQUIRE((ks == nullptr) == (rho == nullptr), "ks and rho come together");
  hipLaunchKernelGGL(qp::euler_collision_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (int)ne, (long)ncell, state_in, out, kr, g_therm, ks, rho, dE, dt, rhs_only);
  return qp::check_launch("qp_euler_collision");
}

// 1 when the register-resident collision kernel is instantiated for this number of energy bins
extern "C" int qp_collision_register_kernel_available(int32_t ne) { return qp::collision_fast_supported(ne); }

extern "C" int qp_collision_pair_available(int32_t ne) { return qp::collision_pair_supported(ne); }

extern "C" int qp_collision_double_step_guarded(const qp_collision_tables* t, const uint8_t* flags, int64_t ncell,
                                                const double* state_in, double* state_out, double* phonon, double dE,
                                                double dt_first, double dt_second, double gen_amount,
                                                int enable_recombination, int enable_scattering, int update_phonons,
                                                double density_floor, void* guard_workspace, double* out_vals,
                                                int64_t* out_idx, void* stream) {
  QP_REQUIRE(t != nullptr, "tables are NULL");
  if (t->struct_size != sizeof(qp_collision_tables)) {
    qp::set_error("qp_collision_double_step_guarded: qp_collision_tables.struct_size is %u, this library expects %zu",
                  t->struct_size, sizeof(qp_collision_tables));
    return QP_ERR_INVALID_ARGUMENT;
  }
  QP_REQUIRE(flags && state_in && state_out && phonon && state_in != state_out, "flags, state_in, state_out (distinct), phonon");
  QP_REQUIRE(guard_workspace && out_vals && out_idx, "guard_workspace, out_vals, out_idx must be non-NULL");
  QP_REQUIRE(ncell > 0 && t->rho != nullptr, "ncell must be positive, rho non-NULL");
  const bool s = enable_scattering && t->ks0, r = enable_recombination && t->kr0;
  auto* parts = (qp::PauliPartial*)guard_workspace;
  if (!qp::collision_pair_dispatch(*t, flags, (long)ncell, state_in, state_out, phonon, dE, dt_first, dt_second, gen_amount, s, r,
                                   update_phonons && (s || r), parts + qp::kGuardMergeBlocks, density_floor,
                                   (hipStream_t)stream)) {
    qp::set_error("qp_collision_double_step_guarded: no fused double half-step kernel for these tables (ne = %d)", t->ne);
    return QP_ERR_UNSUPPORTED;
  }
  const long nparts = ((long)ncell + 127) / 128 * 2;
  qp::pauli_finish(parts + qp::kGuardMergeBlocks, nparts, parts, out_vals, (long*)out_idx, (hipStream_t)stream);
  return qp::check_launch("qp_collision_double_step_guarded");
}

extern "C" int qp_collision_onepass_available(int32_t ne) { return qp::collision_onepass_supported(ne); }

// 2nd bit: the gap-class variant (separable kernel tables, qp_collision_tables::gap_sq ...) exists as well
extern "C" int qp_collision_register_kernel_classes(int32_t ne) { return qp::collision_fast_classes_supported(ne); }
