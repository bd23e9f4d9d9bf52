// Dispatcher of the register-resident collision kernels (qp_collision_fast.inc) + the instantiations for NE <= 16.
// Larger NE live in qp_collision_fast_u*.hip; every NE listed in QP_DIAG_NE_LIST has all process combinations.
#include "qp_collision_fast.inc"

#define QP_DIAG_NE_LIST(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(18) X(20) X(24) X(30) X(32) X(40) X(50)

namespace qp {

QP_DEFINE_DIAG(2) QP_DEFINE_DIAG(3) QP_DEFINE_DIAG(4) QP_DEFINE_DIAG(5) QP_DEFINE_DIAG(6) QP_DEFINE_DIAG(7) QP_DEFINE_DIAG(8) QP_DEFINE_DIAG(9) QP_DEFINE_DIAG(10) QP_DEFINE_DIAG(11) QP_DEFINE_DIAG(12) QP_DEFINE_DIAG(13) QP_DEFINE_DIAG(14) QP_DEFINE_DIAG(15) QP_DEFINE_DIAG(16)

QP_DIAG_NE_LIST(QP_DECLARE_DIAG)

// gap-class (PARAM) variants: the single-pass sizes of this unit
#define QP_DIAGP_NE_LIST(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
QP_DIAGP_NE_LIST(QP_DEFINE_DIAGP)
// ... and the single-pass sizes of the other units
#define QP_DIAGP_NE_LIST_EXT(X) X(18) X(20) X(24) X(30) X(32) X(40) X(50)
#define QP_DECLARE_DIAGP(N)                                                                                               \
  void diag_launcherp_##N##_11(const CollFastView&, const uint8_t*, long, const double*, double*, double*, double, double, \
                               bool, hipStream_t);                                                                         \
  void diag_launcherp_##N##_01(const CollFastView&, const uint8_t*, long, const double*, double*, double*, double, double, \
                               bool, hipStream_t);                                                                         \
  void diag_launcherp_##N##_10(const CollFastView&, const uint8_t*, long, const double*, double*, double*, double, double, \
                               bool, hipStream_t);
QP_DIAGP_NE_LIST_EXT(QP_DECLARE_DIAGP)

__global__ void __launch_bounds__(256) collision_none_kernel(const uint8_t* __restrict__ flags, long ncell, long total,
                                                             const double* __restrict__ sin_, double* __restrict__ sout) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const double v = sin_[t];
  sout[t] = (flags[t % ncell] & QP_FLAG_ACTIVE) ? fmax(v, 0.0) : v;
}

// returns false when NE has no instantiation or the cell count exceeds the 32-bit offset range
// `guard` (may be NULL): per-wave partials of the fused Pauli guard; *guard_done tells whether the kernels wrote them (the
// single-pass sizes do, the split kernels of NE >= 32 and the no-process copy do not)
bool collision_fast_dispatch(int ne, const double* kr0, const double* ks0, const double* rho, const int* diag_bin,
                             const int* anti_bin, double* stash, const uint8_t* flags, long ncell, const double* sin_,
                             double* sout, double* ph, double dE, double dt, int en_r, int en_s, int upd,
                             PauliPartial* guard, double guard_floor, bool* guard_done, hipStream_t stream) {
  if (guard_done) *guard_done = false;
  if (ncell >= (1L << 28)) return false;
  CollFastView v{kr0, ks0, rho, diag_bin, anti_bin, stash, nullptr, nullptr, nullptr, nullptr, nullptr,
                 ne < 32 ? guard : nullptr, guard_floor};
  const bool s = en_s && ks0, r = en_r && kr0, u = upd && (s || r);
  diag_launcher_t fn = nullptr;
  switch (ne) {
#define QP_CASE(N) case N: fn = (s && r) ? diag_launcher_##N##_11 : r ? diag_launcher_##N##_01 : diag_launcher_##N##_10; break;
    QP_DIAG_NE_LIST(QP_CASE)
#undef QP_CASE
    default: return false;
  }
  if (!s && !r) {   // no process enabled: relaxation with zero gain and loss, n' = max(n, 0) on active cells
    const long total = ncell * ne;
    hipLaunchKernelGGL(collision_none_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, flags, ncell,
                       total, sin_, sout);
    return true;
  }
  fn(v, flags, ncell, sin_, sout, ph, dE, dt, u, stream);
  if (guard_done) *guard_done = v.guard != nullptr;
  return true;
}

// Gap classes: `rho` is [nclass][ne]; kr_amp / ks_amp stand in for kr0 / ks0 (NULL = process off).  False when this NE has
// no PARAM instantiation.
bool collision_fast_dispatch_classes(int ne, const double* rho, const int* cls, const double* gap_sq, const double* kr_amp,
                                     const double* ks_amp, const double* pair_inv, const int* diag_bin, const int* anti_bin,
                                     double* stash, const uint8_t* flags, long ncell, const double* sin_, double* sout,
                                     double* ph, double dE, double dt, int en_r, int en_s, int upd, PauliPartial* guard,
                                     double guard_floor, bool* guard_done, hipStream_t stream) {
  if (guard_done) *guard_done = false;
  if (ncell >= (1L << 28)) return false;
  CollFastView v{nullptr, nullptr, rho, diag_bin, anti_bin, stash, cls, gap_sq, kr_amp, ks_amp, pair_inv,
                 ne < 32 ? guard : nullptr, guard_floor};
  const bool s = en_s && ks_amp, r = en_r && kr_amp, u = upd && (s || r);
  if (!s && !r) return false;
  diag_launcher_t fn = nullptr;
  switch (ne) {
#define QP_CASE(N) case N: fn = (s && r) ? diag_launcherp_##N##_11 : r ? diag_launcherp_##N##_01 : diag_launcherp_##N##_10; break;
    QP_DIAGP_NE_LIST(QP_CASE)
    QP_DIAGP_NE_LIST_EXT(QP_CASE)
#undef QP_CASE
    default: return false;
  }
  fn(v, flags, ncell, sin_, sout, ph, dE, dt, u, stream);
  if (guard_done) *guard_done = v.guard != nullptr;
  return true;
}

// 1 when the gap-class (PARAM) variant exists for this NE
int collision_fast_classes_supported(int ne) {
  switch (ne) {
#define QP_CASE(N) case N: return 1;
    QP_DIAGP_NE_LIST(QP_CASE)
    QP_DIAGP_NE_LIST_EXT(QP_CASE)
#undef QP_CASE
    default: return 0;
  }
}

// list of NE with a register kernel (for the host-side choice / tests)
int collision_fast_supported(int ne) {
  switch (ne) {
#define QP_CASE(N) case N: return 1;
    QP_DIAG_NE_LIST(QP_CASE)
#undef QP_CASE
    default: return 0;
  }
}

}  // namespace qp
