// Dispatcher of the one-pass collision kernels (qp_collision_onepass.inc); the instantiations live in
// qp_collision_onepass_u*.hip, one (NE, process combination) per unit so that the build parallelises.
#include <stdlib.h>

#include "qp_collision_onepass.inc"

namespace qp {

QP_DECLARE_ONEPASS(50)
QP_DECLARE_ONEPASS(40)
QP_DECLARE_ONEPASS(32)
QP_DECLARE_ONEPASS(30)
QP_DECLARE_ONEPASS_CLASSES(50, 1, 1)
QP_DECLARE_ONEPASS_CLASSES(50, 0, 1)
QP_DECLARE_ONEPASS_CLASSES(50, 1, 0)

static bool onepass_enabled() {      // QPSIM_COLL_ONEPASS=0: the three-launch split kernels (A/B timing, tests)
  const char* e = getenv("QPSIM_COLL_ONEPASS");
  return !e || atoi(e) != 0;
}

// One gap class.  False when this (NE, processes) has no one-pass instantiation, a needed table is missing, or the kernel
// family is switched off: the caller then takes the split kernels.
bool collision_onepass_dispatch(const qp_collision_tables& tb, double* stash, const uint8_t* flags, long ncell,
                                const double* sin_, double* sout, double* ph, double dE, double dt, bool s, bool r, bool u,
                                hipStream_t stream) {
  if (!onepass_enabled() || tb.nclass != 1 || !(s || r)) return false;
  if ((s && !tb.ks0_diag) || (r && !tb.kr0_anti2)) return false;
  OnePassView v{};
  v.base = CollFastView{tb.kr0, tb.ks0, tb.rho, tb.diag_bin, tb.anti_bin, stash, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, 0.0};
  v.ksd = tb.ks0_diag;
  v.kra2 = tb.kr0_anti2;
  onepass_launcher_t fn = nullptr;
  switch (tb.ne) {
    case 50: fn = (s && r) ? onepass_50_11 : r ? onepass_50_01 : onepass_50_10; break;
    case 40: fn = (s && r) ? onepass_40_11 : r ? onepass_40_01 : onepass_40_10; break;
    case 32: fn = (s && r) ? onepass_32_11 : r ? onepass_32_01 : onepass_32_10; break;
    case 30: fn = (s && r) ? onepass_30_11 : r ? onepass_30_01 : onepass_30_10; break;
    default: return false;
  }
  fn(v, flags, ncell, sin_, sout, ph, dE, dt, u, stream);
  return true;
}

// Gap classes (nclass > 1) with the separable kernel tables (gap_sq, kr_amp, ks_amp, pair_inv): the same kernel with K formed
// per lane.  False when this (NE, processes) has no instantiation or a table is missing: the caller takes the split kernels.
bool collision_onepass_dispatch_classes(const qp_collision_tables& tb, double* stash, const uint8_t* flags, long ncell,
                                        const double* sin_, double* sout, double* ph, double dE, double dt, bool s, bool r,
                                        bool u, hipStream_t stream) {
  if (!onepass_enabled() || tb.nclass < 2 || tb.nclass > kOnePassMaxClasses || !(s || r)) return false;
  if (!tb.cls || !tb.gap_sq || !tb.pair_inv || (s && !tb.ks_amp) || (r && !tb.kr_amp)) return false;
  onepass_launcher_t fn = nullptr;
  switch (tb.ne) {
    case 50:
      fn = (s && r) ? (u ? onepass_classes_50_11_u1 : onepass_classes_50_11_u0)
           : r      ? (u ? onepass_classes_50_01_u1 : onepass_classes_50_01_u0)
                    : (u ? onepass_classes_50_10_u1 : onepass_classes_50_10_u0);
      break;
    default: break;
  }
  if (!fn) return false;
  OnePassView v{};
  v.base = CollFastView{nullptr, nullptr, tb.rho, tb.diag_bin, tb.anti_bin, stash, tb.cls, tb.gap_sq, r ? tb.kr_amp : nullptr,
                        s ? tb.ks_amp : nullptr, tb.pair_inv, nullptr, 0.0};
  v.nclass = tb.nclass;
  fn(v, flags, ncell, sin_, sout, ph, dE, dt, u, stream);
  return true;
}

int collision_onepass_supported(int ne) { return (ne == 50 || ne == 40 || ne == 32 || ne == 30) ? 1 : 0; }

}  // namespace qp
