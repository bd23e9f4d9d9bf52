// Dispatcher of the one-pass collision kernels (qp_collision_onepass.inc); the instantiations live in
// qp_collision_onepass_u*.hip, one (NE, process combination) per unit so that the build parallelises.
#include <stdlib.h>

#include "qp_collision_onepass.inc"

namespace qp {

QP_DECLARE_ONEPASS(50)
QP_DECLARE_ONEPASS(40)
QP_DECLARE_ONEPASS(32)
QP_DECLARE_ONEPASS(30)

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

int collision_onepass_supported(int ne) { return (ne == 50 || ne == 40 || ne == 32 || ne == 30) ? 1 : 0; }

}  // namespace qp
