// Instantiations + dispatcher of the double half-step collision kernel (qp_collision_pair.inc): NE = 4 ... 16.
#include "qp_collision_pair.inc"

namespace qp {

#define QP_PAIR_NE_LIST(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
QP_PAIR_NE_LIST(QP_DEFINE_PAIR)

int collision_pair_supported(int ne) { return (ne >= 4 && ne <= 16) ? 1 : 0; }

// false: no fused kernel for these tables (NE, gap classes, merged phonon bins, cell count) - the caller runs two calls
bool collision_pair_dispatch(const qp_collision_tables& tb, const uint8_t* flags, long ncell, const double* sin_, double* sout,
                             double* ph, double dE, double dt_a, double dt_b, double gen, bool s, bool r, bool u,
                             PauliPartial* guard, double guard_floor, hipStream_t stream) {
  if (!collision_pair_supported(tb.ne) || tb.nclass != 1 || !tb.diag_bin || !tb.anti_bin || !(s || r)) return false;
  if ((tb.flags & (QP_COLL_FORCE_GENERIC | QP_COLL_FORCE_WAVE | QP_COLL_SHARED_BINS)) || ncell >= (1L << 28)) return false;
  CollFastView v{tb.kr0, tb.ks0, tb.rho, tb.diag_bin, tb.anti_bin, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                 guard, guard_floor};
  pair_launcher_t fn = nullptr;
  switch (tb.ne) {
#define QP_CASE(N) case N: fn = (s && r) ? pair_launcher_##N##_11 : r ? pair_launcher_##N##_01 : pair_launcher_##N##_10; break;
    QP_PAIR_NE_LIST(QP_CASE)
#undef QP_CASE
    default: return false;
  }
  fn(v, flags, ncell, sin_, sout, ph, dE, dt_a, dt_b, gen, u, stream);
  return true;
}

}  // namespace qp
