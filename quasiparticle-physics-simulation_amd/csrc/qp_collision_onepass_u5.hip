// One-pass collision kernel, NE = 30 (all process combinations).  NE = 24 was measured too: 606 us per 1024^2 call against 506 us
// of the single-pass register kernel - not instantiated.
#include "qp_collision_onepass.inc"

namespace qp {
QP_DEFINE_ONEPASS(30, 16, 1, 1, 8, 2)
QP_DEFINE_ONEPASS(30, 16, 0, 1, 8, 2)
QP_DEFINE_ONEPASS(30, 16, 1, 0, 8, 2)
}  // namespace qp
