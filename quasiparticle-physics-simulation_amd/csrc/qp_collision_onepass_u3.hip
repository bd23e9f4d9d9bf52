// One-pass collision kernel, NE = 40 (all process combinations).
#include "qp_collision_onepass.inc"

namespace qp {
QP_DEFINE_ONEPASS(40, 14, 1, 1, 8, 2)
QP_DEFINE_ONEPASS(40, 14, 0, 1, 8, 2)
QP_DEFINE_ONEPASS(40, 14, 1, 0, 8, 2)
}  // namespace qp
