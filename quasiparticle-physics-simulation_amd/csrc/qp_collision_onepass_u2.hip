// One-pass collision kernel, NE = 50, single-process combinations.
#include "qp_collision_onepass.inc"

namespace qp {
QP_DEFINE_ONEPASS(50, 14, 0, 1, 8, 2)
QP_DEFINE_ONEPASS(50, 14, 1, 0, 8, 2)
}  // namespace qp
