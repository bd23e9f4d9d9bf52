// One-pass collision kernel, NE = 32 (all process combinations).
#include "qp_collision_onepass.inc"

namespace qp {
QP_DEFINE_ONEPASS(32, 16, 1, 1, 8, 2)
QP_DEFINE_ONEPASS(32, 16, 0, 1, 8, 2)
QP_DEFINE_ONEPASS(32, 16, 1, 0, 8, 2)
}  // namespace qp
