// Register collision kernels, NE = 24 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(24)
QP_DEFINE_DIAGP(24)
}  // namespace qp
