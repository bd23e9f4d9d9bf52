// Register collision kernels, NE = 30, 32 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(30)
QP_DEFINE_DIAG(32)
QP_DEFINE_DIAGP(30)
QP_DEFINE_DIAGP(32)
}  // namespace qp
