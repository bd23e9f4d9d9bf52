// Register collision kernels, NE = 18, 20 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(18)
QP_DEFINE_DIAG(20)
QP_DEFINE_DIAGP(18)
QP_DEFINE_DIAGP(20)
}  // namespace qp
