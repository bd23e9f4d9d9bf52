// Register collision kernels, NE = 17, 18, 19, 20 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(17)
QP_DEFINE_DIAG(18)
QP_DEFINE_DIAG(19)
QP_DEFINE_DIAG(20)
}  // namespace qp
