// Register collision kernels, NE = 25, 28, 30 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(25)
QP_DEFINE_DIAG(28)
QP_DEFINE_DIAG(30)
}  // namespace qp
