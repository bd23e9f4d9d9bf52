// Register collision kernels, NE = 21, 22, 23, 24 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(21)
QP_DEFINE_DIAG(22)
QP_DEFINE_DIAG(23)
QP_DEFINE_DIAG(24)
}  // namespace qp
