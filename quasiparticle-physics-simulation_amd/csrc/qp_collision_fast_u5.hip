// Register collision kernels, NE = 32, 36 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(32)
QP_DEFINE_DIAG(36)
}  // namespace qp
