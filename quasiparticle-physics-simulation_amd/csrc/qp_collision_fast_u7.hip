// Register collision kernels, NE = 45 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(45)
}  // namespace qp
