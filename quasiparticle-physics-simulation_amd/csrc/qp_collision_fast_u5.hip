// Register collision kernels, NE = 40 (see qp_collision_fast.inc).
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG(40)
QP_DEFINE_DIAGP(40)
}  // namespace qp
