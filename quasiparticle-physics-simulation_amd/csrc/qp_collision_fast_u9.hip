// Register collision kernels, NE = 50 (reference default), single-process variants.
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG_SR(50, 0, 1)
QP_DEFINE_DIAG_SR(50, 1, 0)
}  // namespace qp
