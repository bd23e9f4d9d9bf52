// Register collision kernels, NE = 50, gap-class variants of the single-process combinations.
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAGP_SR(50, 0, 1)
QP_DEFINE_DIAGP_SR(50, 1, 0)
}  // namespace qp
