// Register collision kernels, NE = 50 (reference default), gap-class (non-uniform gap) variants.
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAGP_SR(50, 1, 1)
}  // namespace qp
