// Register collision kernels, NE = 50 (reference default), scattering + recombination.
#include "qp_collision_fast.inc"

namespace qp {
QP_DEFINE_DIAG_SR(50, 1, 1)
}  // namespace qp
