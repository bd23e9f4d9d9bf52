// One-pass collision kernel, NE = 50 (the reference's default), scattering + recombination.
#include "qp_collision_onepass.inc"

namespace qp {
QP_DEFINE_ONEPASS(50, 14, 1, 1, 8, 2)
}  // namespace qp
