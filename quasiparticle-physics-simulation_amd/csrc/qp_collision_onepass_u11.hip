// One-pass collision kernel, NE = 50, gap-class form (K per lane from the amplitude tables): scattering only, phonons frozen.
// Block / group sizes from a same-box A/B of nine variants (tools/exp_gap_variants.sh).
#include "qp_collision_onepass.inc"

namespace qp {
QP_DEFINE_ONEPASS_CLASSES(50, 13, 1, 0, 0, 6, 2, 4, 3, 8)
}  // namespace qp
