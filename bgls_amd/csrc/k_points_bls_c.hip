// k_points.inc for BLS12-381, part 3: the trees above the sums' main passes
#define BGLS_UNIT_CURVE BLS381
#define BGLS_UNIT_IS_BN 0
#define BGLS_UNIT_PART 3
#include "k_points.inc"
