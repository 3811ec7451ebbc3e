// k_points.inc for BLS12-381, part 1: the sums
#define BGLS_UNIT_CURVE BLS381
#define BGLS_UNIT_IS_BN 0
#define BGLS_UNIT_PART 1
#include "k_points.inc"
