// k_points.inc for BLS12-381, part 2: scalar multiplications, checks, parsing
#define BGLS_UNIT_CURVE BLS381
#define BGLS_UNIT_IS_BN 0
#define BGLS_UNIT_PART 2
#include "k_points.inc"
