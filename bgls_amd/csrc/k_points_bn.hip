// k_points.inc for alt-bn128 (+ what is not a template of the curve)
#define BGLS_UNIT_CURVE BN254
#define BGLS_UNIT_IS_BN 1
#include "k_points.inc"
