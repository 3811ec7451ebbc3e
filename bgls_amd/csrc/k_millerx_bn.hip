// k_miller_x60 for alt-bn128 (k_millerx.inc)
#define BGLS_MILLER_CURVE BN254
#include "k_millerx.inc"
