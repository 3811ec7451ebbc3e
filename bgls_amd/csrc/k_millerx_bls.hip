// k_miller_x60 for BLS12-381 (k_millerx.inc)
#define BGLS_MILLER_CURVE BLS381
#include "k_millerx.inc"
