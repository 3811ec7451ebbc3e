// k_miller_x60 for BLS12-381, 60 pairings per block (k_millerx.inc)
#define BGLS_MILLER_CURVE BLS381
#define BGLS_MILLER_NP 60
#include "k_millerx.inc"
