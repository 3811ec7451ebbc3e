// k_miller_x60 for BLS12-381, 64 pairings per block: one round of 1024 resident blocks = 2^16 pairings (k_millerx.inc)
#define BGLS_MILLER_CURVE BLS381
#define BGLS_MILLER_NP 64
#include "k_millerx.inc"
