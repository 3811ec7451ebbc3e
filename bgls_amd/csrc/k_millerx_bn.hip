// k_miller_x60 for alt-bn128, 60 pairings per block (k_millerx.inc)
#define BGLS_MILLER_CURVE BN254
#define BGLS_MILLER_NP 60
#include "k_millerx.inc"
