#define BGLS_MILLER_CURVE BN254
#define BGLS_MILLER_IS_BN 1
#include "k_miller.inc"
