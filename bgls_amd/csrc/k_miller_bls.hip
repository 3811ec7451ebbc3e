#define BGLS_MILLER_CURVE BLS381
#define BGLS_MILLER_IS_BN 0
#include "k_miller.inc"
