// k_msm.inc for BLS12-381
#define BGLS_UNIT_CURVE BLS381
#define BGLS_UNIT_IS_BN 0
#include "k_msm.inc"
