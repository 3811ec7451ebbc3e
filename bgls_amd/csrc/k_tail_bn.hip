#define BGLS_TAIL_CURVE BN254
#include "k_tail.inc"
