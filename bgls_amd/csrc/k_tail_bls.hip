#define BGLS_TAIL_CURVE BLS381
#include "k_tail.inc"
