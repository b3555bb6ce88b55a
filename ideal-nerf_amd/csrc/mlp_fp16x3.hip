// The x3 kernel with fp16 instead of bf16 halves (IDN_PREC_FP16X3): same source, same stream geometry,
// same MFMA count; 11+11 significand bits per operand instead of 8+8.  See mlp_bf16x3.hip.
#define IDN_X3_FP16 1
#include "mlp_bf16x3.hip"
