// fp32-grade field kernel on the fp16 matrix pipe, 16x16x32 mapping, operands as two fp16 pieces (three products per block):
// the body is field_eval_split16_impl.h (see its header for the arithmetic and its error bound).
#define MVS16_F16 1
#include "field_eval_split16_impl.h"
