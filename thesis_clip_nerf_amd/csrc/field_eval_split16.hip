// fp32-grade field kernel on the bf16 matrix pipe, 16x16x32 mapping, operands cut exactly into three bf16 pieces (six products per block):
// the body is field_eval_split16_impl.h.
#define MVS16_F16 0
#include "field_eval_split16_impl.h"
