// One flavour of the 4-wave prefill kernel per translation unit, so that the flavours compile in parallel: see the end of
// prefill_w4_kernel.hip (fp16 prescaled Q).
#define SFA_W4_PART 2
#include "prefill_w4_kernel.hip"
