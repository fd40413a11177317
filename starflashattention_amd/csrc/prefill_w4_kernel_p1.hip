// One flavour of the 4-wave prefill kernel per translation unit, so that the flavours compile in parallel: see the end of
// prefill_w4_kernel.hip (fp16 exact scale).
#define SFA_W4_PART 1
#include "prefill_w4_kernel.hip"
