// Translation unit of the mixed-radix kernel family (kernel_mixed.h + generated mixed_table.inc): kept apart from
// kernels.hip so that the two compile in parallel.
#include "kernel_mixed.h"
