// Translation unit of the mixed-radix kernel family (kernel_mixed.h), table part 2 (generated mixed_table_2.inc); six
// parts so that a parallel build is not dominated by one file.
#include "kernel_mixed.h"
namespace vkfft_mi355x {
static const MixedVariant kTable[] = {
#include "mixed_table_2.inc"
};
const MixedVariant* mixed_table_2(int* count) { *count = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
} // namespace vkfft_mi355x
