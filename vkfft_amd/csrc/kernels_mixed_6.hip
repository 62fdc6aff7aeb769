// Translation unit of the mixed-radix kernel family (kernel_mixed.h), table part 6: the hand-written long rows (mixed_table_6.inc).
#include "kernel_mixed.h"
namespace vkfft_mi355x {
static const MixedVariant kTable[] = {
#include "mixed_table_6.inc"
};
const MixedVariant* mixed_table_6(int* count) { *count = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
} // namespace vkfft_mi355x
