// Translation unit of the mixed-radix kernel family (kernel_mixed.h), table part 17: fp64 rows of 4097 ... 8192 points with a factor 11 or 13 in one LDS buffer
// (mixed_table_17.inc, tools/gen_long_rows_table.py).
#include "kernel_mixed.h"
namespace vkfft_mi355x {
static const MixedVariant kTable[] = {
#include "mixed_table_17.inc"
};
const MixedVariant* mixed_table_17(int* count) { *count = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
} // namespace vkfft_mi355x
