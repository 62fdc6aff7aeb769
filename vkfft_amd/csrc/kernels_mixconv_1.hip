// Translation unit of the one-kernel cyclic convolution family (kernel_mixconv.h), table part 1 (generated mixconv_table_1.inc).
#include "kernel_mixconv.h"
namespace vkfft_mi355x {
static const MixConvVariant kTable[] = {
#include "mixconv_table_1.inc"
};
const MixConvVariant* mixconv_table_1(int* count) { *count = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
} // namespace vkfft_mi355x
