// Translation unit of the one-kernel cyclic convolution family (kernel_mixconv.h), table part 4 (generated mixconv_table_4.inc).
#include "kernel_mixconv.h"
namespace vkfft_mi355x {
static const MixConvVariant kTable[] = {
#include "mixconv_table_4.inc"
};
const MixConvVariant* mixconv_table_4(int* count) { *count = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
} // namespace vkfft_mi355x
