// Translation unit of the op-FFT kernel family (kernel_opfft.h), table part f64_row_1 (generated opfft_table_f64_row_1.inc).
#include "kernel_opfft.h"
namespace vkfft_mi355x {
static const OpfftVariant kTable[] = {
#include "opfft_table_f64_row_1.inc"
};
const OpfftVariant* opfft_table_f64_row_1(int* count) { *count = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
} // namespace vkfft_mi355x
