// Translation unit of the op-FFT kernel family (kernel_opfft.h), table part f32_col_2 (opfft_table_f32_col_2.inc, tools/gen_opfft_col_extra.py): strided C2C of the
// 13-smooth lengths 1025 ... 2048 outside the first generator's tables.
#include "kernel_opfft.h"
namespace vkfft_mi355x {
static const OpfftVariant kTable[] = {
#include "opfft_table_f32_col_2.inc"
};
const OpfftVariant* opfft_table_f32_col_2(int* count) { *count = (int)(sizeof(kTable) / sizeof(kTable[0])); return kTable; }
} // namespace vkfft_mi355x
