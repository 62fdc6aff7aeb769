// real-data plans (R2C/C2R, DCT/DST): added next
#include "engine.h"
namespace vkfft_mi355x {}
