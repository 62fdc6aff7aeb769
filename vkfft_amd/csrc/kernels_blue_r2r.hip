// Translation unit of the Bluestein-wrapped real-transform kernels (kernel_blue_r2r.h): instances, registry, launcher.
#include "kernel_blue_r2r.h"
