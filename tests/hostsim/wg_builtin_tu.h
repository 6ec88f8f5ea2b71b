// translation unit of the library-built workgroup kernels for the host build (tests/hostsim/wg_harness.cpp)
#include "cdkf_reg_kernels.h"
#include "cdkf_wg2_kernels.h"
