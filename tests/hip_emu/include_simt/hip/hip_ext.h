// TEST INFRASTRUCTURE: see hip_runtime.h in this directory (hipExtLaunchKernelGGL lives there).
#pragma once
#include <hip/hip_runtime.h>
