// TEST INFRASTRUCTURE ONLY - never shipped, never loaded by ddsp_amd.
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile ddsp_amd/csrc/general.hip (the kernels
// written in plain HIP: one thread per output, no LDS, no cross-lane traffic, no inline asm) into a
// HOST shared object, tests/hip_emu/_build/libddsp_general_emu.so.  A launch becomes a serial loop
// over (block, thread) with threadIdx / blockIdx set for each iteration.  tests/test_general_emulated.py
// calls the same extern "C" entry points on numpy buffers and compares with the oracle, so that index
// arithmetic, argument checks and the host-side launch geometry of those kernels are exercised in the
// CPU test run (`-m "not gpu"`), where no GPU exists.  It says nothing about performance and it is not
// a CPU fallback: the product library is libddsp_amd.so built by hipcc for gfx950 and nothing else.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

using std::max;
using std::min;

template <class Kernel, class... Args>
void ddsp_emu_launch(Kernel kernel, dim3 grid, dim3 block, Args... args) {
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx) {
              threadIdx = dim3(tx, ty, tz);
              kernel(args...);
            }
      }
}
#define hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, ...) \
  ddsp_emu_launch(kernel, grid, block, __VA_ARGS__)

// device intrinsics the kernels use (built with -ffp-contract=off: a*b+c stays two roundings)
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __builtin_amdgcn_sinf(float rev) { return (float)std::sin(6.283185307179586476925 * (double)rev); }
inline float __builtin_amdgcn_cosf(float rev) { return (float)std::cos(6.283185307179586476925 * (double)rev); }
inline float cospif(float x) { return (float)std::cos(3.14159265358979323846 * (double)x); }
inline float sinpif(float x) { return (float)std::sin(3.14159265358979323846 * (double)x); }
