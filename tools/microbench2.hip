// Micro-benchmark of the harmonic inner loop (Chebyshev block of 16 with SGPR amplitude operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float sgpr16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float frac_phase(float theta, float kf) { return fmaf(theta, kf, -rintf(theta * kf)); }

template <int MODE>   // 0: s_load per block (hot line), 1: no loads (amplitudes loaded once), 2: s_load cold (strided rows)
                      // 3: like 1 but amplitudes in VGPRs, 4: like 1 but no sin seeds (pure FMA stream)
__global__ __launch_bounds__(256) void cheb_loop(const float* __restrict__ amp, float* out, int nblocks_per_tile, int ntiles, int row_stride) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * 4 + (threadIdx.x >> 6));
  float total = 0.f;
  sgpr16 a0, a1;
  const float* base = amp + (size_t)__builtin_amdgcn_readfirstlane(wave_global % 1024) * row_stride;
  float va0[16], va1[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) { va0[u] = 0.001f * u + lane * 1e-6f; va1[u] = 0.002f * u + lane * 1e-6f; }
  long long c_start = 0, w_start = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) { c_start = clock64(); w_start = wall_clock64(); }
  if (MODE == 1 || MODE == 4) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a0), "=&s"(a1) : "s"(base), "s"(base + 16) : "memory");
  }
  for (int t = 0; t < ntiles; ++t) {
    const float theta = 0.001f * lane + 0.013f * t;
    const float c2 = 2.0f * __builtin_amdgcn_cosf(theta);
    float acc0 = 0.f, acc1 = 0.f;
    for (int kb = 0; kb < nblocks_per_tile; ++kb) {
      const int k0 = kb * 16;
      if (MODE == 0 || MODE == 2) {
        const float* p0 = base + (MODE == 2 ? (t * nblocks_per_tile + kb) * 32 : 0);
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a0), "=&s"(a1) : "s"(p0), "s"(p0 + 16) : "memory");
      }
      float s0, s1;
      if (MODE == 4) { s0 = theta + k0; s1 = theta * 0.5f + k0; }
      else { s0 = __builtin_amdgcn_sinf(frac_phase(theta, (float)(k0 + 1))); s1 = __builtin_amdgcn_sinf(frac_phase(theta, (float)(k0 + 2))); }
      if (MODE == 3) {
        acc0 = fmaf(va0[0], s0, acc0); acc1 = fmaf(va1[0], s0, acc1);
        acc0 = fmaf(va0[1], s1, acc0); acc1 = fmaf(va1[1], s1, acc1);
#pragma unroll
        for (int u = 2; u < 16; ++u) {
          const float s2 = fmaf(c2, s1, -s0);
          acc0 = fmaf(va0[u], s2, acc0); acc1 = fmaf(va1[u], s2, acc1);
          s0 = s1; s1 = s2;
        }
      } else {
        acc0 = fmaf(a0[0], s0, acc0); acc1 = fmaf(a1[0], s0, acc1);
        acc0 = fmaf(a0[1], s1, acc0); acc1 = fmaf(a1[1], s1, acc1);
#pragma unroll
        for (int u = 2; u < 16; ++u) {
          const float s2 = fmaf(c2, s1, -s0);
          acc0 = fmaf(a0[u], s2, acc0); acc1 = fmaf(a1[u], s2, acc1);
          s0 = s1; s1 = s2;
        }
      }
    }
    total += acc0 * 0.5f + acc1;
  }
  out[blockIdx.x * 256 + threadIdx.x] = total;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long dc = clock64() - c_start, dw = wall_clock64() - w_start;
    out[4096 * 256 - 1] = (float)((double)dc / (double)dw * 100.0);   // shader MHz (wall clock = 100 MHz)
  }
}

template <int MODE>
void run(const char* name, const float* amp, float* out, int blocks, int row_stride) {
  const int nb = 7, nt = 64;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((cheb_loop<MODE>), dim3(blocks), dim3(256), 0, 0, amp, out, nb, nt, row_stride);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((cheb_loop<MODE>), dim3(blocks), dim3(256), 0, 0, amp, out, nb, nt, row_stride);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double wave_blocks = (double)blocks * 4 * nb * nt;          // 16-harmonic blocks executed by waves
  const double us_per_block_simd = ms * 1e3 * 1024.0 / wave_blocks;  // SIMD-time per block
  float mhz = 0; (void)hipMemcpy(&mhz, out + 4096 * 256 - 1, 4, hipMemcpyDeviceToHost);
  printf("[%.0f MHz] ", mhz);
  printf("%-36s blocks=%5d  %8.3f ms   %.4f us SIMD-time per 16-harmonic block (= %.0f cyc @2.0GHz; ideal ~61 units = 122 cyc)\n",
         name, blocks, ms, us_per_block_simd, us_per_block_simd * 2000.0);
}

int main() {
  float *amp, *out;
  const int row_stride = 64 * 7 * 32 + 64;   // floats per wave "row set" for cold mode
  (void)hipMalloc(&amp, (size_t)1024 * row_stride * 4);
  (void)hipMemset(amp, 0, (size_t)1024 * row_stride * 4);
  (void)hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {256, 1024, 2048}) {
    run<1>("no loads (SGPR resident)", amp, out, blocks, row_stride);
    run<0>("s_load hot line per block", amp, out, blocks, row_stride);
    run<2>("s_load cold lines per block", amp, out, blocks, row_stride);
    run<3>("no loads, amplitudes in VGPRs", amp, out, blocks, row_stride);
    run<4>("no loads, no sin seeds", amp, out, blocks, row_stride);
  }
  return 0;
}
