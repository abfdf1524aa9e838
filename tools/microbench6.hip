// LDS fragment reads for a Toeplitz MFMA operand: is a 16-byte ds_read at 2-byte alignment (a) correct and
// (b) how much slower than an aligned one?  Pattern: lane (i = lane & 15, g = lane >> 4) reads the 8 fp16 values
// starting at element  base + 16 * it + 8 * g + i  (the A-fragment of a 16x16x32 MFMA whose rows are a signal
// shifted by one sample per row), against the aligned pattern base + 16 * it + 8 * g (all rows alike).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct __attribute__((packed, aligned(2))) H8 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) H8a4 { uint32_t x, y, z, w; };
template <int MODE>   // 0: aligned b128; 1: 2-byte aligned b128; 2: 4-byte aligned (lane shift doubled) via read2_b32 pairs
__global__ __launch_bounds__(256) void k(uint32_t* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[16384 + 64];
  for (int i = threadIdx.x; i < 16384 + 64; i += 256) sm[i] = (uint16_t)(i * 7 + 3);
  __syncthreads();
  const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  uint32_t acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int base = ((it * 16 + u) * 32 + wv * 512) & 8191;
      if (MODE == 0) {
        const uint4 v = *reinterpret_cast<const uint4*>(sm + base + 8 * g + 8 * (i16 & 1));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      } else if (MODE == 1) {
        const H8 v = *reinterpret_cast<const H8*>(sm + base + 8 * g + i16);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      } else {
        const H8a4 v = *reinterpret_cast<const H8a4*>(sm + base + 8 * g + 2 * (i16 >> 1));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void check(uint32_t* out) {   // every 2-byte offset 0..15 of a 16-byte read
  __shared__ __attribute__((aligned(16))) uint16_t sm[256];
  for (int i = threadIdx.x; i < 256; i += 64) sm[i] = (uint16_t)(i * 7 + 3);
  __syncthreads();
  const H8 v = *reinterpret_cast<const H8*>(sm + threadIdx.x);
  out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
template <int MODE> void run(const char* name, uint32_t* out, long long* cyc) {
  const int blocks = 1024, iters = 200;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(blocks);
  (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
  // per CU: blocks/256 rounds (occupancy permitting) x 4 waves x iters x 16 reads
  printf("%-46s %.3f ms   block clock64 ticks per 16-byte wave-read: %.1f   (4 waves per block, ~%d blocks per CU)\n",
         name, ms, mean / (iters * 16.0), blocks / 256);
}
int main() {
  uint32_t* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, out);
  std::vector<uint32_t> h(256);
  (void)hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t)
    for (int j = 0; j < 4; ++j) {
      const uint32_t lo = (uint16_t)((t + 2 * j) * 7 + 3), hi = (uint16_t)((t + 2 * j + 1) * 7 + 3);
      if (h[t * 4 + j] != (lo | (hi << 16))) ++bad;
    }
  printf("unaligned (2-byte) 16-byte LDS reads: %s (%d wrong dwords of 256)\n", bad ? "WRONG" : "correct", bad);
  run<0>("aligned ds_read_b128", out, cyc);
  run<1>("2-byte-aligned 16-byte read (Toeplitz rows)", out, cyc);
  run<2>("4-byte-aligned 16-byte read", out, cyc);
  return 0;
}
