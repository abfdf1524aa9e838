// How should the FIR wavefronts of noise_mfma65_kernel fetch a Toeplitz fragment - 16 bytes per lane from an address that moves
// by one fp16 element from row to row, kept 4-byte aligned by the two shifted copies (csrc/filtered_noise_mfma.hip)?  Today: two
// ds_read2_b32 (8 LDS cycles per fragment by MI355X_MICROARCH.md's table).  Candidates: ONE ds_read_b128 at a 4-byte aligned
// address, TWO ds_read_b64 at 4-byte aligned addresses (4 cycles if the LDS takes them at the aligned rate).  Reports clocks per
// fragment and wavefront for 2 blocks of 8 wavefronts per CU (the kernel's geometry), 4 reading wavefronts per block, and checks
// that the bytes that arrive are the right ones.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench11.hip -o tools/bin/microbench11
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 4096;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
enum Kind { B128_ALIGNED, READ2_B32_X2, B128_AT4, B64_X2_AT4, B64_X2_ALIGNED, READ2_B64_ALIGNED, N_KIND };
static const char* kNames[N_KIND] = {"ds_read_b128, 16-byte aligned (lane * 16)", "2 x ds_read2_b32, Toeplitz at 4 bytes (today)",
                                     "1 x ds_read_b128 at a 4-byte aligned address", "2 x ds_read_b64 at 4-byte aligned addresses",
                                     "2 x ds_read_b64, 8-byte aligned (Toeplitz at 8 bytes)",
                                     "1 x ds_read2_b64 offset1:1, 8-byte aligned (Toeplitz at 8 bytes)"};

template <int K>
__global__ __launch_bounds__(512) void bench(long long* clocks, unsigned* wrong) {
  __shared__ __attribute__((aligned(16))) unsigned lds[8192];          // 32 KB; dword i holds i
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (unsigned)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) return;                                              // four reading wavefronts per block (the FIR wavefronts)
  const int i = lane & 15, g = lane >> 4;
  // the kernel's pattern: row i reads a 16-byte fragment one dword further down per row; g picks one of four 32-byte groups
  unsigned addr;
  if (K == B128_ALIGNED) addr = 16u * (unsigned)lane + 4096u * (unsigned)wave;
  else if (K == B64_X2_ALIGNED || K == READ2_B64_ALIGNED) addr = 8u * (unsigned)(15 - i) + 160u * (unsigned)g + 4096u * (unsigned)wave;
  else addr = 4u * (unsigned)(15 - i) + 160u * (unsigned)g + 4096u * (unsigned)wave;
  unsigned bad = 0, sum = 0;
  const long long t0 = clock64();
  for (int it = 0; it < ITERS / 8; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned a = addr + 320u * (unsigned)(r & 3);
      if (K == B128_ALIGNED || K == B128_AT4) {
        __asm__ volatile("ds_read_b128 %0, %1" : "=v"(v[r]) : "v"(a));
      } else if (K == READ2_B32_X2) {
        u32x2 lo, hi;
        __asm__ volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(lo) : "v"(a));
        __asm__ volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(hi) : "v"(a));
        v[r] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
      } else if (K == READ2_B64_ALIGNED) {
        __asm__ volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v[r]) : "v"(a));
      } else {
        u32x2 lo, hi;
        __asm__ volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(a));
        __asm__ volatile("ds_read_b64 %0, %1 offset:8" : "=v"(hi) : "v"(a));
        v[r] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
      }
    }
    __asm__ volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned a = (addr + 320u * (unsigned)(r & 3)) >> 2;
      bad += (v[r][0] != a) + (v[r][1] != a + 1) + (v[r][2] != a + 2) + (v[r][3] != a + 3);
      sum += v[r][0];
    }
  }
  const long long t1 = clock64();
  if (lane == 0) clocks[blockIdx.x * 4 + wave] = t1 - t0;
  atomicAdd(wrong, bad + (sum == 0xdeadbeefu));
}

template <int K>
void run(long long* d_clk, unsigned* d_wrong, int blocks) {
  CK(hipMemset(d_wrong, 0, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(512), 0, 0, d_clk, d_wrong);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(512), 0, 0, d_clk, d_wrong);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> clk(blocks * 4);
  unsigned wrong = 0;
  CK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&wrong, d_wrong, 4, hipMemcpyDeviceToHost));
  double mean = 0;
  for (long long c : clk) mean += (double)c;
  mean /= (double)clk.size();
  printf("%-58s  %7.1f clocks per fragment and wavefront   kernel %.3f ms   wrong dwords (two launches): %u\n", kNames[K],
         mean / ITERS, ms, wrong);
}

int main() {
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = 2 * cus;                     // two blocks of 8 wavefronts per CU, four of them reading
  long long* d_clk; unsigned* d_wrong;
  CK(hipMalloc(&d_clk, blocks * 4 * 8)); CK(hipMalloc(&d_wrong, 4));
  printf("%d CUs, %d blocks of 8 wavefronts (4 reading), %d fragments of 16 bytes per lane and wavefront\n", cus, blocks, ITERS);
  run<B128_ALIGNED>(d_clk, d_wrong, blocks);
  run<READ2_B32_X2>(d_clk, d_wrong, blocks);
  run<B128_AT4>(d_clk, d_wrong, blocks);
  run<B64_X2_AT4>(d_clk, d_wrong, blocks);
  run<B64_X2_ALIGNED>(d_clk, d_wrong, blocks);
  run<READ2_B64_ALIGNED>(d_clk, d_wrong, blocks);
  return 0;
}
