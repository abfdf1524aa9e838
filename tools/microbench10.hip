// Does a wavefront that streams MFMAs take issue time away from the other wavefronts of its SIMD?
// One block of 16 wavefronts (4 per SIMD): wavefronts 0-3 (one per SIMD) run role R0, wavefronts 4-15 role R1, each a loop
// of one instruction kind; every wavefront reports clocks per instruction.  Rows: R0 x R1 combinations, with "idle" = the
// wavefronts leave at once.  (harm_table_kernel's tabulators against its interpolators / row makers; the noise kernel's FIR
// wavefronts against its designers.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench10.hip -o tools/bin/microbench10
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2048;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
enum Kind { IDLE, MFMA, MFMA_DEP, MFMA_D2, MFMA_D3, MFMA_4x4, MFMA_4x4_DEP, FMA, PKFMA, DSR2, EXP, FMA64, SALU, N_KIND };
static const char* kNames[N_KIND] = {"idle", "v_mfma_f32_16x16x32_f16 (4 independent)", "v_mfma (one dependent chain)", "v_mfma (two chains, alternating)", "v_mfma (three chains in turn)", "v_mfma_f32_4x4x4_16B_f16 (4 independent)", "v_mfma_f32_4x4x4 (one chain)", "v_fma_f32", "v_pk_fma_f32",
                                     "ds_read2_b32", "v_exp_f32", "v_fma_f64", "s_add_u32"};

template <int K>
__device__ __forceinline__ float run(float seed, int lane, const float* lds, long long& clk) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + (float)(lane + i) * 1e-3f;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f16x8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)(seed + e); fb[e] = (_Float16)(seed - e); }
  f32x2 p[4] = {{a[0], a[1]}, {a[2], a[3]}, {a[4], a[5]}, {a[6], a[7]}};
  double d[4] = {a[0], a[1], a[2], a[3]};
  const unsigned addr = (unsigned)(lane * 8);
  const long long t0 = clock64();
  if constexpr (K == IDLE) { clk = 0; return 0.f; }
  for (int it = 0; it < ITERS / 8; ++it) {
    if constexpr (K == MFMA) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[r & 3], 0, 0, 0);
    } else if constexpr (K == MFMA_DEP) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[0], 0, 0, 0);
    } else if constexpr (K == MFMA_D2) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[r & 1], 0, 0, 0);
    } else if constexpr (K == MFMA_D3) {
#pragma unroll
      for (int r = 0; r < 9; ++r) acc[r % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[r % 3], 0, 0, 0);
    } else if constexpr (K == FMA) {
#pragma unroll
      for (int r = 0; r < 8; ++r) __asm__ volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[r]) : "v"(seed));
    } else if constexpr (K == PKFMA) {
#pragma unroll
      for (int r = 0; r < 8; ++r) __asm__ volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[r & 3]) : "v"(p[(r + 1) & 3]));
    } else if constexpr (K == DSR2) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        f32x2 v;
        __asm__ volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(addr + 512u * (unsigned)r));
        if (r == 7) { __asm__ volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)); a[0] += v[0]; }
      }
    } else if constexpr (K == EXP) {
#pragma unroll
      for (int r = 0; r < 8; ++r) __asm__ volatile("v_exp_f32 %0, %0" : "+v"(a[r]));
    } else if constexpr (K == FMA64) {
#pragma unroll
      for (int r = 0; r < 8; ++r) __asm__ volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[r & 3]) : "v"(d[(r + 1) & 3]));
    }
  }
  clk = clock64() - t0;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += a[i];
  sum += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + p[0][0] + p[1][1] + p[2][0] + p[3][1] + (float)(d[0] + d[1] + d[2] + d[3]) + lds[lane];
  return sum;
}

template <int K0, int K1>
__global__ __launch_bounds__(1024) void k(long long* __restrict__ clocks, float* __restrict__ sink, float seed, unsigned* __restrict__ ids) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 8192; i += 1024) lds[i] = (float)i;
  __syncthreads();
  long long c = 0;
  float v;
  if (wave < 4) v = run<K0>(seed, lane, lds, c);
  else v = run<K1>(seed, lane, lds, c);
  unsigned hwid;
  __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (lane == 0) { clocks[blockIdx.x * 16 + wave] = c; ids[blockIdx.x * 16 + wave] = hwid; }
  if (v == 12345.678f) sink[tid] = v;
}

static unsigned* d_ids;
template <int K0, int K1>
void go(long long* d_clk, float* d_sink, int blocks) {
  std::vector<long long> h(16 * blocks);
  std::vector<unsigned> ids(16 * blocks);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<K0, K1>), dim3(blocks), dim3(1024), 0, 0, d_clk, d_sink, 1.0f, d_ids);
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(h.data(), d_clk, sizeof(long long) * 16 * blocks, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ids.data(), d_ids, sizeof(unsigned) * 16 * blocks, hipMemcpyDeviceToHost));
  printf("R0 = %-42s R1 = %-14s :", kNames[K0], kNames[K1]);
  for (int b = 0; b < blocks; ++b) {
    double r0 = 0, r1 = 0;
    for (int w = 0; w < 4; ++w) r0 += (double)h[b * 16 + w] / 4;
    for (int w = 4; w < 16; ++w) r1 += (double)h[b * 16 + w] / 12;
    printf("  [%5.1f %5.1f simd", r0 / ITERS, r1 / ITERS);
    for (int w = 0; w < 16; ++w) printf("%s%u", w % 4 == 0 ? " " : "", (ids[b * 16 + w] >> 4) & 3);
    printf("]");
  }
  printf("\n");
}

int main() {
  long long* d_clk; float* d_sink;
  CK(hipMalloc(&d_clk, sizeof(long long) * 16 * 256));
  CK(hipMalloc(&d_sink, sizeof(float) * 1024));
  CK(hipMalloc(&d_ids, sizeof(unsigned) * 16 * 256));
  const int B = 3;
  for (int pass = 0; pass < 2; ++pass) {
    go<MFMA, IDLE>(d_clk, d_sink, B); go<IDLE, FMA>(d_clk, d_sink, B);
    go<MFMA, FMA>(d_clk, d_sink, B); go<MFMA_DEP, FMA>(d_clk, d_sink, B); go<MFMA_D2, FMA>(d_clk, d_sink, B); go<MFMA_D3, FMA>(d_clk, d_sink, B);
    go<FMA, FMA>(d_clk, d_sink, B);
    go<IDLE, PKFMA>(d_clk, d_sink, B); go<MFMA, PKFMA>(d_clk, d_sink, B); go<MFMA_DEP, PKFMA>(d_clk, d_sink, B); go<PKFMA, PKFMA>(d_clk, d_sink, B);
    go<IDLE, DSR2>(d_clk, d_sink, B); go<MFMA, DSR2>(d_clk, d_sink, B);
    go<IDLE, EXP>(d_clk, d_sink, B); go<MFMA, EXP>(d_clk, d_sink, B);
    go<IDLE, FMA64>(d_clk, d_sink, B); go<MFMA, FMA64>(d_clk, d_sink, B);
    go<DSR2, FMA>(d_clk, d_sink, B); go<EXP, FMA>(d_clk, d_sink, B); go<EXP, PKFMA>(d_clk, d_sink, B);
  }
  return 0;
}
