// Where does an LDS-resident 8192-point FFT block spend its time?  One block per CU, 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define DDSP_CONV_ADD_DRY 1u
#define DDSP_CONV_MASK_TAP0 2u
#define DDSP_OK 0
#define DDSP_ERR_NULL_POINTER -1
#define DDSP_ERR_BAD_SHAPE -2
#define DDSP_ERR_UNSUPPORTED -3
#define DDSP_ERR_WORKSPACE -4
#define DDSP_ERR_LAUNCH -5
namespace ddsp {
constexpr int kRvP = 4096, kRvN = 8192, kRvThreads = 256;
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y)); }
constexpr int kRvBf = kRvN / 4 / kRvThreads;
}
using namespace ddsp;
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float4* __restrict__ dst, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float2 s[];
  const int tid = threadIdx.x;
  int n = 0;
  long long* st = stamps + blockIdx.x * 16;
#define STAMP() do { if (tid == 0) st[n] = wall_clock64(); ++n; } while (0)
  STAMP();
  const float4* sv = src + (size_t)blockIdx.x * 4096;
  for (int i2 = tid; i2 < 4096; i2 += 256) reinterpret_cast<float4*>(s)[i2] = sv[i2];
  __syncthreads();
  STAMP();
#pragma unroll 1
  for (int q = kRvN / 4; q >= 2; q >>= 2) {
    const float inv_len = 0.25f / (float)q;
    float2 v[kRvBf][4];
    int idx[kRvBf];
#pragma unroll
    for (int u = 0; u < kRvBf; ++u) {
      const int t = tid + kRvThreads * u, pos = t & (q - 1);
      idx[u] = ((t - pos) << 2) + pos;
#pragma unroll
      for (int m = 0; m < 4; ++m) v[u][m] = s[idx[u] + m * q];
    }
#pragma unroll
    for (int u = 0; u < kRvBf; ++u) {
      const int pos = (tid + kRvThreads * u) & (q - 1);
      const float rev = (float)pos * inv_len;
      const float2 w1 = make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
      const float2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
      const float2 t0 = cadd(v[u][0], v[u][2]), t1 = csub(v[u][0], v[u][2]), t2 = cadd(v[u][1], v[u][3]), bd = csub(v[u][1], v[u][3]);
      const float2 t3 = make_float2(bd.y, -bd.x);
      v[u][0] = cadd(t0, t2); v[u][1] = cmulc(cadd(t1, t3), w1); v[u][2] = cmulc(csub(t0, t2), w2); v[u][3] = cmulc(csub(t1, t3), w3);
    }
#pragma unroll
    for (int u = 0; u < kRvBf; ++u) {
#pragma unroll
      for (int m = 0; m < 4; ++m) s[idx[u] + m * q] = v[u][m];
    }
    __syncthreads();
    STAMP();
  }
  float4* dv = dst + (size_t)blockIdx.x * 4096;
  for (int i2 = tid; i2 < 4096; i2 += 256) dv[i2] = reinterpret_cast<const float4*>(s)[i2];
  STAMP();
}
int main() {
  const int blocks = 256;
  float4 *src, *dst; long long* st;
  (void)hipMalloc(&src, (size_t)blocks * 65536); (void)hipMemset(src, 0, (size_t)blocks * 65536);
  (void)hipMalloc(&dst, (size_t)blocks * 65536); (void)hipMalloc(&st, blocks * 128);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, src, dst, st);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, src, dst, st);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(blocks * 16);
  (void)hipMemcpy(h.data(), st, blocks * 128, hipMemcpyDeviceToHost);
  printf("kernel %.1f us; block 0 stamps (us since start):", ms * 1e3);
  for (int i = 1; i < 9; ++i) printf(" %.2f", (h[i] - h[0]) * 0.01);
  printf("\nblock 100:");
  for (int i = 1; i < 9; ++i) printf(" %.2f", (h[100 * 16 + i] - h[100 * 16]) * 0.01);
  printf("\n");
  return 0;
}
