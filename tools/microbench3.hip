// How fast do 2016 blocks x 256 threads each get 6.8 KB of contiguous fp32 from HBM at kernel start?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int MODE>  // 0: 2 float4 loads per thread; 1: 7 dword loads per thread; 2: float4 + extra 4KB L2-resident read
__global__ __launch_bounds__(256) void read_kernel(const float* __restrict__ src, const float* __restrict__ small, float* out, long long* stamps) {
  const int tid = threadIdx.x;
  long long t0 = wall_clock64();
  const float* base = src + (size_t)blockIdx.x * 1600;
  float acc = 0.f;
  if (MODE == 0 || MODE == 2) {
    const float4* b4 = reinterpret_cast<const float4*>(base);
    float4 a = b4[tid];
    float4 c = (tid + 256 < 425) ? b4[tid + 256] : make_float4(0, 0, 0, 0);
    if (MODE == 2) { const float4 e = reinterpret_cast<const float4*>(small)[tid]; acc += e.x + e.w; }
    acc += a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w;
  } else {
    float v[7];
#pragma unroll
    for (int n = 0; n < 7; ++n) v[n] = (tid + 256 * n < 1700) ? base[tid + 256 * n] : 0.f;
#pragma unroll
    for (int n = 0; n < 7; ++n) acc += v[n];
  }
  out[blockIdx.x * 256 + tid] = acc;
  __syncthreads();
  if (tid == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = wall_clock64(); }
}
template <int MODE> void run(const char* name, const float* src, const float* small, float* out, long long* d_st, int blocks) {
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((read_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, src, small, out, d_st);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(blocks * 2);
  (void)hipMemcpy(h.data(), d_st, blocks * 16, hipMemcpyDeviceToHost);
  long long t0 = h[0]; for (int i = 0; i < blocks; ++i) t0 = std::min(t0, h[2 * i]);
  std::vector<double> arr(blocks), st(blocks);
  for (int i = 0; i < blocks; ++i) { st[i] = (h[2 * i] - t0) * 0.01; arr[i] = (h[2 * i + 1] - t0) * 0.01; }
  std::vector<double> s = arr; std::sort(s.begin(), s.end());
  std::vector<double> s2 = st; std::sort(s2.begin(), s2.end());
  printf("%-44s blocks=%d  start max %.1f us | data arrived: min %.1f  median %.1f  90%% %.1f  max %.1f us\n", name, blocks, s2.back(), s[0], s[blocks / 2], s[blocks * 9 / 10], s.back());
}
int main() {
  float *src, *small, *out; long long* st;
  (void)hipMalloc(&src, (size_t)8192 * 1600 * 4 + 65536); (void)hipMemset(src, 0, (size_t)8192 * 1600 * 4 + 65536);
  (void)hipMalloc(&small, 65536); (void)hipMemset(small, 0, 65536);
  (void)hipMalloc(&out, (size_t)8192 * 256 * 4); (void)hipMalloc(&st, 8192 * 16);
  for (int blocks : {2016, 8064}) {
    run<0>("2 x float4 per thread", src, small, out, st, blocks);
    run<1>("7 x dword per thread", src, small, out, st, blocks);
    run<2>("2 x float4 + 4 KB L2-resident float4", src, small, out, st, blocks);
  }
  return 0;
}
