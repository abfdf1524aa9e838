// ALU micro-benchmarks on gfx950: what do the instructions the synth loops are made of cost?
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

template <int MODE>
__global__ __launch_bounds__(256) void alu_kernel(float* out, float seed, const float* __restrict__ tab) {
  float a[UNROLL];
  float th = seed + threadIdx.x * 1e-3f;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) a[u] = th + u;
  float acc0 = 0.f, acc1 = 0.f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) p[u] = v2{th, th + 1.f};
  unsigned ui[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) ui[u] = threadIdx.x * 977u + u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (MODE == 0) a[u] = fmaf(a[u], 1.0001f, 0.5f);                                  // v_fma_f32
      if (MODE == 1) p[u] = __builtin_elementwise_fma(p[u], v2{1.0001f, 1.0002f}, v2{0.5f, 0.25f});  // v_pk_fma_f32
      if (MODE == 2) a[u] = __builtin_amdgcn_sinf(a[u]);                                // v_sin_f32
      if (MODE == 3) {   // the direct synth step: fma + sin + 2 fma (SGPR-like constants)
        const float s = __builtin_amdgcn_sinf(fmaf(th, (float)(u + 1), a[u]));
        acc0 = fmaf(1.25f, s, acc0); acc1 = fmaf(0.75f, s, acc1);
        a[u] += 1e-3f;
      }
      if (MODE == 4) a[u] = __builtin_amdgcn_fractf(a[u] * 1.37f);                      // mul + fract
      if (MODE == 5) ui[u] = __umulhi(ui[u], 0xD2511F53u) ^ (ui[u] * 0xCD9E8D57u);      // mulhi+mullo+xor
      if (MODE == 6) ui[u] = __umul24(ui[u], 0x5F3759u) + 0x9E3779B9u;                   // v_mad_u32_u24
      if (MODE == 7) {   // chebyshev step: s2 = c*s1 - s0 ; 2 acc fma
        const float s2 = fmaf(1.9f, a[u], -acc1 * 1e-9f);
        acc0 = fmaf(1.25f, s2, acc0); acc1 = fmaf(0.75f, s2, acc1); a[u] = s2 * 0.5f;
      }
      if (MODE == 8) a[u] = __builtin_amdgcn_exp2f(a[u] * 1e-3f);                         // v_exp_f32 + mul
    }
  }
  float r = acc0 + acc1;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) r += a[u] + p[u].x + p[u].y + (float)ui[u];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void sin_accuracy_kernel(const float* x, float* y, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_sinf(x[i]);
}

template <int MODE>
double run(float* d_out, const float* tab, int blocks, const char* name, double ops_per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((alu_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, d_out, 0.1f, tab);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((alu_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, d_out, 0.1f, tab);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double inst = (double)blocks * 256 * ITERS * UNROLL * ops_per_iter;   // lane-instructions
  const double wave_inst = inst / 64.0;
  // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
  const double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / wave_inst;
  printf("%-34s %8.3f ms  %8.2f Glane-op/s  ~%.2f cyc/wave-instr/SIMD (at 2.4GHz)\n", name, ms, inst / ms * 1e-6, cyc);
  return ms;
}

int main() {
  int blocks = 256 * 8;    // 8 blocks (32 waves) per CU: full occupancy
  float *d_out, *d_tab; CK(hipMalloc(&d_out, blocks * 256 * 4)); CK(hipMalloc(&d_tab, 4096));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs %d  clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  run<0>(d_out, d_tab, blocks, "v_fma_f32", 1);
  run<1>(d_out, d_tab, blocks, "v_pk_fma_f32 (2 fma/instr)", 1);
  run<2>(d_out, d_tab, blocks, "v_sin_f32", 1);
  run<3>(d_out, d_tab, blocks, "synth step fma+sin+2fma(+add)", 5);
  run<4>(d_out, d_tab, blocks, "v_mul+v_fract", 2);
  run<5>(d_out, d_tab, blocks, "mulhi_u32+mullo_u32+xor", 3);
  run<6>(d_out, d_tab, blocks, "v_mad_u32_u24", 1);
  run<7>(d_out, d_tab, blocks, "cheb step 4fma+mul", 5);
  run<8>(d_out, d_tab, blocks, "v_mul+v_exp_f32", 2);

  // v_sin_f32 accuracy over [0,256) revolutions
  const int n = 1 << 20;
  std::vector<float> hx(n), hy(n);
  for (int i = 0; i < n; ++i) hx[i] = (float)((double)i * 256.0 / n + 1e-4 * (i % 7));
  float *dx, *dy; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4));
  CK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(sin_accuracy_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
  CK(hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst_small = 0;
  for (int i = 0; i < n; ++i) {
    const double ref = sin(2.0 * M_PI * (double)hx[i]);
    const double e = fabs((double)hy[i] - ref);
    if (e > worst) worst = e;
    if (hx[i] < 1.0 && e > worst_small) worst_small = e;
  }
  printf("v_sin_f32 max abs err: x in [0,256): %.3e   x in [0,1): %.3e\n", worst, worst_small);
  return 0;
}
