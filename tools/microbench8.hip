// Phase B of harm_table_kernel (the per-sample table interpolation) in isolation: what does a tile of 64 samples cost
// as a function of the wavefronts per SIMD, the tiles a wavefront carries together, and f0 (LDS bank conflicts)?
// The table rows, the chunk's phase tables and the code (wt_taps, the fp64 phase) are the product's: this file
// includes csrc/harmonic_table.hip and re-states only the loop around them.
//
// One block per CU, 4 W wavefronts, all of them "S" wavefronts; no T role, no barrier, no phase A: an upper bound for
// what phase B can run at.  Reported: shader clocks per tile of SIMD time (= clocks x W / tiles per wavefront).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude tools/microbench8.hip -o tools/bin/microbench8
#include "../ddsp_amd/csrc/harmonic_table.hip"
#include <vector>

namespace ddsp {

constexpr int kMbTiles = 248;         // tiles per wavefront (8 passes over the 31 frames of a chunk)

template <int W, int NT>
__global__ __launch_bounds__(1024) void phase_b_kernel(float* __restrict__ audio, long long* __restrict__ clocks, float f0,
                                                       float sr, int amp_linear) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tab = smem;                                        // [kWtRows][kWtTS]
  ChunkTables& t = *reinterpret_cast<ChunkTables*>(smem + kWtRows * kWtTS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < kWtRows * kWtTS; i += blockDim.x) tab[i] = __sinf(0.01f * (float)i);
  if (tid < kWtRows) {
    const double w = (double)(f0 + 0.37f * (float)tid) / (double)sr;
    t.theta[tid] = 0.123 + 0.0311 * tid;
    t.w[tid] = w;
    t.dw[tid] = 0.37 / (double)sr * (0.5 / 64.0);
    t.kA[tid] = 100; t.kN[tid] = 100;
  }
  if (tid == 0) t.cross = 0;
  __syncthreads();
  const float inv_hop = 1.0f / 64.0f;
  float* out = audio + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * (size_t)(kMbTiles * 64);
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int tile = 0; tile < kMbTiles; tile += NT) {
    int q[kWtNT];
    double cyc[kWtNT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      q[u] = (tile + u + wave) % kWtFrames;
      q[u] = __builtin_amdgcn_readfirstlane(q[u]);
      const double rr = (double)lane;
      cyc[u] = t.theta[q[u]] + (rr + 1.0) * (t.w[q[u]] + t.dw[q[u]] * rr);
    }
    float theta[kWtNT], z[kWtNT], z2[kWtNT];
    bool neg[kWtNT];
    const float* t0p[kWtNT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      theta[u] = (float)__builtin_amdgcn_fract(cyc[u]);
      neg[u] = theta[u] >= 0.5f;
      const float th = neg[u] ? 1.0f - theta[u] : theta[u];
      const float pos = fmaf(th, (float)kWtT, -0.5f);
      const float fl = floorf(pos);
      z[u] = (pos - fl) - 0.5f;
      z2[u] = z[u] * z[u];
      t0p[u] = tab + q[u] * kWtTS + kWtH + (int)fl;
    }
    float acc0[kWtNT] = {0.0f, 0.0f, 0.0f, 0.0f}, acc1[kWtNT] = {0.0f, 0.0f, 0.0f, 0.0f};
    wt_taps<6, 0, NT>(t0p, z, z2, acc0, acc1);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const float lerp = (float)lane * inv_hop;
      const float w_next = amp_linear ? lerp : 0.5f - 0.5f * __builtin_amdgcn_cosf(0.5f * lerp);
      const float w_cur = 1.0f - w_next;
      const float v = w_cur * acc0[u] + w_next * acc1[u];
      out[(size_t)(tile + u) * 64 + lane] = neg[u] ? -v : v;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) clocks[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

}  // namespace ddsp

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W, int NT>
static void run(float* d_audio, long long* d_clk, int n_cu, float f0) {
  using namespace ddsp;
  const size_t lds = 100 * 1024;
  CK(hipFuncSetAttribute((const void*)phase_b_kernel<W, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int threads = 256 * W;
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((phase_b_kernel<W, NT>), dim3(n_cu), dim3(threads), lds, 0, d_audio, d_clk, f0, 16000.0f, 0);
  CK(hipDeviceSynchronize());
  const int nw = n_cu * 4 * W;
  std::vector<long long> h(nw);
  CK(hipMemcpy(h.data(), d_clk, nw * sizeof(long long), hipMemcpyDeviceToHost));
  double sum = 0, mx = 0;
  for (int i = 0; i < nw; ++i) { sum += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
  const double per_tile_wave = sum / nw / kMbTiles;
  printf("  W=%d NT=%d f0=%6.1f: %7.1f clocks per tile as a wavefront sees it, %6.1f of SIMD time (slowest wavefront %7.1f)\n", W, NT,
         f0, per_tile_wave, per_tile_wave / W, mx / kMbTiles);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float* d_audio;
  long long* d_clk;
  CK(hipMalloc(&d_audio, (size_t)n_cu * 16 * ddsp::kMbTiles * 64 * sizeof(float)));
  CK(hipMalloc(&d_clk, (size_t)n_cu * 16 * sizeof(long long)));
  printf("# phase B of harm_table_kernel alone (W = 6 taps, hop 64): clocks per 64-sample tile\n");
  const float f0s[6] = {70.f, 125.f, 200.f, 250.f, 500.f, 1000.f};
  for (int fi = 0; fi < 6; ++fi) {
    const float f0 = f0s[fi];
    if (fi == 0 || fi == 2) {
      run<1, 1>(d_audio, d_clk, n_cu, f0); run<1, 2>(d_audio, d_clk, n_cu, f0); run<1, 4>(d_audio, d_clk, n_cu, f0);
      run<2, 1>(d_audio, d_clk, n_cu, f0); run<2, 2>(d_audio, d_clk, n_cu, f0); run<2, 4>(d_audio, d_clk, n_cu, f0);
      run<3, 1>(d_audio, d_clk, n_cu, f0); run<3, 2>(d_audio, d_clk, n_cu, f0); run<3, 4>(d_audio, d_clk, n_cu, f0);
      run<4, 1>(d_audio, d_clk, n_cu, f0); run<4, 2>(d_audio, d_clk, n_cu, f0); run<4, 4>(d_audio, d_clk, n_cu, f0);
    } else {
      run<2, 4>(d_audio, d_clk, n_cu, f0); run<4, 2>(d_audio, d_clk, n_cu, f0);
    }
  }
  return 0;
}
