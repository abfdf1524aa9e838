// Phase B of harm_table_kernel (the per-sample table interpolation) in isolation: what does a tile of 64 samples cost
// as a function of the wavefronts per SIMD, the tiles a wavefront carries together, and f0 (LDS bank conflicts)?
// The table rows, the chunk's phase tables and the code (wt_taps, the fp64 phase) are the product's: this file
// includes csrc/harmonic_table.hip and re-states only the loop around them.
//
// One block per CU, 4 W wavefronts, all of them "S" wavefronts; no T role, no barrier, no phase A: an upper bound for
// what phase B can run at.  Reported: shader clocks per tile of SIMD time (= clocks x W / tiles per wavefront).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude tools/microbench8.hip -o tools/bin/microbench8
// (AS OF the first two sessions of round 3 - commits 36ebdef and 2406a61: the round-2 kernel whose
// scalar wt_taps / ChunkTables this file includes has since been replaced by the 16-wavefront kernel and wt_taps_pk;
// the numbers it produced are profiles/r03a_microbench_phase_b_alone.txt and r03b_microbench_phase_ab_alone.txt)
#include "../ddsp_amd/csrc/harmonic_table.hip"
#include <vector>

namespace ddsp {

constexpr int kMbTiles = 248;         // tiles per wavefront (8 passes over the 31 frames of a chunk)

// VAR 0: the product's code.  VAR 1: select-free folding (th = 0.5 - |theta - 0.5|, the sign by xor), v_fract_f32 +
// v_cvt_flr_i32_f32 for the table coordinate.
template <int W, int NB, int NT, int VAR>
__global__ __launch_bounds__(256 * W, W * NB) void phase_b_kernel(float* __restrict__ audio, long long* __restrict__ clocks, float f0,
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
    unsigned sgn[kWtNT];
    const float* t0p[kWtNT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      theta[u] = (float)__builtin_amdgcn_fract(cyc[u]);
      if constexpr (VAR == 0) {
        neg[u] = theta[u] >= 0.5f;
        const float th = neg[u] ? 1.0f - theta[u] : theta[u];
        const float pos = fmaf(th, (float)kWtT, -0.5f);
        const float fl = floorf(pos);
        z[u] = (pos - fl) - 0.5f;
        z2[u] = z[u] * z[u];
        t0p[u] = tab + q[u] * kWtTS + kWtH + (int)fl;
      } else {
        const float hm = 0.5f - theta[u];                       // sign bit set: theta > 0.5
        sgn[u] = __builtin_bit_cast(unsigned, hm) & 0x80000000u;
        const float pos = fmaf(-fabsf(hm), (float)kWtT, 0.5f * (float)kWtT - 0.5f);     // (0.5 - |hm|) 512 - 0.5
        z[u] = __builtin_amdgcn_fractf(pos) - 0.5f;
        int fl;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(fl) : "v"(pos));
        z2[u] = z[u] * z[u];
        t0p[u] = tab + q[u] * kWtTS + kWtH + fl;
      }
    }
    float acc0[kWtNT] = {0.0f, 0.0f, 0.0f, 0.0f}, acc1[kWtNT] = {0.0f, 0.0f, 0.0f, 0.0f};
    wt_taps<6, 0, NT>(t0p, z, z2, acc0, acc1);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const float lerp = (float)lane * inv_hop;
      const float w_next = amp_linear ? lerp : 0.5f - 0.5f * __builtin_amdgcn_cosf(0.5f * lerp);
      const float w_cur = 1.0f - w_next;
      const float v = w_cur * acc0[u] + w_next * acc1[u];
      if constexpr (VAR == 0) out[(size_t)(tile + u) * 64 + lane] = neg[u] ? -v : v;
      else out[(size_t)(tile + u) * 64 + lane] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ sgn[u]);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) clocks[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

// Phase A of harm_table_kernel alone: rows from the raw staging buffer -> amplitude planes (exp_sigmoid, Nyquist mask,
// normalisation, fp16 hi / lo split).  VAR 0: the product's code.  VAR 1: the mask by v_med3_f32 instead of compare +
// select, the sum of a row pair by v_permlane16_swap instead of four v_readlane.
constexpr int kMbRowPairs = 256;      // (row pair, h) units per wavefront

template <int W, int NB, int VAR>
__global__ __launch_bounds__(256 * W, W * NB) void phase_a_kernel(float* __restrict__ sink, long long* __restrict__ clocks,
                                                                  float f0, float nyquist) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;                                               // [kWtRows][kWtRS]
  _Float16* planes = reinterpret_cast<_Float16*>(smem + kWtRows * kWtRS + 16);      // [hi, lo][parity][row][k']
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < kWtRows * kWtRS; i += blockDim.x) raw[i] = __sinf(0.37f * (float)i);
  for (int r = tid; r < kWtRows; r += blockDim.x) { raw[r * kWtRS + 128] = f0 + 0.5f * (float)r; raw[r * kWtRS + 129] = 0.3f; }
  __syncthreads();
  const int K4 = 25;
  const int sub = lane >> 5, kq = lane & 31;
  const bool live = kq < K4;
  const float kLog10 = 2.302585092994046f;
  float ipsi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) ipsi[u] = live ? WtPoly<6>::invpsi(4 * kq + u + 1) : 0.0f;
  float kf[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) kf[u] = live ? (float)(4 * kq + u + 1) : 1e30f;
  float keep = 0.0f;
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < kMbRowPairs; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int arow = ((it + wave) * 2 + sub + 16 * h) & 31;
      const float4 xv = *reinterpret_cast<const float4*>(raw + arow * kWtRS + 4 * kq);
      const float2 fa2 = *reinterpret_cast<const float2*>(raw + arow * kWtRS + 128);
      const float f0r = fa2.x;
      float x[4] = {xv.x, xv.y, xv.z, xv.w};
      float part;
      if constexpr (VAR == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x[u] = exp_sigmoid_fast(x[u], kLog10, 2.0f, 1e-7f);
          if (!live || f0r * (float)(4 * kq + u + 1) >= nyquist) x[u] = 0.0f;
        }
        part = (x[0] + x[1]) + (x[2] + x[3]);
        part += dpp_mov0<0xB1, 0xF>(part);
        part += dpp_mov0<0x4E, 0xF>(part);
        part += dpp_mov0<0x141, 0xF>(part);
        part += dpp_mov0<0x140, 0xF>(part);
        const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 0));
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 16));
        const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 32));
        const float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 48));
        part = sub ? s2 + s3 : s0 + s1;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x[u] = exp_sigmoid_fast(x[u], kLog10, 2.0f, 1e-7f);
          // x > 0; kept iff f0 k < nyquist (the product rounded to fp32 as TF's is): median(x, 0, +-huge)
          const float y = (nyquist - f0r * kf[u]) * 1e30f;
          x[u] = __builtin_amdgcn_fmed3f(x[u], 0.0f, y);
        }
        part = (x[0] + x[1]) + (x[2] + x[3]);
        part += dpp_mov0<0xB1, 0xF>(part);
        part += dpp_mov0<0x4E, 0xF>(part);
        part += dpp_mov0<0x141, 0xF>(part);
        part += dpp_mov0<0x140, 0xF>(part);
        float other = part;
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(part), "+v"(other));     // part: rows 0 0 2 2, other: rows 1 1 3 3
        part += other;
      }
      const float inv = __builtin_amdgcn_rcpf(part == 0.0f ? 1e-7f : part);
      const float a_ctl = exp_sigmoid_fast(fa2.y, kLog10, 2.0f, 1e-7f);
      const float a = a_ctl * inv;
      float c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = a * x[u] * ipsi[u];
      _Float16* dst = planes + arow * kWtPS + 2 * kq;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const h16x2 hi = __builtin_amdgcn_cvt_pkrtz(c[par], c[par + 2]);
        const h16x2 lo = __builtin_amdgcn_cvt_pkrtz((c[par] - (float)hi[0]) * kWtLoScale, (c[par + 2] - (float)hi[1]) * kWtLoScale);
        *reinterpret_cast<h16x2*>(dst + (0 * 2 + par) * kWtRows * kWtPS) = hi;
        *reinterpret_cast<h16x2*>(dst + (1 * 2 + par) * kWtRows * kWtPS) = lo;
      }
      keep += c[0];
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  sink[(size_t)blockIdx.x * blockDim.x + tid] = keep;
  if (lane == 0) clocks[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

}  // namespace ddsp

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W, int NB, int NT, int VAR>
static void run(float* d_audio, long long* d_clk, int n_cu, float f0) {
  using namespace ddsp;
  const size_t lds = NB == 1 ? 100 * 1024 : 64 * 1024;
  CK(hipFuncSetAttribute((const void*)phase_b_kernel<W, NB, NT, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int threads = 256 * W;
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((phase_b_kernel<W, NB, NT, VAR>), dim3(n_cu * NB), dim3(threads), lds, 0, d_audio, d_clk, f0, 16000.0f, 0);
  CK(hipDeviceSynchronize());
  const int nw = n_cu * NB * 4 * W;
  std::vector<long long> h(nw);
  CK(hipMemcpy(h.data(), d_clk, nw * sizeof(long long), hipMemcpyDeviceToHost));
  double sum = 0, mx = 0;
  for (int i = 0; i < nw; ++i) { sum += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
  const double per_tile_wave = sum / nw / kMbTiles;
  printf("  phase B var %d  W=%d NT=%d f0=%6.1f: %7.1f clocks per tile as a wavefront sees it, %6.1f of SIMD time (slowest wavefront %7.1f)\n",
         VAR, W * NB, NT, f0, per_tile_wave, per_tile_wave / (W * NB), mx / kMbTiles);
  fflush(stdout);
}

template <int W, int NB, int VAR>
static void run_a(float* d_audio, long long* d_clk, int n_cu, float f0) {
  using namespace ddsp;
  const size_t lds = NB == 1 ? 100 * 1024 : 64 * 1024;
  CK(hipFuncSetAttribute((const void*)phase_a_kernel<W, NB, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int threads = 256 * W;
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((phase_a_kernel<W, NB, VAR>), dim3(n_cu * NB), dim3(threads), lds, 0, d_audio, d_clk, f0, 8000.0f);
  CK(hipDeviceSynchronize());
  const int nw = n_cu * NB * 4 * W;
  std::vector<long long> h(nw);
  CK(hipMemcpy(h.data(), d_clk, nw * sizeof(long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < nw; ++i) sum += (double)h[i];
  const double per_row_wave = sum / nw / (kMbRowPairs * 2);       // a (row pair, h) unit = 2 rows
  printf("  phase A var %d  W=%d f0=%6.1f: %7.1f clocks per ROW as a wavefront sees it, %6.1f of SIMD time\n", VAR, W * NB, f0,
         per_row_wave, per_row_wave / (W * NB));
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float* d_audio;
  long long* d_clk;
  CK(hipMalloc(&d_audio, (size_t)n_cu * 32 * ddsp::kMbTiles * 64 * sizeof(float)));
  CK(hipMalloc(&d_clk, (size_t)n_cu * 32 * sizeof(long long)));
  printf("# phase B of harm_table_kernel alone (6 taps, hop 64): clocks per 64-sample tile; W = wavefronts per SIMD\n");
  const float f0s[3] = {70.f, 200.f, 1000.f};
  for (int fi = 0; fi < 3; ++fi) {
    const float f0 = f0s[fi];
    run<2, 1, 4, 0>(d_audio, d_clk, n_cu, f0); run<2, 1, 4, 1>(d_audio, d_clk, n_cu, f0);
    run<3, 1, 2, 0>(d_audio, d_clk, n_cu, f0); run<3, 1, 2, 1>(d_audio, d_clk, n_cu, f0);
    run<4, 1, 2, 0>(d_audio, d_clk, n_cu, f0); run<4, 1, 2, 1>(d_audio, d_clk, n_cu, f0);
    run<4, 1, 1, 1>(d_audio, d_clk, n_cu, f0);
    run<3, 2, 2, 0>(d_audio, d_clk, n_cu, f0); run<3, 2, 2, 1>(d_audio, d_clk, n_cu, f0); run<3, 2, 1, 1>(d_audio, d_clk, n_cu, f0);
    run<4, 2, 1, 0>(d_audio, d_clk, n_cu, f0); run<4, 2, 1, 1>(d_audio, d_clk, n_cu, f0);
  }
  printf("# phase A of harm_table_kernel alone (K = 100): clocks per amplitude row\n");
  for (int fi = 0; fi < 2; ++fi) {
    const float f0 = fi ? 200.f : 70.f;
    run_a<2, 1, 0>(d_audio, d_clk, n_cu, f0); run_a<2, 1, 1>(d_audio, d_clk, n_cu, f0);
    run_a<3, 1, 0>(d_audio, d_clk, n_cu, f0); run_a<3, 1, 1>(d_audio, d_clk, n_cu, f0);
    run_a<4, 1, 0>(d_audio, d_clk, n_cu, f0); run_a<4, 1, 1>(d_audio, d_clk, n_cu, f0);
    run_a<3, 2, 0>(d_audio, d_clk, n_cu, f0); run_a<3, 2, 1>(d_audio, d_clk, n_cu, f0);
    run_a<4, 2, 0>(d_audio, d_clk, n_cu, f0); run_a<4, 2, 1>(d_audio, d_clk, n_cu, f0);
  }
  return 0;
}
