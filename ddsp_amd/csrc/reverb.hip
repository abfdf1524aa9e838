// Long-impulse-response convolution for gfx950 (MI355X): effects.Reverb (ddsp/effects.py:27-117) and
// the single-frame case of core.fft_convolve (ddsp/core.py:1382-1473).
//
// The reference zero-pads audio [B,N] and one IR [B,L] (L = 48 000 by default) to one FFT of
// 131 072 points.  Here the same linear convolution is evaluated as a uniformly partitioned
// overlap-save convolution whose FFTs (8192 points, 64 KB of float2) live entirely in LDS:
//
//   x blocks   X_j = FFT(x[(j-1)P .. (j+1)P)),  P = 4096, j = 0 .. nb-1      (rv_fft_kernel)
//   IR parts   H_p = FFT([h[pP .. (p+1)P), 0 ... 0]),   p = 0 .. np-1        (rv_fft_kernel)
//   per bin    Y_j = sum_{p <= j} X_{j-p} H_p   - a sliding window over j in registers, every
//              spectrum is read once and Y_j overwrites X_j in place         (rv_mac_kernel)
//   y[jP + i] = IFFT(Y_j)[P + i],  out[n] = y[n + delay] (+ audio[n])        (rv_ifft_kernel)
//
// The forward transform is a radix-2 decimation-in-frequency FFT (natural order in, bit-reversed
// out), the inverse a decimation-in-time FFT (bit-reversed in, natural out): the per-bin products do
// not care about the order, so no bit-reversal pass exists anywhere.  Twiddles come from
// v_sin_f32 / v_cos_f32 on exact binary fractions of a revolution (abs. error 1.2e-7,
// profiles/r01_microbench_alu.txt).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "common.h"
#include "profile.h"
#include "../../include/ddsp_amd.h"

namespace ddsp {

constexpr int kRvP = 4096;             // output samples per block = taps per IR partition
constexpr int kRvN = 2 * kRvP;         // FFT size
constexpr int kRvMaxParts = 16;        // IR partitions held in registers by the MAC kernel
constexpr int kRvThreads = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// natural order in, bit-reversed order out; kernel exp(-2 pi i nk/N)
__device__ __forceinline__ void fft_dif_forward(float2* s, int tid) {
#pragma unroll 1
  for (int half = kRvN / 2; half >= 1; half >>= 1) {
    const float inv_len = 0.5f / (float)half;                 // exact: powers of two
#pragma unroll 4
    for (int t = tid; t < kRvN / 2; t += kRvThreads) {
      const int pos = t & (half - 1);
      const int i0 = ((t - pos) << 1) + pos, i1 = i0 + half;
      const float2 a = s[i0], b = s[i1];
      const float rev = (float)pos * inv_len;                 // revolutions, exact
      const float c = __builtin_amdgcn_cosf(rev), sn = __builtin_amdgcn_sinf(rev);
      const float2 d = make_float2(a.x - b.x, a.y - b.y);
      s[i0] = make_float2(a.x + b.x, a.y + b.y);
      s[i1] = make_float2(fmaf(d.x, c, d.y * sn), fmaf(d.y, c, -d.x * sn));   // d * (c - i sn)
    }
    __syncthreads();
  }
}

// bit-reversed order in, natural order out; kernel exp(+2 pi i nk/N), unscaled
__device__ __forceinline__ void fft_dit_inverse(float2* s, int tid) {
#pragma unroll 1
  for (int half = 1; half <= kRvN / 2; half <<= 1) {
    const float inv_len = 0.5f / (float)half;
#pragma unroll 4
    for (int t = tid; t < kRvN / 2; t += kRvThreads) {
      const int pos = t & (half - 1);
      const int i0 = ((t - pos) << 1) + pos, i1 = i0 + half;
      const float2 a = s[i0], b = s[i1];
      const float rev = (float)pos * inv_len;
      const float c = __builtin_amdgcn_cosf(rev), sn = __builtin_amdgcn_sinf(rev);
      const float2 w = make_float2(fmaf(b.x, c, -b.y * sn), fmaf(b.x, sn, b.y * c));   // b * (c + i sn)
      s[i0] = make_float2(a.x + w.x, a.y + w.y);
      s[i1] = make_float2(a.x - w.x, a.y - w.y);
    }
    __syncthreads();
  }
}

struct RvArgs {
  int N, L, nb, np, delay;
  unsigned flags;
  int ir_batch;                         // 1: one IR for every batch row
};

// blockIdx.x = block / partition, blockIdx.y = batch row.  IS_IR: the rows are IR partitions.
template <bool IS_IR>
__global__ __launch_bounds__(kRvThreads) void rv_fft_kernel(const float* __restrict__ src,
                                                            float2* __restrict__ spec, RvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float2 s[];
  const int tid = threadIdx.x, j = blockIdx.x, b = blockIdx.y;
  const int len = IS_IR ? p.L : p.N;
  const float* __restrict__ row = src + (size_t)b * len;
  // x block: samples (j-1)P .. (j+1)P-1;  IR partition: taps pP .. (p+1)P-1 then P zeros
  const int base = IS_IR ? j * kRvP : (j - 1) * kRvP;
  const int live = IS_IR ? kRvP : kRvN;
  const bool vec = ((len & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  for (int i4 = tid; i4 < kRvN / 4; i4 += kRvThreads) {
    const int i = 4 * i4, g = base + i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < live && g + 3 >= 0 && g < len) {
      if (vec && g >= 0 && g + 3 < len) {
        v = *reinterpret_cast<const float4*>(row + g);
      } else {
        if (g >= 0 && g < len) v.x = row[g];
        if (g + 1 >= 0 && g + 1 < len) v.y = row[g + 1];
        if (g + 2 >= 0 && g + 2 < len) v.z = row[g + 2];
        if (g + 3 >= 0 && g + 3 < len) v.w = row[g + 3];
      }
      // effects.Reverb._mask_dry_ir (effects.py:50-60): tap 0 carries the dry signal -> 0
      if (IS_IR && g == 0 && (p.flags & DDSP_CONV_MASK_TAP0)) v.x = 0.0f;
    }
    reinterpret_cast<float4*>(s)[2 * i4] = make_float4(v.x, 0.f, v.y, 0.f);
    reinterpret_cast<float4*>(s)[2 * i4 + 1] = make_float4(v.z, 0.f, v.w, 0.f);
  }
  __syncthreads();
  fft_dif_forward(s, tid);
  float4* __restrict__ dst = reinterpret_cast<float4*>(spec + ((size_t)b * gridDim.x + j) * kRvN);
  for (int i2 = tid; i2 < kRvN / 2; i2 += kRvThreads) dst[i2] = reinterpret_cast<const float4*>(s)[i2];
}

// One thread per pair of bins (16-byte accesses).  The IR spectra of all partitions sit in
// registers, the x spectra slide through a register window, Y_j replaces X_j in memory.
__global__ __launch_bounds__(kRvThreads) void rv_mac_kernel(float4* __restrict__ xspec,
                                                            const float4* __restrict__ hspec, RvArgs p) {
  const int idx = blockIdx.x * kRvThreads + threadIdx.x;      // < kRvN / 2
  const int b = blockIdx.y;
  const float4* __restrict__ hb = hspec + (size_t)(p.ir_batch == 1 ? 0 : b) * p.np * (kRvN / 2) + idx;
  float4* __restrict__ xb = xspec + (size_t)b * p.nb * (kRvN / 2) + idx;
  float4 h[kRvMaxParts], w[kRvMaxParts];
#pragma unroll
  for (int q = 0; q < kRvMaxParts; ++q) {
    h[q] = (q < p.np) ? hb[(size_t)q * (kRvN / 2)] : make_float4(0.f, 0.f, 0.f, 0.f);
    w[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int j = 0; j < p.nb; ++j) {
#pragma unroll
    for (int q = kRvMaxParts - 1; q > 0; --q) w[q] = w[q - 1];
    w[0] = xb[(size_t)j * (kRvN / 2)];
    float2 y0 = make_float2(0.f, 0.f), y1 = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < kRvMaxParts; ++q) {
      if (q < p.np) {                                          // wave-uniform
        const float2 a0 = cmul(make_float2(w[q].x, w[q].y), make_float2(h[q].x, h[q].y));
        const float2 a1 = cmul(make_float2(w[q].z, w[q].w), make_float2(h[q].z, h[q].w));
        y0.x += a0.x; y0.y += a0.y; y1.x += a1.x; y1.y += a1.y;
      }
    }
    xb[(size_t)j * (kRvN / 2)] = make_float4(y0.x, y0.y, y1.x, y1.y);
  }
}

__global__ __launch_bounds__(kRvThreads) void rv_ifft_kernel(const float2* __restrict__ yspec,
                                                             const float* __restrict__ audio,
                                                             float* __restrict__ out, RvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float2 s[];
  const int tid = threadIdx.x, j = blockIdx.x, b = blockIdx.y;
  const float4* __restrict__ srcv = reinterpret_cast<const float4*>(yspec + ((size_t)b * gridDim.x + j) * kRvN);
  for (int i2 = tid; i2 < kRvN / 2; i2 += kRvThreads) reinterpret_cast<float4*>(s)[i2] = srcv[i2];
  __syncthreads();
  fft_dit_inverse(s, tid);
  // overlap-save: the last P samples of the block are y[jP .. (j+1)P); out[n] = y[n + delay]
  const float scale = 1.0f / (float)kRvN;
  const bool dry = (p.flags & DDSP_CONV_ADD_DRY) != 0;
  const float* __restrict__ arow = audio + (size_t)b * p.N;
  float* __restrict__ orow = out + (size_t)b * p.N;
  for (int i = tid; i < kRvP; i += kRvThreads) {
    const int n = j * kRvP + i - p.delay;
    if (n >= 0 && n < p.N) {
      float v = s[kRvP + i].x * scale;
      if (dry) v += arow[n];
      orow[n] = v;
    }
  }
}

static inline int rv_blocks(int N, int delay) { return (N + delay + kRvP - 1) / kRvP; }
static inline int rv_parts(int L) { return (L + kRvP - 1) / kRvP; }

}  // namespace ddsp

using namespace ddsp;

extern "C" size_t ddsp_fft_convolve_long_workspace_bytes(int B, int Bir, int N, int L, int delay) {
  if (B <= 0 || Bir <= 0 || N <= 0 || L <= 0 || delay < 0) return 0;
  const size_t spectra = (size_t)B * rv_blocks(N, delay) + (size_t)Bir * rv_parts(L);
  return spectra * kRvN * sizeof(float2);
}

extern "C" int ddsp_fft_convolve_long_f32(const float* audio, const float* impulse_response,
                                          float* out, void* workspace, size_t workspace_bytes,
                                          int B, int Bir, int N, int L, int delay, unsigned flags,
                                          void* stream) {
  if (!audio || !impulse_response || !out || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || L <= 0 || delay < 0 || (Bir != B && Bir != 1)) return DDSP_ERR_BAD_SHAPE;
  if (B > 65535 || rv_parts(L) > kRvMaxParts) return DDSP_ERR_UNSUPPORTED;
  if (workspace_bytes < ddsp_fft_convolve_long_workspace_bytes(B, Bir, N, L, delay) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  RvArgs p;
  p.N = N; p.L = L; p.nb = rv_blocks(N, delay); p.np = rv_parts(L); p.delay = delay;
  p.flags = flags; p.ir_batch = Bir;
  float2* xspec = (float2*)workspace;
  float2* hspec = xspec + (size_t)B * p.nb * kRvN;
  const size_t lds = (size_t)kRvN * sizeof(float2);
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute((const void*)rv_fft_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRvN * sizeof(float2)));
    (void)hipFuncSetAttribute((const void*)rv_fft_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRvN * sizeof(float2)));
    (void)hipFuncSetAttribute((const void*)rv_ifft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRvN * sizeof(float2)));
    return true;
  }();
  (void)attr_set;
  {
    ProfileScope prof(kReverbFft, st);
    hipLaunchKernelGGL((rv_fft_kernel<true>), dim3((unsigned)p.np, (unsigned)Bir), dim3(kRvThreads), lds, st,
                       impulse_response, hspec, p);
    hipLaunchKernelGGL((rv_fft_kernel<false>), dim3((unsigned)p.nb, (unsigned)B), dim3(kRvThreads), lds, st,
                       audio, xspec, p);
  }
  {
    ProfileScope prof(kReverbMac, st);
    hipLaunchKernelGGL(rv_mac_kernel, dim3(kRvN / 2 / kRvThreads, (unsigned)B), dim3(kRvThreads), 0, st,
                       (float4*)xspec, (const float4*)hspec, p);
  }
  {
    ProfileScope prof(kReverbIfft, st);
    hipLaunchKernelGGL(rv_ifft_kernel, dim3((unsigned)p.nb, (unsigned)B), dim3(kRvThreads), lds, st,
                       (const float2*)xspec, audio, out, p);
  }
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}
