// Can the two per-harmonic FMAs take their amplitude from a LANE of a VGPR (DPP row_newbcast:n)
// instead of from an SGPR?  (1) semantics of row_newbcast on gfx950, (2) issue cost of
// v_fmac_f32_dpp against the SGPR-operand and plain-VGPR forms, in the shape of the synthesis loop
// (per 4 harmonics: 4 recurrence FMAs + 8 accumulate FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void sem_kernel(int* out) {
  const int lane = threadIdx.x;
  int v;
  asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(lane));
  out[lane] = v;
}

#define Q_DPP(N0, N1, N2, N3)                                                                      \
  asm volatile(                                                                                    \
      "v_fma_f32 %[s0], %[c4], %[s2], -%[s0]\n\t"                                                  \
      "v_fma_f32 %[s1], %[c4], %[s3], -%[s1]\n\t"                                                  \
      "v_fma_f32 %[s2], %[c4], %[s0], -%[s2]\n\t"                                                  \
      "v_fmac_f32_dpp %[e0], %[a0], %[s0] row_newbcast:" #N0 " row_mask:0xf bank_mask:0xf\n\t"     \
      "v_fma_f32 %[s3], %[c4], %[s1], -%[s3]\n\t"                                                  \
      "v_fmac_f32_dpp %[e1], %[a1], %[s0] row_newbcast:" #N0 " row_mask:0xf bank_mask:0xf\n\t"     \
      "v_fmac_f32_dpp %[o0], %[a0], %[s1] row_newbcast:" #N1 " row_mask:0xf bank_mask:0xf\n\t"     \
      "v_fmac_f32_dpp %[o1], %[a1], %[s1] row_newbcast:" #N1 " row_mask:0xf bank_mask:0xf\n\t"     \
      "v_fmac_f32_dpp %[e0], %[a0], %[s2] row_newbcast:" #N2 " row_mask:0xf bank_mask:0xf\n\t"     \
      "v_fmac_f32_dpp %[e1], %[a1], %[s2] row_newbcast:" #N2 " row_mask:0xf bank_mask:0xf\n\t"     \
      "v_fmac_f32_dpp %[o0], %[a0], %[s3] row_newbcast:" #N3 " row_mask:0xf bank_mask:0xf\n\t"     \
      "v_fmac_f32_dpp %[o1], %[a1], %[s3] row_newbcast:" #N3 " row_mask:0xf bank_mask:0xf"         \
      : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3), [e0] "+v"(e0), [e1] "+v"(e1),  \
        [o0] "+v"(o0), [o1] "+v"(o1)                                                               \
      : [c4] "v"(c4), [a0] "v"(a0), [a1] "v"(a1))

// MODE 0: SGPR operands (today's loop)  1: DPP row_newbcast  2: plain VGPR operands (bound)
template <int MODE>
__global__ __launch_bounds__(256, 8) void loop_kernel(const float* __restrict__ amp, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  float c4 = 1.9f + 1e-4f * lane, s0 = 0.1f, s1 = 0.2f, s2 = 0.3f, s3 = 0.4f;
  float e0 = 0, e1 = 0, o0 = 0, o1 = 0;
  if (MODE == 1) {
    float a0 = amp[lane & 15], a1 = amp[16 + (lane & 15)];
    for (int it = 0; it < iters; ++it) {
      Q_DPP(0, 1, 2, 3); Q_DPP(4, 5, 6, 7); Q_DPP(8, 9, 10, 11); Q_DPP(12, 13, 14, 15);
    }
  } else if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      typedef float sgpr16 __attribute__((ext_vector_type(16)));
      sgpr16 a0, a1;
      const float* p0 = amp; const float* p1 = amp + 16;
      asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(a0), "=&s"(a1) : "s"(p0), "s"(p1) : "memory");
      float s[4] = {s0, s1, s2, s3};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int u = i & 3;
        const float sn = fmaf(c4, s[(u + 2) & 3], -s[u]);
        s[u] = sn;
        if (u & 1) { o0 = fmaf(a0[i], sn, o0); o1 = fmaf(a1[i], sn, o1); }
        else { e0 = fmaf(a0[i], sn, e0); e1 = fmaf(a1[i], sn, e1); }
      }
      s0 = s[0]; s1 = s[1]; s2 = s[2]; s3 = s[3];
    }
  } else if (MODE == 4) {
    // rows interleaved {a0[k], a1[k]}: one v_pk_fma_f32 per harmonic does both accumulates
    typedef float sgpr16 __attribute__((ext_vector_type(16)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 ev = {0.f, 0.f}, ov = {0.f, 0.f};
    float s[4] = {s0, s1, s2, s3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        sgpr16 a;
        const float* p0 = amp;
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(p0) : "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int u = i & 3;
          const float sn = fmaf(c4, s[(u + 2) & 3], -s[u]);
          s[u] = sn;
          const f2 a2 = {a[2 * i], a[2 * i + 1]};
          const f2 s2 = {sn, sn};
          if (u & 1) ov = __builtin_elementwise_fma(a2, s2, ov);
          else ev = __builtin_elementwise_fma(a2, s2, ev);
        }
      }
    }
    e0 = ev.x; e1 = ev.y; o0 = ov.x; o1 = ov.y;
    s0 = s[0]; s1 = s[1]; s2 = s[2]; s3 = s[3];
  } else if (MODE == 3) {
    // double-buffered s_load_dwordx8: wait for the current buffer (issued one block ago), issue the
    // next, compute - the load latency sits under 24 VALU instructions of this wave
    typedef float sgpr8 __attribute__((ext_vector_type(8)));
    const float* p0 = amp; const float* p1 = amp + 16;
    sgpr8 a0, a1, b0, b1;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0" : "=&s"(a0), "=&s"(a1) : "s"(p0), "s"(p1) : "memory");
    float s[4] = {s0, s1, s2, s3};
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 %0, %4, 0x20\n\ts_load_dwordx8 %1, %5, 0x20"
                   : "=&s"(b0), "=&s"(b1), "+s"(a0), "+s"(a1) : "s"(p0), "s"(p1) : "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int u = i & 3;
        const float sn = fmaf(c4, s[(u + 2) & 3], -s[u]);
        s[u] = sn;
        if (u & 1) { o0 = fmaf(a0[i], sn, o0); o1 = fmaf(a1[i], sn, o1); }
        else { e0 = fmaf(a0[i], sn, e0); e1 = fmaf(a1[i], sn, e1); }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %5, 0x0"
                   : "=&s"(a0), "=&s"(a1), "+s"(b0), "+s"(b1) : "s"(p0), "s"(p1) : "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int u = i & 3;
        const float sn = fmaf(c4, s[(u + 2) & 3], -s[u]);
        s[u] = sn;
        if (u & 1) { o0 = fmaf(b0[i], sn, o0); o1 = fmaf(b1[i], sn, o1); }
        else { e0 = fmaf(b0[i], sn, e0); e1 = fmaf(b1[i], sn, e1); }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
    s0 = s[0] + a0[0]; s1 = s[1]; s2 = s[2]; s3 = s[3];
  } else {
    float a0 = amp[lane & 15], a1 = amp[16 + (lane & 15)];
    for (int it = 0; it < iters; ++it) {
      float s[4] = {s0, s1, s2, s3};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int u = i & 3;
        const float sn = fmaf(c4, s[(u + 2) & 3], -s[u]);
        s[u] = sn;
        if (u & 1) { o0 = fmaf(a0, sn, o0); o1 = fmaf(a1, sn, o1); }
        else { e0 = fmaf(a0, sn, e0); e1 = fmaf(a1, sn, e1); }
      }
      s0 = s[0]; s1 = s[1]; s2 = s[2]; s3 = s[3];
      asm volatile("" : "+v"(a0), "+v"(a1));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = e0 + e1 + o0 + o1 + s0;
}

// numerical check of the DPP block against the scalar formulation
__global__ void check_kernel(const float* __restrict__ amp, float* out) {
  const int lane = threadIdx.x & 63;
  float c4 = 1.9f + 1e-4f * lane, s0 = 0.1f, s1 = 0.2f, s2 = 0.3f, s3 = 0.4f;
  float e0 = 0, e1 = 0, o0 = 0, o1 = 0;
  float a0 = amp[lane & 15], a1 = amp[16 + (lane & 15)];
  Q_DPP(0, 1, 2, 3); Q_DPP(4, 5, 6, 7); Q_DPP(8, 9, 10, 11); Q_DPP(12, 13, 14, 15);
  float r0 = 0, r1 = 0, q0 = 0, q1 = 0;
  float s[4] = {0.1f, 0.2f, 0.3f, 0.4f};
  for (int i = 0; i < 16; ++i) {
    const int u = i & 3;
    const float sn = fmaf(c4, s[(u + 2) & 3], -s[u]);
    s[u] = sn;
    if (u & 1) { q0 = fmaf(amp[i], sn, q0); q1 = fmaf(amp[16 + i], sn, q1); }
    else { r0 = fmaf(amp[i], sn, r0); r1 = fmaf(amp[16 + i], sn, r1); }
  }
  out[lane] = fabsf(e0 - r0) + fabsf(e1 - r1) + fabsf(o0 - q0) + fabsf(o1 - q1);
}

template <int MODE> void run(const char* name, const float* amp, float* out) {
  const int iters = 2000, blocks = 2048;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((loop_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, amp, out, iters);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((loop_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, amp, out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  // per SIMD: 8 waves x iters x 48 VALU (a v_pk_fma counted as the two FMAs it replaces)
  printf("%-28s %.3f ms   %.3f ns per VALU instruction per SIMD\n", name, ms, ms * 1e6 / (8.0 * iters * 48));
}

int main() {
  int* d; (void)hipMalloc(&d, 256);
  hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, d);
  int h[64]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  printf("row_newbcast:5 of lane id:"); for (int i = 0; i < 64; i += 5) printf(" [%d]=%d", i, h[i]); printf("\n");
  float ha[32]; for (int i = 0; i < 32; ++i) ha[i] = 0.01f * (i + 1);
  float *amp, *out; (void)hipMalloc(&amp, 128); (void)hipMemcpy(amp, ha, 128, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 2048 * 256 * 4);
  hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, amp, out);
  float ho[64]; (void)hipMemcpy(ho, out, 256, hipMemcpyDeviceToHost);
  float mx = 0; for (int i = 0; i < 64; ++i) mx = ho[i] > mx ? ho[i] : mx;
  printf("DPP block vs scalar formulation: max abs diff %.3g\n", mx);
  run<0>("SGPR operands (s_load x16)", amp, out);
  run<1>("DPP row_newbcast", amp, out);
  run<2>("plain VGPR operands", amp, out);
  run<3>("SGPR, double-buffered x8", amp, out);
  run<4>("SGPR pairs, v_pk_fma_f32", amp, out);
  return 0;
}
