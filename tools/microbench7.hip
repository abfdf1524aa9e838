// Issue cost of every instruction class the two default kernels are made of, as a function of how many wavefronts
// share a SIMD ("a per-class cycle account ... at 2-3 wavefronts / SIMD").
//
// One block per CU (the block asks for 100 KB of LDS), W wavefronts per SIMD (block = 256 W threads); every wavefront
// runs ITERS iterations of 16 independent instances of one instruction (inline asm: exactly that instruction) between
// two s_memtime stamps (shader clocks).  Reported per class and W:
//   wave  = clocks per instruction as one wavefront sees it (its own issue interval)
//   simd  = wave / W = clocks of SIMD time per wavefront-instruction (the throughput price)
// "dep" classes run ONE dependent chain instead of 16 independent ones (the latency).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench7.hip -o tools/bin/microbench7
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum Cls {
  FMA_F32, FMAAK_F32, MUL_F32, ADD_U32, XOR_B32, LSHL_ADD, CNDMASK, FLOOR_F32, CVT_I32_F32, CVT_F32_I32,
  FMA_F64, FRACT_F64, CVT_F32_F64, CVT_F64_F32, CMP_F64, ADD_F64,
  EXP_F32, LOG_F32, RCP_F32, SIN_F32, COS_F32,
  CVT_PKRTZ, CVT_F32_F16, CVT_F16_F32, PK_FMA_F32,
  MOV_DPP, ADD_DPP, READLANE, READFIRSTLANE, PERMLANE32_SWAP,
  MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, ALIGNBIT,
  S_MOV, FMA_PLUS_SALU, FMA_SGPR_OPERAND,
  DS_READ2_B32, DS_READ2_B32_STRIDE6, DS_READ2_B32_STRIDE16, DS_READ_B32, DS_READ_B64, DS_READ_B128,
  DS_WRITE_B32, DS_WRITE_B64, DS_WRITE_B128, DS_BPERMUTE,
  MFMA_16X16X32_F16, MFMA_PLUS_8FMA,
  DEP_FMA_F32, DEP_FMA_F64, DEP_EXP_F32, DEP_DS_READ_B32,
  N_CLS
};
static const char* kNames[N_CLS] = {
  "v_fma_f32", "v_fmaak_f32 (literal)", "v_mul_f32", "v_add_u32", "v_xor_b32", "v_lshl_add_u32", "v_cndmask_b32", "v_floor_f32",
  "v_cvt_i32_f32", "v_cvt_f32_i32",
  "v_fma_f64", "v_fract_f64", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_cmp_le_f64", "v_add_f64",
  "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sin_f32", "v_cos_f32",
  "v_cvt_pkrtz_f16_f32", "v_cvt_f32_f16", "v_cvt_f16_f32", "v_pk_fma_f32",
  "v_mov_b32_dpp row_shr:1", "v_add_f32_dpp row_shr:1", "v_readlane_b32", "v_readfirstlane_b32", "v_permlane32_swap",
  "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_alignbit_b32",
  "s_mov_b32", "v_fma_f32 + s_add_u32 (pairs; per pair)", "v_fma_f32 with SGPR operand",
  "ds_read2_b32 (conflict free)", "ds_read2_b32 (lane stride 6.4 dwords)", "ds_read2_b32 (lane stride 16 dwords)",
  "ds_read_b32", "ds_read_b64", "ds_read_b128",
  "ds_write_b32", "ds_write_b64", "ds_write_b128", "ds_bpermute_b32",
  "v_mfma_f32_16x16x32_f16", "v_mfma + 8 v_fma_f32 (per group of 9)",
  "dep v_fma_f32 chain", "dep v_fma_f64 chain", "dep v_exp_f32 chain", "dep ds_read_b32 chain",
};

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int C>
__global__ __launch_bounds__(1024) void cls_kernel(long long* __restrict__ clocks, float* __restrict__ sink, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = (float)(i & 1023) * 4.0f;     // also valid byte offsets for the chain
  __syncthreads();
  float a[16];
  double d[16];
  unsigned u[16];
  unsigned long long q[16];
  f32x4 acc[4];
  f16x8 fa, fb;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = seed + (float)(lane + i) * 1e-3f;
    d[i] = (double)a[i] * 1.000001;
    u[i] = (unsigned)(lane * 977 + i * 13 + 1);
    q[i] = (unsigned long long)u[i] * 0x9E3779B97F4A7C15ull;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (float)(lane + i)); fb[i] = (_Float16)(0.002f * (float)(lane - i)); }
  const float b = seed * 0.999f + 1.0f, c = seed * 0.5f;
  const double db = 0.9999991, dc = 1e-7;
  float sg = seed * 3.0f;
  sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sg)));
  // LDS byte addresses: conflict free (lane * 4, second dword 64 lanes on), table-read like strides
  unsigned addr_cf = (unsigned)lane * 4u;
  unsigned addr_s6 = ((unsigned)((float)lane * 6.4f)) * 4u;
  unsigned addr_s16 = (unsigned)lane * 64u;
  unsigned addr8 = (unsigned)lane * 8u, addr16 = (unsigned)lane * 16u;
  unsigned chain = (unsigned)lane * 4u;
  int sacc = 0;
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_FMAAK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f510b5f" : "+v"(a[i]) : "v"(b));
#define X_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define X_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define X_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define X_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define X_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
#define X_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define X_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define X_CVTF(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
#define X_FRACT64(i) asm volatile("v_fract_f64 %0, %0" : "+v"(d[i]));
#define X_CVT3264(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define X_CVT6432(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define X_CMP64(i) asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(d[i]), "v"(db) : "vcc");
#define X_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
#define X_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define X_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define X_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define X_SIN(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
#define X_COS(i) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));
#define X_PKRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b));
#define X_CVT_F32_F16(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define X_CVT_F16_F32(i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(db));
#define X_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
#define X_ADDDPP(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
#define X_READLANE(i) { int s_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s_) : "v"(a[i])); sacc ^= s_; }
#define X_READFIRST(i) { int s_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s_) : "v"(a[i])); sacc ^= s_; }
#define X_PERMSWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 8) & 15]));
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]) : "vcc");
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define X_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define X_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(u[i]));
#define X_SMOV(i) { int s_; asm volatile("s_mov_b32 %0, 0x1234567" : "=s"(s_)); sacc ^= s_; }
#define X_FMASALU(i) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); asm volatile("s_add_u32 %0, %0, 0x1234567" : "+s"(sacc) : : "scc"); }
#define X_FMASGPR(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sg), "v"(c));
#define X_DSR2(i) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(q[i]) : "v"(addr_cf), "n"(i * 2), "n"(64 + i * 2));
#define X_DSR2S6(i) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(q[i]) : "v"(addr_s6), "n"(i), "n"(i + 1));
#define X_DSR2S16(i) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(q[i]) : "v"(addr_s16), "n"(i), "n"(i + 1));
#define X_DSR32(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(u[i]) : "v"(addr_cf), "n"(i * 256));
#define X_DSR64(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[i]) : "v"(addr8), "n"(i * 512));
#define X_DSW32(i) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr_cf), "v"(a[i]), "n"(i * 256));
#define X_DSW64(i) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr8), "v"(d[i]), "n"(i * 512));
#define X_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(u[i]) : "v"(addr_cf), "v"(a[i]));
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (C == FMA_F32) { REP16(X_FMA) }
    if constexpr (C == FMAAK_F32) { REP16(X_FMAAK) }
    if constexpr (C == MUL_F32) { REP16(X_MUL) }
    if constexpr (C == ADD_U32) { REP16(X_ADDU) }
    if constexpr (C == XOR_B32) { REP16(X_XOR) }
    if constexpr (C == LSHL_ADD) { REP16(X_LSHLADD) }
    if constexpr (C == CNDMASK) { REP16(X_CNDMASK) }
    if constexpr (C == FLOOR_F32) { REP16(X_FLOOR) }
    if constexpr (C == CVT_I32_F32) { REP16(X_CVTI) }
    if constexpr (C == CVT_F32_I32) { REP16(X_CVTF) }
    if constexpr (C == FMA_F64) { REP16(X_FMA64) }
    if constexpr (C == FRACT_F64) { REP16(X_FRACT64) }
    if constexpr (C == CVT_F32_F64) { REP16(X_CVT3264) }
    if constexpr (C == CVT_F64_F32) { REP16(X_CVT6432) }
    if constexpr (C == CMP_F64) { REP16(X_CMP64) }
    if constexpr (C == ADD_F64) { REP16(X_ADD64) }
    if constexpr (C == EXP_F32) { REP16(X_EXP) }
    if constexpr (C == LOG_F32) { REP16(X_LOG) }
    if constexpr (C == RCP_F32) { REP16(X_RCP) }
    if constexpr (C == SIN_F32) { REP16(X_SIN) }
    if constexpr (C == COS_F32) { REP16(X_COS) }
    if constexpr (C == CVT_PKRTZ) { REP16(X_PKRTZ) }
    if constexpr (C == CVT_F32_F16) { REP16(X_CVT_F32_F16) }
    if constexpr (C == CVT_F16_F32) { REP16(X_CVT_F16_F32) }
    if constexpr (C == PK_FMA_F32) { REP16(X_PKFMA) }
    if constexpr (C == MOV_DPP) { REP16(X_MOVDPP) }
    if constexpr (C == ADD_DPP) { REP16(X_ADDDPP) }
    if constexpr (C == READLANE) { REP16(X_READLANE) }
    if constexpr (C == READFIRSTLANE) { REP16(X_READFIRST) }
    if constexpr (C == PERMLANE32_SWAP) { REP16(X_PERMSWAP) }
    if constexpr (C == MAD_U64_U32) { REP16(X_MAD64) }
    if constexpr (C == MUL_LO_U32) { REP16(X_MULLO) }
    if constexpr (C == MUL_HI_U32) { REP16(X_MULHI) }
    if constexpr (C == MAD_U32_U24) { REP16(X_MAD24) }
    if constexpr (C == ALIGNBIT) { REP16(X_ALIGNBIT) }
    if constexpr (C == S_MOV) { REP16(X_SMOV) }
    if constexpr (C == FMA_PLUS_SALU) { REP16(X_FMASALU) }
    if constexpr (C == FMA_SGPR_OPERAND) { REP16(X_FMASGPR) }
    if constexpr (C == DS_READ2_B32) { REP16(X_DSR2) LGKM0 }
    if constexpr (C == DS_READ2_B32_STRIDE6) { REP16(X_DSR2S6) LGKM0 }
    if constexpr (C == DS_READ2_B32_STRIDE16) { REP16(X_DSR2S16) LGKM0 }
    if constexpr (C == DS_READ_B32) { REP16(X_DSR32) LGKM0 }
    if constexpr (C == DS_READ_B64) { REP16(X_DSR64) LGKM0 }
    if constexpr (C == DS_READ_B128) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr16), "n"(0));
        asm volatile("" : : "v"(v));
      }
      LGKM0
    }
    if constexpr (C == DS_WRITE_B32) { REP16(X_DSW32) LGKM0 }
    if constexpr (C == DS_WRITE_B64) { REP16(X_DSW64) LGKM0 }
    if constexpr (C == DS_WRITE_B128) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("ds_write_b128 %0, %1" : : "v"(addr16), "v"(acc[i & 3]));
      LGKM0
    }
    if constexpr (C == DS_BPERMUTE) { REP16(X_BPERM) LGKM0 }
    if constexpr (C == MFMA_16X16X32_F16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(fa), "v"(fb));
    }
    if constexpr (C == MFMA_PLUS_8FMA) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(fa), "v"(fb));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[8 * g + i]) : "v"(b), "v"(c));
      }
    }
    if constexpr (C == DEP_FMA_F32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
    }
    if constexpr (C == DEP_FMA_F64) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(db), "v"(dc));
    }
    if constexpr (C == DEP_EXP_F32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));
    }
    if constexpr (C == DEP_DS_READ_B32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v;
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(chain) : "memory");
        asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(chain) : "v"(v));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = (float)sacc + (float)chain;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a[i] + (float)d[i] + (float)u[i] + (float)(unsigned)q[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[(size_t)blockIdx.x * blockDim.x + tid] = r;
  if (lane == 0) clocks[(size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

template <int C>
static void run_cls(long long* d_clk, float* d_sink, int n_cu, FILE* out) {
  const int Ws[5] = {1, 2, 3, 4, 8 > 4 ? 4 : 4};      // (a 1024-thread block is 4 per SIMD; 8 per SIMD: two blocks)
  double wave_c[6], simd_c[6];
  int nW = 0;
  for (int wi = 0; wi < 5; ++wi) {
    const int W = wi < 4 ? Ws[wi] : 8;
    const int blocks_per_cu = W == 8 ? 2 : 1;
    const int threads = 256 * (W == 8 ? 4 : W);
    const size_t lds = W == 8 ? 64 * 1024 : 100 * 1024;       // one (two) block(s) per CU
    CK(hipFuncSetAttribute((const void*)cls_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = n_cu * blocks_per_cu;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((cls_kernel<C>), dim3(grid), dim3(threads), lds, 0, d_clk, d_sink, 0.37f);
    CK(hipDeviceSynchronize());
    const int nw = grid * (threads / 64);
    std::vector<long long> h(nw);
    CK(hipMemcpy(h.data(), d_clk, nw * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < nw; ++i) sum += (double)h[i];
    const double per_instr = sum / nw / ((double)ITERS * 16.0);
    wave_c[nW] = per_instr;
    simd_c[nW] = per_instr / W;
    ++nW;
  }
  fprintf(out, "%-44s", kNames[C]);
  for (int i = 0; i < nW; ++i) fprintf(out, "  %6.2f/%5.2f", wave_c[i], simd_c[i]);
  fprintf(out, "\n");
  fflush(out);
}

template <int C>
static void run_all(long long* d_clk, float* d_sink, int n_cu, FILE* out) {
  if constexpr (C < N_CLS) {
    run_cls<C>(d_clk, d_sink, n_cu, out);
    run_all<C + 1>(d_clk, d_sink, n_cu, out);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  long long* d_clk;
  float* d_sink;
  CK(hipMalloc(&d_clk, (size_t)n_cu * 2 * 16 * sizeof(long long)));
  CK(hipMalloc(&d_sink, (size_t)n_cu * 2 * 1024 * sizeof(float)));
  printf("# %s, %d CUs.  clocks per wavefront-instruction: as one wavefront sees it / of SIMD time (= wave / W)\n", prop.name, n_cu);
  printf("# 16 independent instances per iteration unless 'dep'; s_memtime (shader clocks) around %d iterations\n", ITERS);
  printf("%-44s  %12s  %12s  %12s  %12s  %12s\n", "class \\ wavefronts per SIMD", "W=1", "W=2", "W=3", "W=4", "W=8");
  run_all<0>(d_clk, d_sink, n_cu, stdout);
  return 0;
}
