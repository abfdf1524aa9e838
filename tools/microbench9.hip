// Second pass of the per-class issue-cost table (tools/microbench7.hip): the classes the first pass raised questions
// about (v_cndmask forms, compares, select-free alternatives, VOP3 modifiers, SALU kinds, scalar loads), with 8
// independent chains and at most 64 VGPRs so that 6 and 8 wavefronts per SIMD are really resident (the first pass
// needed ~110 VGPRs: its "W=8" column was two blocks of 4 per SIMD running one after the other).
// Reported per class and W: clocks per instruction as one wavefront sees it / of SIMD time (= wave / W).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench9.hip -o tools/bin/microbench9
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 1024;

enum Cls {
  FMA_F32, FMA_F32_NEGABS, CNDMASK_VCC, CNDMASK_E64, CMP_THEN_CNDMASK, CMP_F32_VCC, CMP_F32_SGPR, MED3_F32, BFI_B32, MAX_F32, AND_OR,
  FRACT_F32, CVT_FLR_I32, FMA_F64, CVT_F32_U32, MUL_LO_U32, LSHLREV, SUB_F32_ABS,
  EXP_F32, READLANE, READFIRSTLANE, MOV_DPP, PERMLANE32_SWAP, BPERMUTE,
  S_MOV, S_ADD, S_WAITCNT, S_NOP, S_CSELECT, S_LOAD, S_MUL,
  FMA_PLUS_SADD, FMA_PLUS_SNOP, FMA_PLUS_WAITCNT, FMA2_PLUS_SADD, FMA4_PLUS_SADD,
  MFMA_16X16X32, MFMA_PLUS_4FMA,
  DS_READ2_B32, DS_READ_B64, DS_READ_B128, DS_WRITE_B32, DS_WRITE_B64, DS_WRITE_B128,
  DSR2_PLUS_8FMA,
  N_CLS
};
static const char* kNames[N_CLS] = {
  "v_fma_f32", "v_fma_f32 with -|x| modifiers", "v_cndmask_b32 (vcc, never written)", "v_cndmask_b32_e64 (SGPR pair)",
  "v_cmp_lt_f32 vcc ; v_cndmask vcc (per pair)", "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32_e64 -> SGPR pair", "v_med3_f32", "v_bfi_b32", "v_max_f32", "v_and_or_b32",
  "v_fract_f32", "v_cvt_flr_i32_f32", "v_fma_f64", "v_cvt_f32_u32", "v_mul_lo_u32", "v_lshlrev_b32", "v_sub_f32 with |x|",
  "v_exp_f32", "v_readlane_b32", "v_readfirstlane_b32", "v_mov_b32_dpp row_shr:1", "v_permlane32_swap", "ds_bpermute_b32",
  "s_mov_b32", "s_add_u32", "s_waitcnt lgkmcnt(0) (nothing pending)", "s_nop 0", "s_cselect_b32", "s_load_dword (same line) + wait", "s_mul_i32",
  "v_fma_f32 + s_add_u32 (per pair)", "v_fma_f32 + s_nop (per pair)", "v_fma_f32 + s_waitcnt (per pair)", "2 v_fma_f32 + s_add_u32 (per triple)",
  "4 v_fma_f32 + s_add_u32 (per group of 5)",
  "v_mfma_f32_16x16x32_f16", "v_mfma + 4 v_fma_f32 (per group of 5)",
  "ds_read2_b32 (conflict free)", "ds_read_b64", "ds_read_b128", "ds_write_b32", "ds_write_b64", "ds_write_b128",
  "ds_read2_b32 + 8 v_fma_f32 (per group of 9)",
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int C>
__global__ __launch_bounds__(1024, 8) void cls_kernel(long long* __restrict__ clocks, float* __restrict__ sink, float seed,
                                                      const int* __restrict__ gconst) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (float)(i & 1023) * 4.0f;
  __syncthreads();
  float a[8];
  unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + (float)(lane + i) * 1e-3f;
    u[i] = (unsigned)(lane * 977 + i * 13 + 1);
  }
  double d0 = (double)a[0] * 1.000001, d1 = (double)a[1] * 1.000002, d2 = (double)a[2], d3 = (double)a[3];
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f16x8 fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (float)(lane + i)); fb[i] = (_Float16)(0.002f * (float)(lane - i)); }
  const float b = seed * 0.999f + 1.0f, c = seed * 0.5f;
  const double db = 0.9999991, dc = 1e-7;
  unsigned addr_cf = (unsigned)lane * 4u, addr8 = (unsigned)lane * 8u, addr16 = (unsigned)lane * 16u;
  int sacc = 0;
  unsigned long long smask = 0x5555555555555555ull;
  asm volatile("" : "+s"(smask));
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_FMAMOD(i) asm volatile("v_fma_f32 %0, -|%0|, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_CNDVCC(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define X_CNDE64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(smask));
#define X_CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define X_CMPVCC(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define X_CMPSG(i) { unsigned long long m_; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m_) : "v"(a[i]), "v"(b)); }
#define X_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define X_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define X_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(b), "v"(c));
#define X_FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
#define X_CVTFLR(i) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define X_CVTU(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define X_LSHL(i) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(u[i]));
#define X_SUBABS(i) asm volatile("v_sub_f32_e64 %0, %1, |%0|" : "+v"(a[i]) : "v"(b));
#define X_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define X_READLANE(i) { int s_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s_) : "v"(a[i])); }
#define X_READFIRST(i) { int s_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s_) : "v"(a[i])); }
#define X_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
#define X_PERMSWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]));
#define X_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(u[i]) : "v"(addr_cf), "v"(a[i]));
#define X_SMOV(i) { int s_; asm volatile("s_mov_b32 %0, 0x1234567" : "=s"(s_)); }
#define X_SADD(i) asm volatile("s_add_u32 %0, %0, 0x1234567" : "+s"(sacc) : : "scc");
#define X_SWAIT(i) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define X_SNOP(i) asm volatile("s_nop 0");
#define X_SCSEL(i) asm volatile("s_cselect_b32 %0, %0, 0x1234567" : "+s"(sacc));
#define X_SLOAD(i) { int s_; asm volatile("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(s_) : "s"(gconst) : "memory"); sacc ^= s_; }
#define X_SMUL(i) asm volatile("s_mul_i32 %0, %0, 0x1234567" : "+s"(sacc));
#define X_FMASADD(i) { X_FMA(i) X_SADD(i) }
#define X_FMASNOP(i) { X_FMA(i) X_SNOP(i) }
#define X_FMAWAIT(i) { X_FMA(i) X_SWAIT(i) }
#define X_DSR2(i) { unsigned long long q_; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(q_) : "v"(addr_cf), "n"(i * 2), "n"(64 + i * 2)); asm volatile("" : : "v"(q_)); }
#define X_DSR64(i) { unsigned long long q_; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q_) : "v"(addr8), "n"(i * 512)); asm volatile("" : : "v"(q_)); }
#define X_DSR128(i) { f32x4 q_; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q_) : "v"(addr16), "n"(i * 1024)); asm volatile("" : : "v"(q_)); }
#define X_DSW32(i) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr_cf), "v"(a[i]), "n"(i * 256));
#define X_DSW64(i) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr8), "v"(d0), "n"(i * 512));
#define X_DSW128(i) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr16), "v"(acc0), "n"(i * 1024));
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (C == FMA_F32) { REP8(X_FMA) REP8(X_FMA) }
    if constexpr (C == FMA_F32_NEGABS) { REP8(X_FMAMOD) REP8(X_FMAMOD) }
    if constexpr (C == CNDMASK_VCC) { REP8(X_CNDVCC) REP8(X_CNDVCC) }
    if constexpr (C == CNDMASK_E64) { REP8(X_CNDE64) REP8(X_CNDE64) }
    if constexpr (C == CMP_THEN_CNDMASK) { REP8(X_CMPCND) REP8(X_CMPCND) }
    if constexpr (C == CMP_F32_VCC) { REP8(X_CMPVCC) REP8(X_CMPVCC) }
    if constexpr (C == CMP_F32_SGPR) { REP8(X_CMPSG) REP8(X_CMPSG) }
    if constexpr (C == MED3_F32) { REP8(X_MED3) REP8(X_MED3) }
    if constexpr (C == BFI_B32) { REP8(X_BFI) REP8(X_BFI) }
    if constexpr (C == MAX_F32) { REP8(X_MAX) REP8(X_MAX) }
    if constexpr (C == AND_OR) { REP8(X_ANDOR) REP8(X_ANDOR) }
    if constexpr (C == FRACT_F32) { REP8(X_FRACT) REP8(X_FRACT) }
    if constexpr (C == CVT_FLR_I32) { REP8(X_CVTFLR) REP8(X_CVTFLR) }
    if constexpr (C == FMA_F64) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(db), "v"(dc));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(db), "v"(dc));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2) : "v"(db), "v"(dc));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d3) : "v"(db), "v"(dc));
      }
    }
    if constexpr (C == CVT_F32_U32) { REP8(X_CVTU) REP8(X_CVTU) }
    if constexpr (C == MUL_LO_U32) { REP8(X_MULLO) REP8(X_MULLO) }
    if constexpr (C == LSHLREV) { REP8(X_LSHL) REP8(X_LSHL) }
    if constexpr (C == SUB_F32_ABS) { REP8(X_SUBABS) REP8(X_SUBABS) }
    if constexpr (C == EXP_F32) { REP8(X_EXP) REP8(X_EXP) }
    if constexpr (C == READLANE) { REP8(X_READLANE) REP8(X_READLANE) }
    if constexpr (C == READFIRSTLANE) { REP8(X_READFIRST) REP8(X_READFIRST) }
    if constexpr (C == MOV_DPP) { REP8(X_MOVDPP) REP8(X_MOVDPP) }
    if constexpr (C == PERMLANE32_SWAP) { REP8(X_PERMSWAP) REP8(X_PERMSWAP) }
    if constexpr (C == BPERMUTE) { REP8(X_BPERM) REP8(X_BPERM) LGKM0 }
    if constexpr (C == S_MOV) { REP8(X_SMOV) REP8(X_SMOV) }
    if constexpr (C == S_ADD) { REP8(X_SADD) REP8(X_SADD) }
    if constexpr (C == S_WAITCNT) { REP8(X_SWAIT) REP8(X_SWAIT) }
    if constexpr (C == S_NOP) { REP8(X_SNOP) REP8(X_SNOP) }
    if constexpr (C == S_CSELECT) { REP8(X_SCSEL) REP8(X_SCSEL) }
    if constexpr (C == S_LOAD) { REP8(X_SLOAD) REP8(X_SLOAD) }
    if constexpr (C == S_MUL) { REP8(X_SMUL) REP8(X_SMUL) }
    if constexpr (C == FMA_PLUS_SADD) { REP8(X_FMASADD) REP8(X_FMASADD) }
    if constexpr (C == FMA_PLUS_SNOP) { REP8(X_FMASNOP) REP8(X_FMASNOP) }
    if constexpr (C == FMA_PLUS_WAITCNT) { REP8(X_FMAWAIT) REP8(X_FMAWAIT) }
    if constexpr (C == FMA2_PLUS_SADD) {
#pragma unroll
      for (int g = 0; g < 8; ++g) { X_FMA(g) X_FMA((g + 1) & 7) X_SADD(0) }
      X_FMA(0) X_FMA(1) X_FMA(2) X_FMA(3) X_FMA(4) X_FMA(5) X_FMA(6) X_FMA(7)      // (24 + 8 = 32 instructions: 16 "triples" of 2 slots)
    }
    if constexpr (C == FMA4_PLUS_SADD) {
#pragma unroll
      for (int g = 0; g < 4; ++g) { X_FMA(0) X_FMA(1) X_FMA(2) X_FMA(3) X_SADD(0) X_FMA(4) X_FMA(5) X_FMA(6) X_FMA(7) X_SADD(0) }
    }
    if constexpr (C == MFMA_16X16X32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(fa), "v"(fb));
      }
    }
    if constexpr (C == MFMA_PLUS_4FMA) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fa), "v"(fb));
        X_FMA(0) X_FMA(1) X_FMA(2) X_FMA(3)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(fa), "v"(fb));
        X_FMA(4) X_FMA(5) X_FMA(6) X_FMA(7)
      }
    }
    if constexpr (C == DS_READ2_B32) { REP8(X_DSR2) REP8(X_DSR2) LGKM0 }
    if constexpr (C == DS_READ_B64) { REP8(X_DSR64) REP8(X_DSR64) LGKM0 }
    if constexpr (C == DS_READ_B128) { REP8(X_DSR128) REP8(X_DSR128) LGKM0 }
    if constexpr (C == DS_WRITE_B32) { REP8(X_DSW32) REP8(X_DSW32) LGKM0 }
    if constexpr (C == DS_WRITE_B64) { REP8(X_DSW64) REP8(X_DSW64) LGKM0 }
    if constexpr (C == DS_WRITE_B128) { REP8(X_DSW128) REP8(X_DSW128) LGKM0 }
    if constexpr (C == DSR2_PLUS_8FMA) {
#pragma unroll
      for (int g = 0; g < 2; ++g) { X_DSR2(g) REP8(X_FMA) }
      LGKM0
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = (float)sacc + (float)d0 + (float)d1 + (float)d2 + (float)d3;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i] + (float)u[i];
  r += acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  sink[(size_t)blockIdx.x * blockDim.x + tid] = r;
  if (lane == 0) clocks[(size_t)blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

static int instr_per_iter(int c) {
  switch (c) {
    case FMA_F64: return 16;
    case CMP_THEN_CNDMASK: case FMA_PLUS_SADD: case FMA_PLUS_SNOP: case FMA_PLUS_WAITCNT: return 16;     // pairs
    case FMA2_PLUS_SADD: return 16;
    case FMA4_PLUS_SADD: return 8;
    case MFMA_PLUS_4FMA: return 8;
    case DSR2_PLUS_8FMA: return 2;
    default: return 16;
  }
}

template <int C>
static void run_cls(long long* d_clk, float* d_sink, const int* d_const, int n_cu) {
  const int Wb[6] = {1, 2, 3, 4, 3, 4}, NB[6] = {1, 1, 1, 1, 2, 2};       // W = Wb * NB = 1 2 3 4 6 8
  printf("%-46s", kNames[C]);
  for (int wi = 0; wi < 6; ++wi) {
    const int W = Wb[wi] * NB[wi];
    const int threads = 256 * Wb[wi];
    const size_t lds = NB[wi] == 1 ? 100 * 1024 : 64 * 1024;
    CK(hipFuncSetAttribute((const void*)cls_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = n_cu * NB[wi];
    for (int rep = 0; rep < 2; ++rep)
      hipLaunchKernelGGL((cls_kernel<C>), dim3(grid), dim3(threads), lds, 0, d_clk, d_sink, 0.37f, d_const);
    CK(hipDeviceSynchronize());
    const int nw = grid * (threads / 64);
    std::vector<long long> h(nw);
    CK(hipMemcpy(h.data(), d_clk, nw * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < nw; ++i) sum += (double)h[i];
    const double per = sum / nw / ((double)ITERS * instr_per_iter(C));
    printf("  %6.2f/%5.2f", per, per / W);
  }
  printf("\n");
  fflush(stdout);
}

template <int C>
static void run_all(long long* d_clk, float* d_sink, const int* d_const, int n_cu) {
  if constexpr (C < N_CLS) {
    run_cls<C>(d_clk, d_sink, d_const, n_cu);
    run_all<C + 1>(d_clk, d_sink, d_const, n_cu);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  long long* d_clk;
  float* d_sink;
  int* d_const;
  CK(hipMalloc(&d_clk, (size_t)n_cu * 2 * 16 * sizeof(long long)));
  CK(hipMalloc(&d_sink, (size_t)n_cu * 2 * 1024 * sizeof(float)));
  CK(hipMalloc(&d_const, 256));
  CK(hipMemset(d_const, 0, 256));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)cls_kernel<FMA_F32>, 1024, 64 * 1024));
  printf("# %d CUs; occupancy API: %d blocks of 1024 threads per CU with 64 KB of LDS each (2 = the W=8 column is real)\n", n_cu, nb);
  printf("# clocks per wavefront-instruction (or per group where the name says so): as one wavefront sees it / of SIMD time\n");
  printf("%-46s  %12s  %12s  %12s  %12s  %12s  %12s\n", "class \\ wavefronts per SIMD", "W=1", "W=2", "W=3", "W=4", "W=6", "W=8");
  run_all<0>(d_clk, d_sink, d_const, n_cu);
  return 0;
}
