// Shared device helpers for the gfx950 kernels (wave64 only; no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddsp {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

// ---- DPP reductions: no LDS round trip (ds_bpermute costs ~100 cycles per dependent step) ---------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov0(float v) {    // lanes outside ROW_MASK read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes, result in every lane (as a wave-uniform SGPR value)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov0<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov0<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov0<0x141, 0xF>(v);    // row_half_mirror
  v += dpp_mov0<0x140, 0xF>(v);    // row_mirror: every lane holds its 16-lane row's sum
  v += dpp_mov0<0x142, 0xA>(v);    // row_bcast:15 -> rows 1,3
  v += dpp_mov0<0x143, 0xC>(v);    // row_bcast:31 -> rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// minimum over the 64 lanes, result in every lane (as a wave-uniform value): the DPP steps of wave_sum_dpp with fminf; lanes a
// step does not write keep their own value (`old` = the value itself)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_min_dpp(float v) {
  v = fminf(v, dpp_keep<0xB1, 0xF>(v));     // quad_perm [1,0,3,2]
  v = fminf(v, dpp_keep<0x4E, 0xF>(v));     // quad_perm [2,3,0,1]
  v = fminf(v, dpp_keep<0x141, 0xF>(v));    // row_half_mirror
  v = fminf(v, dpp_keep<0x140, 0xF>(v));    // row_mirror: every lane holds its 16-lane row's minimum
  v = fminf(v, dpp_keep<0x142, 0xA>(v));    // row_bcast:15 -> rows 1, 3
  v = fminf(v, dpp_keep<0x143, 0xC>(v));    // row_bcast:31 -> rows 2, 3: lane 63 holds the minimum of all
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// maximum of NON-NEGATIVE values over the 64 lanes, wave-uniform result (lanes a DPP step does not reach read 0, which is
// neutral for values >= 0; a NaN is ignored by fmaxf - the value that carries it still poisons whatever it is used in)
__device__ __forceinline__ float wave_max_nonneg_dpp(float v) {
  v = fmaxf(v, dpp_mov0<0xB1, 0xF>(v));
  v = fmaxf(v, dpp_mov0<0x4E, 0xF>(v));
  v = fmaxf(v, dpp_mov0<0x141, 0xF>(v));
  v = fmaxf(v, dpp_mov0<0x140, 0xF>(v));
  v = fmaxf(v, dpp_mov0<0x142, 0xA>(v));
  v = fmaxf(v, dpp_mov0<0x143, 0xC>(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Power-of-two normalisation ahead of an fp16 hi / lo split (the splits x = hi + lo / 2048 are exact to 22 bits only
// while x is inside fp16's normal range; gradients, caller-supplied audio and impulse responses come at any scale, and the fp32
// reference - tf.signal's FFTs, tf.GradientTape - is scale invariant).  e with m 2^-e in [1/2, 1); 0 for m = 0, inf, NaN; clamped
// so that v_ldexp_f32 by -e and by +e are both exact inverses on normal numbers.  Data is multiplied by 2^-e before the split and
// the accumulator by 2^e (x 2^e' of the other operand) after the product: powers of two, so nothing is rounded twice.
__device__ __forceinline__ int pow2_exponent(float m) {
  const int field = (int)((__builtin_bit_cast(uint32_t, m) >> 23) & 0xffu);
  int e = field - 126;
  if (m == 0.0f || field == 255) e = 0;
  return e < -125 ? -125 : (e > 126 ? 126 : e);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov0(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_mov0<0xB1, 0xF>(v);
  v += dpp_mov0<0x4E, 0xF>(v);
  v += dpp_mov0<0x141, 0xF>(v);
  v += dpp_mov0<0x140, 0xF>(v);
  v += dpp_mov0<0x142, 0xA>(v);
  v += dpp_mov0<0x143, 0xC>(v);
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Keeps a value in its register: the compiler may not rematerialise it from its parts at every use (used for LDS
// byte offsets that every unrolled iteration reuses with an immediate added; without it each use recomputes base + plane).
#if defined(__AMDGCN__)
#define DDSP_KEEP_IN_VGPR(v) __asm__ volatile("" : "+v"(v))
#define DDSP_WAIT_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)      /* s_waitcnt vmcnt(0): every vector memory operation issued so far */
#define DDSP_WAVE_LDS_SYNC() do { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); } while (0)   /* s_waitcnt lgkmcnt(0): this wavefront's LDS operations have landed (a wavefront's own LDS traffic is in order: no block barrier needed where producer and consumer are lanes of one wavefront) */
#else
#define DDSP_KEEP_IN_VGPR(v) ((void)0)
#define DDSP_WAIT_VMCNT0() ((void)0)
#define DDSP_WAVE_LDS_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// ---- individually rounded fp32 steps ---------------------------------------------------------
// hipcc contracts a*b+c into an FMA by default, and the HIP header's __fmul_rn / __fadd_rn are
// plain `*` / `+` (no OCML rounded ops in this build), so they do not stop it: round 1's 'linear'
// resample differed from TF's op order by one ulp on the MI355X because of exactly that.  These
// do: the pragma drops the `contract` flag from the instruction, and it survives inlining.
__device__ __forceinline__ float rn_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float rn_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float rn_sub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float rn_div(float a, float b) {
#pragma clang fp contract(off)
  return a / b;
}

// branch-free core.exp_sigmoid for the default constants' shape: the limits are right on both
// tails (x -> -inf: exp2 overflows to +inf, log2(inf) = inf, exp2(-inf) = 0; x -> +inf: 1).
__device__ __forceinline__ float exp_sigmoid_fast(float x, float log_exponent, float max_value,
                                                  float threshold) {
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);          // e^-x
  const float l2 = __builtin_amdgcn_logf(1.0f + e);                          // log2(1 + e^-x)
  return fmaf(max_value, __builtin_amdgcn_exp2f(-log_exponent * l2), threshold);
}

// core.exp_sigmoid (ddsp/core.py:386-404): max_value * sigmoid(x)**log(exponent) + threshold.
// sigmoid(x)**p == exp(-p * softplus(-x)); evaluated with the hardware exp2/log2
// (v_exp_f32 / v_log_f32), relative error ~1e-6, far below the parity tolerance.
__device__ __forceinline__ float exp_sigmoid(float x, float log_exponent, float max_value,
                                             float threshold) {
  const float ax = fabsf(x);
  const float sp_tail = __logf(1.0f + __expf(-ax));       // log(1 + e^-|x|)
  const float softplus_neg = (x >= 0.0f) ? sp_tail : (ax + sp_tail);  // log(1 + e^-x)
  return max_value * __expf(-log_exponent * softplus_neg) + threshold;
}

// sin(2*pi*x) for x in revolutions, |x| <= 256: one v_sin_f32.
__device__ __forceinline__ float sin_rev(float x) { return __builtin_amdgcn_sinf(x); }

// ---- Philox4x32-R (Salmon et al. SC'11); restated in oracle/ddsp_oracle.py ----------
// The on-chip noise of FilteredNoise uses R = kNoiseRounds = 10, the paper's default.  The reference draws from
// TensorFlow's stateful generator, whose stream cannot be matched from outside TF, so the generator is
// this library's own contract; round 3 measured R = 7 (the smallest count that passes BigCrush, the paper's table 2)
// in its place: 12.5 / 33.6 us against 12.6 / 33.8 us per launch at batch 32 / 128 (profiles/r03k_*) - since the
// wavefronts that make the noise are no longer what a tick waits for, three rounds less buy nothing, and the contract
// stays as it was.
constexpr int kNoiseRounds = 10;
struct U4 { uint32_t x, y, z, w; };
template <int R>
__device__ __forceinline__ U4 philox4x32(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    // one 32 x 32 -> 64 bit multiply each (v_mad_u64_u32) instead of a mul_hi and a mul_lo: integer multiplies run at
    // a quarter of the rate and are a third of what a noise sample costs
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) { return philox4x32<10>(c, k0, k1); }
__device__ __forceinline__ U4 noise_philox(U4 c, uint32_t k0, uint32_t k1) { return philox4x32<kNoiseRounds>(c, k0, k1); }
// ---- the generated noise (FilteredNoise with no noise supplied; ddsp/synths.py:192-193 draws tf.random.uniform(-1, 1)) -------
// Round 4: 2048 equally spaced levels, u = (2 k - 2047) / 2048, k an 11-bit field of a Philox word - zero mean, variance 1/3 to
// 2e-7, every value EXACTLY an fp16 number.  Rounds 1-3 made 23-bit uniforms as TensorFlow does; the stream was never TF's (its
// generator is stateful: SURVEY H6), and what those twelve extra bits cost was a third of the FIR's matrix products and half of its
// LDS traffic: the Toeplitz operand of noise_mfma65_kernel had to be carried as an fp16 hi / lo pair.  A sample whose value IS an
// fp16 number needs no lo part.  (Noise handed in by the caller - the parity entry - keeps the hi / lo pair and its 22 bits.)
// Eight samples per Philox4x32 block instead of four: sample n of a row is field (n >> 1) & 3, n & 1 of block n >> 3 - bits
// [10:0] (n even) or [26:16] (n odd) of that word.  oracle/ddsp_oracle.py device_uniform_noise restates this bit for bit.
__device__ __forceinline__ float noise_level(uint32_t k11) {          // k11 in [0, 2047]
  return (float)(2 * (int)k11 - 2047) * (1.0f / 2048.0f);             // exact
}
__device__ __forceinline__ float noise_even(uint32_t word) { return noise_level(word & 0x7FFu); }
__device__ __forceinline__ float noise_odd(uint32_t word) { return noise_level((word >> 16) & 0x7FFu); }
// the four samples 8 q + 4 h .. + 3 (h = 0, 1) of block q
__device__ __forceinline__ float4 noise_quad(const U4& r, int h) {
  const uint32_t w0 = h ? r.z : r.x, w1 = h ? r.w : r.y;
  return make_float4(noise_even(w0), noise_odd(w0), noise_even(w1), noise_odd(w1));
}
// ---- the full-resolution option (round 5: FilteredNoise(noise_bits=23), C flag DDSP_NOISE_BITS_23) ------------------------------
// tf.random.uniform(-1, 1) has 2^23 levels (ddsp/synths.py:192-193).  With 23 bits a sample is built the way TensorFlow builds
// an fp32 uniform - the top 23 bits of a word as the mantissa of a number in [1, 2), then 2 u - 3, every step exact - one sample
// per Philox word: sample n of a row is word n & 3 of block (n >> 3, row, 1 + ((n >> 2) & 1), 0).  (The third counter word tells
// the half-octets apart and keeps this stream disjoint from the 11-bit one, whose third word is 0.)  Such a sample is not an fp16
// number: the matrix-core kernels carry it as an fp16 hi / lo pair, as they carry noise the caller supplies - three products per
// k-step instead of two, twice the LDS traffic of the Toeplitz operand; bench.py reports the step both ways.
__device__ __forceinline__ float bits_to_pm1(uint32_t bits) {
  const float u = __builtin_bit_cast(float, (bits >> 9) | 0x3F800000u);
  return fmaf(u, 2.0f, -3.0f);                 // == (u - 1) 2 - 1: every step of either form is exact
}
// the four samples i .. i + 3 of global batch row `row`, i a multiple of 4
__device__ __forceinline__ float4 noise_quad_at(uint32_t i, uint64_t row, uint32_t k0, uint32_t k1, bool bits23) {
  if (bits23) {
    const U4 r = noise_philox(U4{i >> 3, (uint32_t)row, 1u + ((i >> 2) & 1u), 0u}, k0, k1);
    return make_float4(bits_to_pm1(r.x), bits_to_pm1(r.y), bits_to_pm1(r.z), bits_to_pm1(r.w));
  }
  const U4 r = noise_philox(U4{i >> 3, (uint32_t)row, 0u, 0u}, k0, k1);
  return noise_quad(r, (int)((i >> 2) & 1u));
}
// noise sample n of global batch row `row`
__device__ __forceinline__ float philox_noise(uint32_t n, uint64_t row, uint32_t k0,
                                              uint32_t k1, bool bits23 = false) {
  if (bits23) {
    const U4 r = noise_philox(U4{n >> 3, (uint32_t)row, 1u + ((n >> 2) & 1u), 0u}, k0, k1);
    const uint32_t w = n & 3u;
    return bits_to_pm1((w == 0) ? r.x : (w == 1) ? r.y : (w == 2) ? r.z : r.w);
  }
  const U4 r = noise_philox(U4{n >> 3, (uint32_t)row, 0u, 0u}, k0, k1);
  const uint32_t w = (n >> 1) & 3u;
  const uint32_t word = (w == 0) ? r.x : (w == 1) ? r.y : (w == 2) ? r.z : r.w;
  return (n & 1u) ? noise_odd(word) : noise_even(word);
}

// n / d for a divisor fixed at launch, without the divide: an integer division by a runtime value expands to ~40
// instructions (v_rcp_iflag_f32 and a correction, vector ALU even for wave-uniform operands) whose latency the
// persistent kernels paid two or three times per tick.  magic = floor(2^32 / d) (2^32 - 1 for d = 1) makes
// mulhi(n, magic) the quotient or one less for every n < 2^32; one compare puts it right.
struct FastDiv { uint32_t d, magic; };
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.magic = d <= 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / d);
  return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, FastDiv f, uint32_t& rem) {
  uint32_t q = __umulhi(n, f.magic);
  uint32_t r = n - q * f.d;
  if (r >= f.d) { ++q; r -= f.d; }
  rem = r;
  return q;
}

// A 16-byte global load that is ISSUED where it is written and whose result is first touched at load_settle():
// a plain load may be sunk by the optimiser to just before its first use (it moved four of them below a block of
// MFMAs that was there to hide their latency - harmonic_table.hip), a volatile assembly statement may not.
typedef float ddsp_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_issue(ddsp_f32x4& dst, const float4* src) {
#if defined(__AMDGCN__)
  __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src));
#else
  const float4 v = *src;
  dst = (ddsp_f32x4){v.x, v.y, v.z, v.w};
#endif
}
// every load issued so far has landed; the four values are ordered behind the wait
__device__ __forceinline__ void load_settle(ddsp_f32x4& a, ddsp_f32x4& b, ddsp_f32x4& c, ddsp_f32x4& d) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#else
  (void)a; (void)b; (void)c; (void)d;
#endif
}

// A global load of one dword / 16 bytes that is ISSUED where it is written (see load_issue) and first touched behind
// loads_landed(): the S-wavefronts of harm_wt16_kernel fetch the rows of a later chunk a whole tick before they use them.
__device__ __forceinline__ void load_issue(float& dst, const float* src) {
#if defined(__AMDGCN__)
  __asm__ volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(src));
#else
  dst = *src;
#endif
}
// The same from a wave-uniform base (scalar register pair) and a 32-bit byte offset per lane: no 64-bit address arithmetic.
__device__ __forceinline__ void load_issue(ddsp_f32x4& dst, const char* base, unsigned byte_offset) {
#if defined(__AMDGCN__)
  __asm__ volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_offset), "s"(base));
#else
  const float4 v = *reinterpret_cast<const float4*>(base + byte_offset);
  dst = (ddsp_f32x4){v.x, v.y, v.z, v.w};
#endif
}
__device__ __forceinline__ void load_issue(float& dst, const char* base, unsigned byte_offset) {
#if defined(__AMDGCN__)
  __asm__ volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_offset), "s"(base));
#else
  dst = *reinterpret_cast<const float*>(base + byte_offset);
#endif
}
// The same with five wait states in front and an immediate byte offset (0 .. 4095).  A vector instruction that writes a
// scalar register (v_readlane: how the compiler brings back a base it has spilled to a register lane; v_readfirstlane)
// must be five wait states ahead of a vector memory instruction that reads it, and the compiler's hazard recogniser does
// not look inside assembly statements: harm_table_kernel's instances for 129 .. 200 harmonics have that many bases
// (profiles/r03u_*: stale bases, wrong samples that move from run to run).  tests/test_isa_guards.py checks every pinned
// load of every instance for this.
template <int IMM>
__device__ __forceinline__ void load_issue_spaced(ddsp_f32x4& dst, const char* base, unsigned byte_offset) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(byte_offset), "s"(base), "n"(IMM));
#else
  const float4 v = *reinterpret_cast<const float4*>(base + byte_offset + IMM);
  dst = (ddsp_f32x4){v.x, v.y, v.z, v.w};
#endif
}
template <int IMM>
__device__ __forceinline__ void load_issue_spaced(float& dst, const char* base, unsigned byte_offset) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(byte_offset), "s"(base), "n"(IMM));
#else
  dst = *reinterpret_cast<const float*>(base + byte_offset + IMM);
#endif
}
// every load this wavefront has issued has landed; the listed values are ordered behind the wait
__device__ __forceinline__ void loads_landed(ddsp_f32x4& a, float& b, float& c) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
#else
  (void)a; (void)b; (void)c;
#endif
}
__device__ __forceinline__ void loads_landed(float& a) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(a));
#else
  (void)a;
#endif
}
__device__ __forceinline__ void loads_landed(float& a, float& b) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b));
#else
  (void)a; (void)b;
#endif
}
__device__ __forceinline__ void loads_landed(float& a, float& b, float& c) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
#else
  (void)a; (void)b; (void)c;
#endif
}

// v_permlane16_swap_b32 with both operands the same value: every lane receives the sum of its own 16-lane row's value
// and its partner row's (rows 0 <-> 1, 2 <-> 3).  The instruction swaps the odd rows of its first operand with the even
// rows of its second: afterwards one register holds rows (0, 0, 2, 2), the other rows (1, 1, 3, 3).
__device__ __forceinline__ float row_pair_sum(float v) {
  typedef unsigned ddsp_u32x2 __attribute__((ext_vector_type(2)));
  const unsigned bits = __builtin_bit_cast(unsigned, v);
  const ddsp_u32x2 r = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}

// v_permlane32_swap_b32 likewise for the two halves of the wavefront: every lane receives its half's value plus the
// other half's.
__device__ __forceinline__ float wave_half_sum(float v) {
  typedef unsigned ddsp_u32x2 __attribute__((ext_vector_type(2)));
  const unsigned bits = __builtin_bit_cast(unsigned, v);
  const ddsp_u32x2 r = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}

}  // namespace ddsp
