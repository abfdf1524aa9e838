// harm_wt4_kernel: Harmonic.__call__ (ddsp/synths.py:94-146) on per-frame wavetables - the arithmetic of harm_table_kernel
// (harmonic_table.hip: row makers, tabulation on the matrix cores, Kaiser-Bessel interpolation; every formula below is that
// kernel's, statement for statement, so the two produce the same bits wherever they choose the same table size) - as FOUR
// INDEPENDENT BLOCKS OF FOUR WAVEFRONTS PER CU instead of one block of sixteen.  (Included by harmonic_table.hip, inside
// namespace ddsp, after that kernel's helpers and constant fragment sets.)
//
// Why (round 6).  Rounds 3-5 measured harm_table_kernel at 40 us per launch at batch 128 against 12 us of instruction issue: its
// sixteen wavefronts march in lock step - one block-wide barrier per tick, every tick as long as its slowest role, an HBM round
// trip inside the empty skeleton (15.6 of 38 us with no work in it, HISTORY.md) - and nothing else can be resident on the CU to
// fill the gaps, because the block owns its whole register file and 127 KB of its LDS.  Role ablations, priorities, slot
// assignments and prefetch distances moved it by a microsecond in three rounds.  Here the same work is cut the other way:
//   * a block is 4 wavefronts (one per SIMD) and 38 KB of LDS; four blocks are resident per CU and run UNSYNCHRONISED: while one
//     waits at its barrier, for a row from HBM or for its LDS reads, the SIMD issues for the other three;
//   * a block works on UNITS of <= 15 frames (16 amplitude rows = one MFMA N-tile) and every wavefront does a quarter of every
//     stage of a unit - two row pairs of phase A (rows straight from HBM into registers, fetched a whole unit ahead), a quarter
//     of the table positions (its share of the constant sine factor stays in 64 registers, as a tabulator's did), a quarter of
//     the tiles of phase B - so no stage waits for a specialised wavefront;
//   * two block barriers per unit (planes -> tabulation -> tables), four wavefronts each; the stages of consecutive units are
//     rotated so that a wavefront's vector-memory wait only ever sees operations issued a stage earlier:
//         wait | phase A (unit n+1) | fetch rows of unit n+2 | phase B (unit n), stores | barrier | tabulate (n+1) | barrier
//   * descriptors, table sizes and the chunk walk are wave-uniform arithmetic every wavefront does for itself (no descriptor
//     ring, no hand-off through LDS).
// The table size of a frame is a function of its SEGMENT's lowest f0, segments being kW4Segment = 60 frames at fixed positions
// of the clip (4 units of 15), so that a row run alone, in a batch of 32 or of 128 is tabulated identically (the bit-equality
// contract, tests/test_gpu_determinism.py); units never straddle a segment.
//
// K <= 128 (W = 6 up to 100 harmonics, 8 beyond); 129 .. 200 harmonics stay on harm_table_kernel's WIDE instances, whose constant
// factor does not fit a wavefront's registers.

constexpr int kW4Rows = 16;                 // amplitude rows of a unit: ONE MFMA N-tile
constexpr int kW4Frames = kW4Rows - 1;      // frames of a unit (row r + 1 is the "next" row of frame r)
constexpr int kW4Segment = 4 * kW4Frames;   // frames that share a table size
constexpr int kW4NT = 2;                    // tiles of 64 samples a wavefront carries through phase B together
#ifndef DDSP_W4_BLOCKS_PER_CU
#define DDSP_W4_BLOCKS_PER_CU 3
#endif
constexpr int kW4BlocksPerCU = DDSP_W4_BLOCKS_PER_CU;      // resident blocks per CU the register budget is set for (3: 168 registers; 4: 128)

struct W4Desc { int b, j0, nfr, fresh, seg0; };      // frames j0 .. j0 + nfr - 1 of clip b (nfr == 0: none); seg0: the segment's first frame

// a unit's per-frame tables (ChunkTables of harm_table_kernel, 16 rows) and its rows' amplitudes
struct W4Tables {
  double theta[kW4Rows], w[kW4Rows], dw[kW4Rows];
  float4 cx[kW4Rows];
  float f0[kW4Rows + 2];
  int kA[kW4Rows], kN[kW4Rows];
  float ck[kW4Rows];
  float amp[kW4Rows + 4];
  int cross;
};

// the next unit of the block's run of frames (wave-uniform arithmetic; wt_next_chunk with units of 15 and the segment's start)
__device__ __forceinline__ W4Desc w4_next_unit(WtWalk& w, int& seg0, const TableArgs& p) {
  W4Desc d{0, 0, 0, 0, 0};
  if (w.pos >= w.end) return d;
  if (w.seg_left == 0) {
    uint32_t j;
    const int b = (int)fastdiv((uint32_t)w.pos, p.f_div, j);
    d.fresh = (w.pos == w.pos_first || (int)j == 0) ? 1 : 0;
    w.b = b;
    w.j = (int)j;
    uint32_t in_seg;
    (void)fastdiv(j, p.seg_div, in_seg);
    seg0 = (int)j - (int)in_seg;
    w.seg_left = min(min(w.end - w.pos, p.F - w.j), kW4Segment - (int)in_seg);
    const int n = (w.seg_left + kW4Frames - 1) / kW4Frames;
    w.base = w.seg_left / n;
    w.rem = w.seg_left - w.base * n;
  }
  const int len = w.base + (w.rem > 0 ? 1 : 0);
  if (w.rem > 0) --w.rem;
  d.b = w.b; d.j0 = w.j; d.nfr = len; d.seg0 = seg0;
  w.pos += len; w.j += len; w.seg_left -= len;
  return d;
}

template <int W, int NK, bool ONE_TILE, bool ADD, bool ROWS16>
__global__ __launch_bounds__(256, kW4BlocksPerCU) void harm_wt4_kernel(
    const float* __restrict__ amplitudes, const float* __restrict__ hd, const float* __restrict__ f0_all,
    typename WtOut<ADD>::type audio, float* __restrict__ ctl_amp, float* __restrict__ ctl_hd, const float* add_in, TableArgs p) {
  static_assert(NK <= 2, "129 .. 200 harmonics: harm_table_kernel's WIDE instances");
  constexpr int kWtH = WtGeom<W>::H, kWtTS = WtGeom<W>::TS, kWtPS = WtPlane<NK>::PS;
  constexpr int kPlane = kW4Rows * kWtPS;                         // halves of one (part, parity) plane
  __shared__ __attribute__((aligned(16))) float tab[kW4Rows * kWtTS];
  __shared__ __attribute__((aligned(16))) _Float16 planes_all[2][4 * kPlane];      // [hi, lo][parity][row][k']
  __shared__ __attribute__((aligned(16))) W4Tables t_all[2];

  const int tid = threadIdx.x, lane = tid & 63;
  const int rw = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 3
  const int F = p.F, K = p.K;
  const int K4 = (K + 3) >> 2;
  const float kLog10 = 2.302585092994046f;                         // tf.math.log(exponent), ddsp/core.py:403
  const int mi = lane & 15, mg = lane >> 4;                        // MFMA fragment coordinates
#ifdef DDSP_W4_TIMELINE
  const bool dbg_on = p.dbg != nullptr && (int)blockIdx.x == p.rows16 >> 8 && lane == 0;      // (the block to stamp rides in rows16's upper bits)
  int dbg_unit = 0;
#define DDSP_W4_STAMP(i) do { if (dbg_on && dbg_unit < 24) p.dbg[(rw * 24 + dbg_unit) * 8 + (i)] = clock64(); } while (0)
#else
#define DDSP_W4_STAMP(i) do { } while (0)
#endif

  // ---- this wavefront's share of the constant factor (harm_table_frags.h), in MFMA A-operand layout ------------------------------
  f16x8 ahi[2][2][NK], alo[2][2][NK];
  int frag_T = 0;
  auto fetch_fragments = [&](int T) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const WtFragSet& set = W == 6 ? kWtFragSet6 : kWtFragSet8;
    unsigned l16 = 16u * (unsigned)(tid & 63);
    DDSP_KEEP_IN_VGPR(l16);
    if (T == kWtT) {
      const char* base = reinterpret_cast<const char*>(set.t512.v[rw]);      // [part][parity][tt][ks][lane][4]
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const unsigned off = (unsigned)(((par * 2 + tt) * 2 + ks) * 1024);
            ahi[par][tt][ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + off)));
            alo[par][tt][ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + (off + 8192u))));
          }
    } else if (rw < (T >> 6)) {
      const WtFragsSmall& f = T == 256 ? set.t256 : T == 128 ? set.t128 : set.t64;
      const char* base = reinterpret_cast<const char*>(f.v[rw]);              // [part][parity][lane][4]
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        ahi[par][0][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + 1024u * par)));
        alo[par][0][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + 1024u * par + 2048u)));
      }
    }
  };
  auto fragments_landed = [&]() {
#if defined(__AMDGCN__)
    __asm__ volatile("s_waitcnt vmcnt(0)");
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
          __asm__ volatile("" : "+v"(ahi[par][tt][ks]), "+v"(alo[par][tt][ks]));
#endif
  };
  fetch_fragments(kWtT);                 // what most launches start with (and the headline never leaves)
  frag_T = kWtT;

  // ---- the walk over this block's run of frames -----------------------------------------------------------------------------------
  WtWalk walk;
  walk.pos = (int)blockIdx.x * p.frames_per_block;
  walk.pos_first = walk.pos;
  walk.end = min(walk.pos + p.frames_per_block, p.total_frames);
  walk.seg_left = 0; walk.base = 0; walk.rem = 0; walk.b = 0; walk.j = 0;

  constexpr int NU = 2;                                            // row pairs per wavefront: rows 4 rw .. 4 rw + 3 of the unit
  const int u0 = NU * rw;
  struct Rows { ddsp_f32x4 x[NU]; float f0[NU]; };
  Rows rows;
  // side loads, pinned like the rows: wavefront 3's f0 (phase tables), wavefront 0's amplitudes, everybody's segment f0 (table size)
  float pf_cur = 0.0f, pf_next = 0.0f, pf_first = 0.0f, pseg = 0.0f;
  float& pamp = pf_cur;                    // (a wavefront is wavefront 0 OR wavefront 3: one register)
  const unsigned row_bytes = 4u * (unsigned)K;

  // every load a unit's phase A (and its phase tables) needs, issued together a unit ahead
  auto prefetch = [&](const W4Desc& d, bool new_segment) {
    // (everything that depends on the lane number is made HERE from an opaque copy, once per unit: hoisted out of the unit loop
    // it would live in registers this kernel does not have - 64 of its 128 hold the constant factor)
    int lane_ = tid & 63;
    DDSP_KEEP_IN_VGPR(lane_);
    const int sub = lane_ >> 5, kq = lane_ & 31;
    const unsigned kq16 = 16u * (unsigned)min(kq, K4 - 1);
    const size_t r0 = (size_t)d.b * (size_t)F;                     // (an empty descriptor: row 0 of clip 0)
    const char* hb = reinterpret_cast<const char*>(hd) + r0 * row_bytes;
    const char* fb = reinterpret_cast<const char*>(f0_all) + r0 * 4;
    unsigned ro[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const unsigned jr = (unsigned)min(d.j0 + 2 * (u0 + i) + sub, F - 1);
      ro[i] = __umul24(jr, row_bytes);
      load_issue(rows.f0[i], fb, 4u * jr);
    }
    if constexpr (ROWS16) {
#pragma unroll
      for (int i = 0; i < NU; ++i) load_issue(rows.x[i], hb, ro[i] + kq16);
    } else {
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        float e0, e1, e2, e3;
        load_issue(e0, hb, ro[i] + 4u * (unsigned)min(4 * kq + 0, K - 1));
        load_issue(e1, hb, ro[i] + 4u * (unsigned)min(4 * kq + 1, K - 1));
        load_issue(e2, hb, ro[i] + 4u * (unsigned)min(4 * kq + 2, K - 1));
        load_issue(e3, hb, ro[i] + 4u * (unsigned)min(4 * kq + 3, K - 1));
        rows.x[i] = (ddsp_f32x4){e0, e1, e2, e3};
      }
    }
    if (rw == 3) {
      load_issue(pf_cur, fb, 4u * (unsigned)min(d.j0 + lane_, F - 1));
      load_issue(pf_next, fb, 4u * (unsigned)min(d.j0 + lane_ + 1, F - 1));
      load_issue(pf_first, fb, 0u);
    }
    if (rw == 0) load_issue(pamp, reinterpret_cast<const char*>(amplitudes) + r0 * 4, 4u * (unsigned)min(d.j0 + lane_, F - 1));
    if (new_segment) load_issue(pseg, fb, 4u * (unsigned)min(d.seg0 + lane_, F - 1));
  };
  auto prefetch_landed = [&]() {
#if defined(__AMDGCN__)
    __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(rows.x[0]), "+v"(rows.x[1]), "+v"(rows.f0[0]), "+v"(rows.f0[1]), "+v"(pf_cur),
                     "+v"(pf_next), "+v"(pf_first), "+v"(pseg));
#endif
  };

  // ---- phase A of a unit: its rows -> planes (core.exp_sigmoid, remove_above_nyquist, safe_divide; harm_table_kernel's phase_a),
  //      the rows' amplitudes (wavefront 0) -----------------------------------------------------------------------------------------
  auto phase_a = [&](const W4Desc& d, _Float16* planes, W4Tables& t) {
    const int nfr = d.nfr;
    int lane_ = tid & 63;
    DDSP_KEEP_IN_VGPR(lane_);                       // (per-lane constants re-made every unit: see prefetch)
    const int sub = lane_ >> 5, kq = lane_ & 31;
    const bool live = kq < K4;
    float kf[4], nyq_u[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool alive = 4 * kq + u + 1 <= K;        // (K need not be a multiple of 4: the last lane's tail is dead)
      kf[u] = alive ? (float)(4 * kq + u + 1) : 0.0f;
      nyq_u[u] = alive ? p.nyquist : -1.0f;
    }
  const f32x2 kf_o = {kf[0], kf[2]}, kf_e = {kf[1], kf[3]};
    auto exp_sigmoid2 = [&](f32x2 v) -> f32x2 {                      // exp_sigmoid_fast on a pair
      const f32x2 t = v * -1.4426950408889634f;
      const f32x2 e = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
      const f32x2 m = (f32x2){__builtin_amdgcn_logf(e[0]), __builtin_amdgcn_logf(e[1])} * -kLog10;
      const f32x2 g = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
      return __builtin_elementwise_fma(g, (f32x2){2.0f, 2.0f}, (f32x2){1e-7f, 1e-7f});
    };
    const float kHuge = 0x1p100f;
    const f32x2 nyq_o = {nyq_u[0] * kHuge, nyq_u[2] * kHuge}, nyq_e = {nyq_u[1] * kHuge, nyq_u[3] * kHuge};
    auto nyq_mask2 = [&](f32x2 e, float f0r, f32x2 kfp, f32x2 nyq_s) -> f32x2 {
      f32x2 prod;
      { _Pragma("clang fp contract(off)") prod = kfp * f0r; }
      const f32x2 y = __builtin_elementwise_fma(prod, (f32x2){-kHuge, -kHuge}, nyq_s);
      return (f32x2){__builtin_amdgcn_fmed3f(e[0], 0.0f, y[0]), __builtin_amdgcn_fmed3f(e[1], 0.0f, y[1])};
    };

    f32x2 xo[NU], xe[NU];
    float part[NU], inv[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      xo[i] = nyq_mask2(exp_sigmoid2((f32x2){rows.x[i][0], rows.x[i][2]}), rows.f0[i], kf_o, nyq_o);
      xe[i] = nyq_mask2(exp_sigmoid2((f32x2){rows.x[i][1], rows.x[i][3]}), rows.f0[i], kf_e, nyq_e);
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) { const f32x2 h = xo[i] + xe[i]; part[i] = h[0] + h[1]; }
#pragma unroll
    for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0xB1, 0xF>(part[i]);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0x4E, 0xF>(part[i]);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0x141, 0xF>(part[i]);     // row_half_mirror
#pragma unroll
    for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0x140, 0xF>(part[i]);     // row_mirror: every lane holds its 16-lane row's sum
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      part[i] = row_pair_sum(part[i]);            // + the other 16 lanes of this matrix row
      inv[i] = __builtin_amdgcn_rcpf(fmaxf(part[i], 1e-7f));
    }
    if (ctl_hd != nullptr) {
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int arow = 2 * (u0 + i) + sub;
        const int crow = d.b * F + d.j0 + arow;
        if (arow < nfr) {
          const f32x2 ho = xo[i] * inv[i], he = xe[i] * inv[i];
          if constexpr (ROWS16) {
            if (live) reinterpret_cast<float4*>(ctl_hd)[(size_t)crow * K4 + kq] = make_float4(ho[0], he[0], ho[1], he[1]);
          } else {
            float* __restrict__ crow_p = ctl_hd + (size_t)crow * K + 4 * kq;
            const float h4[4] = {ho[0], he[0], ho[1], he[1]};
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (4 * kq + u < K) crow_p[u] = h4[u];
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const f32x2 c[2] = {xo[i] * inv[i], xe[i] * inv[i]};       // k odd (k' = 2 kq, 2 kq + 1), k even
      _Float16* dst = planes + (2 * (u0 + i) + sub) * kWtPS + 2 * kq;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const h16x2 hi = __builtin_amdgcn_cvt_pkrtz(c[par][0], c[par][1]);
        const h16x2 lo = wt_rest_halves(c[par] * kWtLoScale, hi);
        *reinterpret_cast<h16x2*>(dst + (0 * 2 + par) * kPlane) = hi;
        *reinterpret_cast<h16x2*>(dst + (1 * 2 + par) * kPlane) = lo;
      }
    }
    if (rw == 0) {                                 // the rows' amplitudes (core.exp_sigmoid, ddsp/synths.py:120-121): lanes = rows
      const float a = exp_sigmoid_fast(pamp, kLog10, 2.0f, 1e-7f);
      if (lane_ <= kW4Rows) t.amp[lane_] = a;
      if (ctl_amp != nullptr && lane_ < nfr) ctl_amp[(size_t)d.b * F + d.j0 + lane_] = a;
    }
  };

  // wavefront 3: the fp64 prefix of the clip's f0 before frame j0 (exact in fp64: the same bits in any order)
  double before = 0.0;
  auto row_prefix = [&](const W4Desc& d) -> double {
    int lane_ = tid & 63;
    DDSP_KEEP_IN_VGPR(lane_);
    const float* __restrict__ f0row = f0_all + (size_t)d.b * F;
    double part = 0.0;
    for (int j = lane_; j < d.j0; j += 256) {                     // four loads in flight per pass
      const float a = f0row[j];
      const float b = f0row[min(j + 64, d.j0 - 1)], c = f0row[min(j + 128, d.j0 - 1)], e = f0row[min(j + 192, d.j0 - 1)];
      part += (double)a;
      if (j + 64 < d.j0) part += (double)b;
      if (j + 128 < d.j0) part += (double)c;
      if (j + 192 < d.j0) part += (double)e;
    }
    return wave_sum_dpp(part);
  };
  // wavefront 3: the unit's per-frame phase tables (lanes = frames; harm_table_kernel's tabulator 3) ...
  auto phase_tables = [&](const W4Desc& d, W4Tables& t) {
    int lane_ = tid & 63, Kc = K;
    DDSP_KEEP_IN_VGPR(lane_);
#if defined(__AMDGCN__)
    __asm__ volatile("" : "+s"(Kc));
#endif
    const int nfr = d.nfr;
    const float fj = pf_cur, fj1 = pf_next;
    const double fa = (double)fj, fb = (double)fj1;
    const double mine = (lane_ < nfr) ? fa : 0.0;
    double incl = mine;                                 // inclusive scan over the unit's frames (lanes 0 .. 15)
    incl += dpp_mov0<0x111, 0xF>(incl);   // row_shr:1
    incl += dpp_mov0<0x112, 0xF>(incl);   // row_shr:2
    incl += dpp_mov0<0x114, 0xF>(incl);   // row_shr:4
    incl += dpp_mov0<0x118, 0xF>(incl);   // row_shr:8
    const double s_excl = before + (incl - mine);
    const double run = p.hop_d * s_excl + (fa - (double)pf_first) * p.half_hm1;
    const double cyc = run * p.inv_sr;
    const float fmx = fmaxf(fj, fj1), fmn = fminf(fj, fj1);
    int kA = Kc, kN = Kc;
    if (fmx > 0.0f) kA = (int)fminf((float)Kc, floorf(p.nyq_lo * __builtin_amdgcn_rcpf(fmx)));
    if (fmn > 0.0f) kN = (int)fminf((float)Kc, floorf(p.nyq_hi * __builtin_amdgcn_rcpf(fmn)));
    kA = max(min(kA, kN), 0);
    const bool crossing = lane_ < nfr && kA < kN;
    const unsigned long long any = __builtin_amdgcn_ballot_w64(crossing);
    if (lane_ == 0) t.cross = any != 0ull ? 1 : 0;
    if (lane_ <= kW4Rows) t.f0[lane_] = fj;
    if (lane_ < kW4Rows) {
      t.theta[lane_] = cyc - floor(cyc);
      t.w[lane_] = fa * p.inv_sr;
      t.dw[lane_] = (fb - fa) * p.inv_sr * p.inv_2hop;
      t.kA[lane_] = kA;
      t.kN[lane_] = kN;
    }
    // the sum over this unit's frames: lane 15 holds the inclusive sum of lanes 0 .. 15 (lanes >= nfr added 0)
    const long long bits15 = __builtin_bit_cast(long long, incl);
    const unsigned lo15 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits15 & 0xffffffffll), 15);
    const unsigned hi15 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits15 >> 32), 15);
    before += __builtin_bit_cast(double, (long long)(((unsigned long long)hi15 << 32) | lo15));
  };
  // ... and, behind the unit's first barrier (the amplitudes come from the planes the other wavefronts wrote), the FIRST crossing
  // harmonic of every frame that has one, ready-made (ChunkTables.cx / ck of harm_table_kernel); everything from LDS
  auto crossing_tables = [&](const W4Desc& d, const _Float16* planes, W4Tables& t) {
    if (__builtin_amdgcn_readfirstlane(t.cross) == 0) return;
    const int fr = (tid & 63) & (kW4Rows - 1);
    const float fj = t.f0[fr], fj1 = t.f0[fr + 1];
    const int kA = t.kA[fr], kN = t.kN[fr];
    const bool crossing = (tid & 63) < d.nfr && kA < kN;
    float4 cx = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float ck = 0.0f;
    if (crossing) {
      const int k = kA;
      const float kfl = (float)(k + 1);
      const _Float16* pl = planes + ((k & 1) * kW4Rows + fr) * kWtPS + (k >> 1);
      const float c0 = fmaf((float)pl[2 * kPlane], 1.0f / kWtLoScale, (float)pl[0]);
      const float c1 = fmaf((float)pl[2 * kPlane + kWtPS], 1.0f / kWtLoScale, (float)pl[kWtPS]);
      const float top = rn_mul(fj, kfl), bot = rn_mul(fj1, kfl);
      cx = make_float4(rn_mul(c0, t.amp[fr]), rn_mul(c1, t.amp[fr + 1]), top, rn_sub(bot, top));
      ck = kfl;
    }
    if ((tid & 63) < kW4Rows) { t.cx[fr] = cx; t.ck[fr] = ck; }
  };

  // ---- the table of a unit: O and E on the quarter range, folded into the half table (harm_table_kernel's tabulators, one row tile) --
  auto tabulate = [&](const _Float16* planes, const W4Tables& t, int Tm) {
    const _Float16* bsrc = planes + mi * kWtPS + 8 * mg;
    const float am = t.amp[mi], am_lo = am * (1.0f / kWtLoScale);      // this lane's table row's amplitude
    float* trow = tab + mi * kWtTS + kWtH;
    if (Tm != kWtT) {
      // ---- 256, 128 or 64 points: T / 64 position tiles - wavefront rw has tile rw or none -, one k-step -------------------
      if (rw < (Tm >> 6)) {
        const int half = Tm >> 1;
        f32x4 soe[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          const f16x8 bhi = *reinterpret_cast<const f16x8*>(bsrc + (0 * 2 + par) * kPlane);
          const f16x8 blo = *reinterpret_cast<const f16x8*>(bsrc + (1 * 2 + par) * kPlane);
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][0][0], bhi, zero, 0, 0, 0);
          f32x4 accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][0][0], blo, zero, 0, 0, 0);
          accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[par][0][0], bhi, accx, 0, 0, 0);
          soe[par] = acc * am + accx * am_lo;
        }
        const int n0 = 16 * rw + 4 * mg;
        const f32x4 sp = soe[0] + soe[1], sm = soe[0] - soe[1];     // S(n) = O + E, S(T/2-1-n) = O - E
        *reinterpret_cast<f32x4*>(trow + n0) = sp;
        *reinterpret_cast<f32x4*>(trow + (half - 4 - n0)) = (f32x4){sm.w, sm.z, sm.y, sm.x};
        if (n0 == 0) {                                              // halos (four entries either side: K <= 128)
          *reinterpret_cast<f32x4*>(trow - 4) = (f32x4){-sp.w, -sp.z, -sp.y, -sp.x};
          *reinterpret_cast<f32x4*>(trow + half) = (f32x4){-sm.x, -sm.y, -sm.z, -sm.w};
        }
      }
      return;
    }
    f32x4 soe[2][2];                                     // [parity][position tile]: a_j (hi.hi + (hi.lo + lo.hi) / 2048)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      f32x4 acc[2], accx[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accx[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const f16x8 bhi = *reinterpret_cast<const f16x8*>(bsrc + (0 * 2 + par) * kPlane + 32 * ks);
        const f16x8 blo = *reinterpret_cast<const f16x8*>(bsrc + (1 * 2 + par) * kPlane + 32 * ks);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][tt][ks], bhi, acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][tt][ks], blo, accx[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[par][tt][ks], bhi, accx[tt], 0, 0, 0);
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) soe[par][tt] = acc[tt] * am + accx[tt] * am_lo;
    }
    // D[row = 4 (lane >> 4) + reg][col = lane & 15]: this lane holds positions n0 .. n0+3 of table row mi
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int n0 = 16 * (2 * rw + tt) + 4 * mg;
      const f32x4 so = soe[0][tt], se = soe[1][tt];     // odd, even harmonics
      const f32x4 sp = so + se;                         // S(n)         = O + E
      const f32x4 sm = so - se;                         // S(T/2-1-n)   = O - E
      *reinterpret_cast<f32x4*>(trow + n0) = sp;
      *reinterpret_cast<f32x4*>(trow + (kWtHalf - 4 - n0)) = (f32x4){sm.w, sm.z, sm.y, sm.x};
      if (n0 == 0) {                                     // halos: S(-1-m) = -S(m), S(T/2+m) = -S(T/2-1-m)
        *reinterpret_cast<f32x4*>(trow - kWtH) = (f32x4){-sp.w, -sp.z, -sp.y, -sp.x};
        *reinterpret_cast<f32x4*>(trow + kWtHalf) = (f32x4){-sm.x, -sm.y, -sm.z, -sm.w};
      }
    }
  };

  // ---- phase B of a unit: tiles of 64 samples, lanes = samples (harm_table_kernel's interpolators, two tiles at a time) -----------
  WtPkCoefs<W> coef;
  coef.init();
  auto phase_b = [&](const W4Desc& d, const _Float16* planes, const W4Tables& t, float Tf) {
    const int nfr = d.nfr;
    const int row0 = d.b * F + d.j0;
    const int hop = p.hop;
    const float inv_hop = 1.0f / (float)hop;
    const bool unit_cross = __builtin_amdgcn_readfirstlane(t.cross) != 0;
    const int n_tiles = ONE_TILE ? nfr : nfr * ((hop + 63) >> 6);
    const bool ragged = !ONE_TILE && p.ragged != 0;
    const size_t unit0 = (size_t)row0 * (size_t)hop;
    char* out_unit = reinterpret_cast<char*>(audio + unit0);          // (add_in may be this very buffer: no __restrict__)
    const char* add_unit = ADD ? reinterpret_cast<const char*>(add_in + unit0) : nullptr;
    auto tiles = [&](int tile, auto nt_tag) {
      constexpr int NT = decltype(nt_tag)::value;
      int q[kWtNT], r[kWtNT];
      double cyc[kWtNT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int tl = tile + 4 * u;
        if (ONE_TILE) { q[u] = tl; r[u] = lane; }
        else {
          uint32_t rem;
          q[u] = (int)fastdiv((uint32_t)tl, p.tpf_div, rem);
          r[u] = (int)rem * 64 + lane;
        }
        const double rr = (double)r[u];
        // inclusive cumsum of f[t]/sr inside the frame: (r+1) w + r (r+1) dw, in revolutions
        cyc[u] = fma(rr + 1.0, fma(t.dw[q[u]], rr, t.w[q[u]]), t.theta[q[u]]);
      }
      const unsigned o32 = 4u * (unsigned)(tile * 64 + lane);          // bytes
      float addv[kWtNT];
      if (ADD) {
        if (!ragged) {
#pragma unroll
          for (int u = 0; u < NT; ++u) addv[u] = *reinterpret_cast<const float*>(add_unit + (o32 + 1024u * (unsigned)u));
        } else {
#pragma unroll
          for (int u = 0; u < NT; ++u)
            addv[u] = r[u] < hop ? *reinterpret_cast<const float*>(add_unit + 4u * (unsigned)(q[u] * hop + r[u])) : 0.0f;
        }
      }
      float theta[kWtNT];
      f32x2 zz[kWtNT];
      unsigned sgn[kWtNT];
      const float* t0[kWtNT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        theta[u] = (float)__builtin_amdgcn_fract(cyc[u]);               // v_fract_f64: [0, 1]
        const float hm = 0.5f - theta[u];                               // S(1 - theta) = -S(theta): sign bit <=> theta > 1/2
        sgn[u] = __builtin_bit_cast(unsigned, hm) & 0x80000000u;
        const float th = 0.5f - fabsf(hm);                              // [0, 0.5]
        const float pos = fmaf(th, Tf, -0.5f);                          // table coordinate, [-0.5, T / 2 - 0.5]
        const float z = __builtin_amdgcn_fractf(pos) - 0.5f;            // (v_fract_f32: pos - floor(pos), below 1)
        zz[u] = (f32x2){z, z * z};
        t0[u] = tab + q[u] * kWtTS + kWtH + wt_floor_int(pos);          // floor(pos) in [-1, T / 2 - 1]
      }
#pragma unroll
      for (int u = NT; u < kWtNT; ++u) { t0[u] = tab; zz[u] = (f32x2){0.0f, 0.0f}; }
      f32x2 acc0[kWtNT], acc1[kWtNT];
#pragma unroll
      for (int u = 0; u < kWtNT; ++u) { acc0[u] = (f32x2){0.0f, 0.0f}; acc1[u] = (f32x2){0.0f, 0.0f}; }
      wt_taps_pk<W, NT>(t0, zz, coef, acc0, acc1);
      float out[kWtNT], w_cur[kWtNT], w_next[kWtNT], lerp[kWtNT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        lerp[u] = (float)r[u] * inv_hop;
        w_next[u] = p.amp_linear ? lerp[u] : fmaf(-0.5f, __builtin_amdgcn_cosf(0.5f * lerp[u]), 0.5f);
        w_cur[u] = 1.0f - w_next[u];
        const float s0 = acc0[u][0] + acc0[u][1], s1 = acc1[u][0] + acc1[u][1];
        const float v = fmaf(w_next[u], s1, rn_mul(w_cur[u], s0));
        out[u] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ sgn[u]);
      }
      if (unit_cross)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int kA = __builtin_amdgcn_readfirstlane(t.kA[q[u]]);
        const int kN = __builtin_amdgcn_readfirstlane(t.kN[q[u]]);
        if (kA < kN) {         // harmonics crossing Nyquist inside this frame: audio-rate mask, TF's fp32 op order
          {
            const float4 cx = t.cx[q[u]];
            const float ck = t.ck[q[u]];
            const float fk = rn_add(cx.z, rn_mul(cx.w, lerp[u]));
            const float ak = fmaf(w_next[u], cx.y, rn_mul(w_cur[u], cx.x));
            const float sv = sin_rev(fmaf(theta[u], ck, -rintf(theta[u] * ck)));     // exact fractional part of k theta
            if (fk >= p.nyquist) out[u] = fmaf(-ak, sv, out[u]);
          }
          if (kA + 1 < kN) {   // more than one: the general form (an f0 that moves by several per cent within a frame)
            const float fj = t.f0[q[u]], fj1 = t.f0[q[u] + 1];
            const float am0 = t.amp[q[u]], am1 = t.amp[q[u] + 1];
            for (int k = kA + 1; k < kN; ++k) {
              const float kfl = (float)(k + 1);
              const float top = fj * kfl, bot = fj1 * kfl;
              const float fk = rn_add(top, rn_mul(rn_sub(bot, top), lerp[u]));
              const _Float16* pl = planes + ((k & 1) * kW4Rows + q[u]) * kWtPS + (k >> 1);
              const float c0 = fmaf((float)pl[2 * kPlane], 1.0f / kWtLoScale, (float)pl[0]);
              const float c1 = fmaf((float)pl[2 * kPlane + kWtPS], 1.0f / kWtLoScale, (float)pl[kWtPS]);
              const float ak = fmaf(w_next[u], rn_mul(c1, am1), rn_mul(w_cur[u], rn_mul(c0, am0)));
              const float sv = sin_rev(fmaf(theta[u], kfl, -rintf(theta[u] * kfl)));
              if (fk >= p.nyquist) out[u] = fmaf(-ak, sv, out[u]);
            }
          }
        }
      }
      if (ADD)
#pragma unroll
        for (int u = 0; u < NT; ++u) out[u] += addv[u];
      if (!ragged) {
#pragma unroll
        for (int u = 0; u < NT; ++u) *reinterpret_cast<float*>(out_unit + (o32 + 1024u * (unsigned)u)) = out[u];          // N == F * hop
      } else {
#pragma unroll
        for (int u = 0; u < NT; ++u)
          if (r[u] < hop) *reinterpret_cast<float*>(out_unit + 4u * (unsigned)(q[u] * hop + r[u])) = out[u];
      }
    };
    // tiles rw, rw + 4, rw + 8, ..: two at a time
    for (int base = rw; base < n_tiles; base += 8) {
      if (base + 4 < n_tiles) tiles(base, std::integral_constant<int, 2>{});
      else tiles(base, std::integral_constant<int, 1>{});
    }
  };

  // The table size of a unit's segment from the smallest f0 of its frames and the frame behind them (wt_table_size): every
  // wavefront for itself, from its own load
  auto segment_size = [&]() -> int {
    int lane_ = lane;
    DDSP_KEEP_IN_VGPR(lane_);
    return wt_table_size(wave_min_dpp(lane_ <= kW4Segment ? pseg : __builtin_inff()), p.size_thr);
  };

  // ---- the first unit --------------------------------------------------------------------------------------------------------------
  int seg_walk = 0;
  W4Desc dC = w4_next_unit(walk, seg_walk, p), dN{0, 0, 0, 0, 0};
  if (dC.nfr == 0) return;
  prefetch(dC, true);
  dN = w4_next_unit(walk, seg_walk, p);
  prefetch_landed();                               // (the constant fragments too)
  fragments_landed();
  int Tc = segment_size(), Tn = Tc;
  if (Tc != frag_T) { fetch_fragments(Tc); frag_T = Tc; fragments_landed(); }
  if (rw == 3) {
    if (dC.fresh) before = row_prefix(dC);
    phase_tables(dC, t_all[0]);
  }
  phase_a(dC, planes_all[0], t_all[0]);
  if (dN.nfr > 0) prefetch(dN, dN.b != dC.b || dN.seg0 != dC.seg0);
  __syncthreads();
  if (rw == 3) crossing_tables(dC, planes_all[0], t_all[0]);
  tabulate(planes_all[0], t_all[0], Tc);
  __syncthreads();
  // ---- steady state: wait | phase A (next unit) | fetch (the unit after) | phase B (this unit) | barrier | tabulate (next) | barrier
  for (int u = 0;; ++u) {
    const int cur = u & 1, nxt = cur ^ 1;
    const bool more = dN.nfr > 0;
    W4Desc dNN{0, 0, 0, 0, 0};
    DDSP_W4_STAMP(0);
    if (more) {
      const bool new_seg = dN.b != dC.b || dN.seg0 != dC.seg0;
      prefetch_landed();
      DDSP_W4_STAMP(1);
      if (new_seg) {
        Tn = segment_size();
        if (Tn != frag_T) { fetch_fragments(Tn); frag_T = Tn; fragments_landed(); }
      }
      if (rw == 3) {
        if (dN.fresh) before = row_prefix(dN);
        phase_tables(dN, t_all[nxt]);
      }
      phase_a(dN, planes_all[nxt], t_all[nxt]);
      dNN = w4_next_unit(walk, seg_walk, p);
      if (dNN.nfr > 0) prefetch(dNN, dNN.b != dN.b || dNN.seg0 != dN.seg0);
    }
    DDSP_W4_STAMP(2);
    phase_b(dC, planes_all[cur], t_all[cur], (float)Tc);
    DDSP_W4_STAMP(3);
    if (!more) break;
    __syncthreads();
    DDSP_W4_STAMP(4);
    if (rw == 3) crossing_tables(dN, planes_all[nxt], t_all[nxt]);
    tabulate(planes_all[nxt], t_all[nxt], Tn);
    DDSP_W4_STAMP(5);
    __syncthreads();
    DDSP_W4_STAMP(6);
#ifdef DDSP_W4_TIMELINE
    ++dbg_unit;
#endif
    dC = dN; dN = dNN; Tc = Tn;
  }
#if defined(__AMDGCN__)
  __asm__ volatile("s_waitcnt vmcnt(0)");
#endif
#undef DDSP_W4_STAMP
}

// the launch: four blocks of four wavefronts per CU, each with a contiguous run of frames
int launch_harm_wt4(const float* amplitudes, const float* hd, const float* f0, float* audio, float* ctl_amp, float* ctl_hd,
                    const float* add_in, TableArgs p, hipStream_t st) {
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  p.seg_div = make_fastdiv((uint32_t)kW4Segment);
  int blocks = (p.total_frames + kW4Frames - 1) / kW4Frames;
  if (blocks > kW4BlocksPerCU * n_cu) blocks = kW4BlocksPerCU * n_cu;
  p.frames_per_block = (p.total_frames + blocks - 1) / blocks;
  blocks = (p.total_frames + p.frames_per_block - 1) / p.frames_per_block;       // no empty block
  const dim3 grid((unsigned)blocks), block(256);
  const int K = p.K;
  p.dbg = nullptr;
#ifdef DDSP_W4_TIMELINE
  // DDSP_EXP_TABLE_TIMELINE=1 (a -DDDSP_W4_TIMELINE build): block DDSP_W4_DBG_BLOCK records wall-clock stamps per unit and wavefront
  static const bool timeline = getenv("DDSP_EXP_TABLE_TIMELINE") != nullptr;
  static const int dbg_block = [] { const char* e = getenv("DDSP_W4_DBG_BLOCK"); return e ? atoi(e) : 0; }();
  static long long* dbg_buf = nullptr;
  if (timeline) {
    if (!dbg_buf && hipMalloc(&dbg_buf, 4 * 24 * 8 * sizeof(long long)) != hipSuccess) dbg_buf = nullptr;
    if (dbg_buf) (void)hipMemsetAsync(dbg_buf, 0, 4 * 24 * 8 * sizeof(long long), st);
    p.dbg = dbg_buf;
    p.rows16 |= dbg_block << 8;
  }
#endif
  hipEvent_t ev0, ev1;
  profile_kernel_events(kHarmTable, &ev0, &ev1);
#define DDSP_LAUNCH_W4__(W, NK, ONE, ADD, R16)                                                                      \
  hipExtLaunchKernelGGL((harm_wt4_kernel<W, NK, ONE, ADD, R16>), grid, block, 0, st, ev0, ev1, 0, amplitudes, hd, f0, \
                        audio, ctl_amp, ctl_hd, add_in, p)
#define DDSP_LAUNCH_W4_(W, NK, ONE, ADD)                                                                        \
  do {                                                                                                         \
    if (p.rows16 & 1) DDSP_LAUNCH_W4__(W, NK, ONE, ADD, true);                                                 \
    else DDSP_LAUNCH_W4__(W, NK, ONE, ADD, false);                                                             \
  } while (0)
#define DDSP_LAUNCH_W4(W, NK)                                                                                   \
  do {                                                                                                         \
    if (p.hop == 64) {                                                                                         \
      if (add_in != nullptr) DDSP_LAUNCH_W4_(W, NK, true, true);                                               \
      else DDSP_LAUNCH_W4_(W, NK, true, false);                                                                \
    } else {                                                                                                   \
      if (add_in != nullptr) DDSP_LAUNCH_W4_(W, NK, false, true);                                              \
      else DDSP_LAUNCH_W4_(W, NK, false, false);                                                               \
    }                                                                                                          \
  } while (0)
  if (K <= 64) DDSP_LAUNCH_W4(6, 1);
  else if (K <= 100) DDSP_LAUNCH_W4(6, 2);
  else DDSP_LAUNCH_W4(8, 2);
#undef DDSP_LAUNCH_W4
#undef DDSP_LAUNCH_W4_
#undef DDSP_LAUNCH_W4__
#ifdef DDSP_W4_TIMELINE
  if (p.dbg) {
    static long long host[4 * 24 * 8];
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(host, p.dbg, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess) {
      long long t0 = host[0];
      for (int w = 0; w < 4; ++w) if (host[w * 24 * 8] && host[w * 24 * 8] < t0) t0 = host[w * 24 * 8];
      for (int w = 0; w < 4; ++w)
        for (int i = 0; i < 24 && host[(w * 24 + i) * 8] != 0; ++i) {
          const long long* r = host + (w * 24 + i) * 8;
          fprintf(stderr, "[w4] wave %d unit %2d  start %8lld  wait +%5lld  phaseA +%5lld  phaseB +%5lld  bar1 +%5lld  tabulate +%5lld  bar2 +%5lld\n",
                  w, i, r[0] - t0, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5]);
        }
    }
  }
#endif
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}
