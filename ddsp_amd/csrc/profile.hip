#include "profile.h"
#include "../../include/ddsp_amd.h"
#include <vector>
#include <mutex>

namespace ddsp {
namespace {
struct Rec { int id; hipEvent_t e0, e1; };
std::mutex g_mu;
bool g_on = false;
unsigned g_mask = 0;
size_t g_cap = 0;
unsigned g_stride = 1;                 // bracket every g_stride-th launch of a selected kernel
unsigned g_seen[kNumKernels] = {};
std::vector<Rec> g_recs;
const char* const kNames[kNumKernels] = {
    "harm_controls_kernel", "harm_synth_kernel", "noise_controls_kernel", "noise_ir_kernel",
    "tv_fir_kernel", "uniform_noise_kernel", "add_kernel", "exp_sigmoid_kernel",
    "harm_fused_kernel", "noise_fused65_kernel", "rv_fft_kernel", "rv_mac_kernel",
    "rv_ifft_kernel", "stft_l1_kernel", "harm_bwd_pq_kernel", "harm_bwd_chain_kernel",
    "noise_bwd_taps_kernel", "noise_bwd_mags_kernel", "stft_l1_bwd_kernel", "harm_table_kernel",
    "noise_mfma65_kernel", "harm_bwd_table_kernel", "noise_bwd_mfma_kernel", "tv_fir_mfma_kernel", "noise_ir_gemm_kernel"};
static_assert(kNumKernels <= 32, "the selection mask is 32 bits");
}  // namespace

void profile_record(int kernel_id, hipStream_t st, bool start) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on || !((g_mask >> kernel_id) & 1u)) return;
  if (start) {
    if (g_recs.size() >= g_cap) return;
    if (g_seen[kernel_id]++ % g_stride != 0) return;
    Rec r; r.id = kernel_id;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    g_recs.push_back(r);
  } else {
    if ((g_seen[kernel_id] - 1) % g_stride != 0) return;          // its start was not recorded
    for (size_t i = g_recs.size(); i-- > 0;) {
      if (g_recs[i].id == kernel_id) { (void)hipEventRecord(g_recs[i].e1, st); break; }
    }
  }
}
void profile_kernel_events(int kernel_id, hipEvent_t* start, hipEvent_t* stop) {
  *start = nullptr; *stop = nullptr;
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on || !((g_mask >> kernel_id) & 1u) || g_recs.size() >= g_cap) return;
  if (g_seen[kernel_id]++ % g_stride != 0) return;
  Rec r; r.id = kernel_id;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  g_recs.push_back(r);
  *start = r.e0; *stop = r.e1;
}
}  // namespace ddsp

using namespace ddsp;

extern "C" int ddsp_profile_kernel_count(void) { return kNumKernels; }
extern "C" const char* ddsp_profile_kernel_name(int id) {
  return (id >= 0 && id < kNumKernels) ? kNames[id] : "";
}

extern "C" int ddsp_profile_begin(unsigned kernel_mask, int max_records) {
  return ddsp_profile_begin_sampled(kernel_mask, max_records, 1);
}

extern "C" int ddsp_profile_begin_sampled(unsigned kernel_mask, int max_records, int stride) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_on) return DDSP_ERR_UNSUPPORTED;
  if (stride < 1) return DDSP_ERR_BAD_SHAPE;
  g_stride = (unsigned)stride;
  for (int i = 0; i < kNumKernels; ++i) g_seen[i] = 0;
  g_recs.clear();
  g_cap = max_records > 0 ? (size_t)max_records : 0;
  g_recs.reserve(g_cap);
  g_mask = kernel_mask;
  g_on = true;
  return DDSP_OK;
}

extern "C" int ddsp_profile_end(double* total_ms, int* counts) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on) return DDSP_ERR_UNSUPPORTED;
  g_on = false;
  for (int i = 0; i < kNumKernels; ++i) { if (total_ms) total_ms[i] = 0.0; if (counts) counts[i] = 0; }
  int rc = DDSP_OK;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      if (total_ms) total_ms[r.id] += ms;
      if (counts) counts[r.id] += 1;
    } else {
      rc = DDSP_ERR_LAUNCH;
    }
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_recs.clear();
  return rc;
}
