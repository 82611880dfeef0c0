#include "prof.h"

#include <vector>

#include "isi_common.h"

namespace isi {
namespace prof {

namespace {
struct Record {
  int kernel;
  double flops, bytes;
  hipEvent_t start, stop;
};
struct State {
  bool on = false;
  std::vector<Record> recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;  // reused across enable() calls
  size_t used = 0;
};
thread_local State g;
constexpr size_t kMaxRecords = 1 << 18;
}  // namespace

const char *kernel_name(int id) {
  switch (id) {
    case K_CONV_128x128: return "conv_igemm_f32_kernel<128,128,2,2,*>";
    case K_CONV_128x64: return "conv_igemm_f32_kernel<128,64,2,2,*>";
    case K_CONV_128x32: return "conv_igemm_f32_kernel<128,32,4,1,*>";
    case K_CONV_GATHER: return "conv_igemm_f32_kernel<*,2> (gather)";
    case K_VQ_NEAREST: return "vq_conv1x1_nearest_kernel / vq_nearest_kernel";
    case K_RESBLOCK: return "resblock_pair_kernel + resblock_f32_kernel";
    case K_CONVT_SMALL: return "convT_k4s2_small_kernel";
    case K_REL_ATTENTION: return "rel_attention_f32_kernel";
    case K_CONV_BF16X3: return "conv_igemm_f32_kernel<..,bf16x3>";
    case K_REL_ATTENTION_BWD: return "rel_attention_bwd_{kv,q}_kernel";
    case K_CONV_BF16X6: return "conv_igemm_f32_kernel<..,bf16x6>";
    case K_CONV_F16X3: return "conv_pair_kernel<..> + convT_pair_kernel<..> + conv_igemm_f32_kernel<..,f16x3>";
    case K_CONV_PAIR_128_PAIROUT: return "conv_pair_kernel<128, true, 0> (f16x3)";
    default: return "?";
  }
}

bool enabled() { return g.on; }

Scope::Scope(int kernel_id, double flops, double bytes, hipStream_t s) : slot(-1), stream(s) {
  if (!g.on || g.recs.size() >= kMaxRecords) return;
  if (g.used == g.pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g.pool.emplace_back(a, b);
  }
  auto ev = g.pool[g.used++];
  g.recs.push_back(Record{kernel_id, flops, bytes, ev.first, ev.second});
  slot = (int)g.recs.size() - 1;
}

hipEvent_t Scope::start() const { return g.recs[slot].start; }
hipEvent_t Scope::stop() const { return g.recs[slot].stop; }

int enable(int on) {   // 0: pause, 1: start (records cleared), 2: resume
  if (on == 1) {
    g.recs.clear();
    g.used = 0;
  }
  g.on = on != 0;
  return ISI_OK;
}

int read(int kernel_id, long long *launches, double *ms, double *flops, double *bytes) {
  if (kernel_id < 0 || kernel_id >= K_COUNT || !launches || !ms || !flops || !bytes)
    return invalid("prof_read: bad argument");
  long long n = 0;
  double t = 0, f = 0, b = 0;
  for (auto &r : g.recs) {
    if (r.kernel != kernel_id) continue;
    if (hipEventSynchronize(r.stop) != hipSuccess) return check_launch("hipEventSynchronize");
    float dt = 0.f;
    if (hipEventElapsedTime(&dt, r.start, r.stop) != hipSuccess) return check_launch("hipEventElapsedTime");
    ++n; t += dt; f += r.flops; b += r.bytes;
  }
  *launches = n; *ms = t; *flops = f; *bytes = b;
  return ISI_OK;
}

}  // namespace prof
}  // namespace isi
