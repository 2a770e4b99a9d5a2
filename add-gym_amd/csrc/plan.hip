// Recorded plans and multi-stream schedules (include/addhip.h, "recorded plans"): the host-side runtime that replays the update step --
// what the reference does in Python call by call inside PPOAgent._update_model / ADDAgent._compute_loss (learning/ppo_agent.py:171-275,
// learning/add/add_agent.py:141-202) -- as ONE C call per section, for any host language.  No kernels here: a plan is a list of
// closures over the other entry points (record.h), a schedule is a list of plan ranges on streams joined by HIP events.
#include <string>
#include "common.h"
#include "record.h"

struct addhip_plan {
  struct Call {
    const char* name;  // the entry point's name (string literal)
    std::function<int(void*)> fn;
    std::vector<addhip_gemm_t> gemms;  // GEMM launches only: their problem descriptors, for introspection (roofline accounting)
  };
  std::vector<Call> calls;
  int schedules = 0;  // schedules created over this plan and not yet destroyed: addhip_plan_destroy refuses while any is alive
};

struct addhip_schedule {
  addhip_plan* plan;
  std::vector<addhip_section_t> sections;
  int num_streams;
  hipEvent_t fork;
  std::vector<hipEvent_t> done;       // one per section that another section waits for (else nullptr)
  std::vector<hipEvent_t> join;       // one per side stream
};

namespace {
thread_local addhip_plan* g_recording = nullptr;
}

namespace addhip {
bool recording() { return g_recording != nullptr; }
int record_push(const char* name, std::function<int(void*)> fn, const addhip_gemm_t* gemms, int n_gemms) {
  addhip_plan::Call c{name, std::move(fn), {}};
  if (gemms && n_gemms > 0) c.gemms.assign(gemms, gemms + n_gemms);
  g_recording->calls.push_back(std::move(c));
  return 0;
}
// composite entry points (learner.hip): a call refused half way through its launches leaves the plan as it found it
int record_size() { return g_recording ? (int)g_recording->calls.size() : 0; }
void record_truncate(int size) {
  if (g_recording && size >= 0 && (size_t)size < g_recording->calls.size()) g_recording->calls.resize((size_t)size);
}
}  // namespace addhip

extern "C" int addhip_plan_create(addhip_plan_t** out) {
  ADDHIP_REQUIRE(out, "plan_create: null argument");
  *out = new addhip_plan();
  return 0;
}

extern "C" int addhip_plan_destroy(addhip_plan_t* plan) {
  ADDHIP_REQUIRE(!plan || plan->schedules == 0, "plan_destroy: %d schedule(s) still refer to this plan (destroy them first)", plan ? plan->schedules : 0);
  if (plan && plan == g_recording) g_recording = nullptr;
  delete plan;
  return 0;
}

extern "C" int addhip_plan_record_begin(addhip_plan_t* plan) {
  ADDHIP_REQUIRE(plan, "plan_record_begin: null plan");
  ADDHIP_REQUIRE(g_recording == nullptr, "plan_record_begin: this thread is already recording a plan");
  g_recording = plan;
  return 0;
}

extern "C" int addhip_plan_record_end(addhip_plan_t* plan) {
  ADDHIP_REQUIRE(plan && g_recording == plan, "plan_record_end: this plan is not being recorded on this thread");
  g_recording = nullptr;
  return 0;
}

extern "C" int addhip_plan_size(const addhip_plan_t* plan) {
  ADDHIP_REQUIRE(plan, "plan_size: null plan");
  return (int)plan->calls.size();
}

extern "C" const char* addhip_plan_call_name(const addhip_plan_t* plan, int32_t index) {
  if (!plan || index < 0 || index >= (int32_t)plan->calls.size()) return nullptr;
  return plan->calls[index].name;
}

extern "C" int addhip_plan_call_gemms(const addhip_plan_t* plan, int32_t index, addhip_gemm_t* out, int32_t capacity) {
  ADDHIP_REQUIRE(plan && index >= 0 && index < (int32_t)plan->calls.size(), "plan_call_gemms: bad index");
  const auto& g = plan->calls[index].gemms;
  ADDHIP_REQUIRE(g.empty() || (out && capacity >= (int32_t)g.size()), "plan_call_gemms: output holds %d descriptors, the call has %d", capacity, (int)g.size());
  for (size_t i = 0; i < g.size(); ++i) out[i] = g[i];
  return (int)g.size();
}

extern "C" int addhip_plan_run(const addhip_plan_t* plan, int32_t first, int32_t last, void* stream) {
  ADDHIP_REQUIRE(plan, "plan_run: null plan");
  const int32_t n = (int32_t)plan->calls.size();
  if (last < 0) last = n;
  ADDHIP_REQUIRE(first >= 0 && first <= last && last <= n, "plan_run: range [%d, %d) outside the plan's %d calls", first, last, n);
  ADDHIP_REQUIRE(g_recording == nullptr, "plan_run: called while recording (a plan cannot be recorded into a plan)");
  for (int32_t i = first; i < last; ++i)
    if (int rc = plan->calls[i].fn(stream)) return rc;  // (the failing entry point has set the error text)
  return 0;
}

// ------------------------------------------------------------------ schedules
extern "C" int addhip_schedule_create(const addhip_plan_t* plan, const addhip_section_t* sections, int32_t count, int32_t num_streams,
                                      addhip_schedule_t** out) {
  ADDHIP_REQUIRE(plan && sections && out && count > 0 && num_streams >= 1 && num_streams <= ADDHIP_MAX_STREAMS, "schedule_create: bad arguments");
  const int32_t n = (int32_t)plan->calls.size();
  std::vector<char> waited(count, 0);
  for (int32_t k = 0; k < count; ++k) {
    const addhip_section_t& s = sections[k];
    ADDHIP_REQUIRE(s.stream >= 0 && s.stream < num_streams, "schedule_create: section %d names stream %d of %d", k, s.stream, num_streams);
    ADDHIP_REQUIRE(s.first >= 0 && s.first <= s.last && s.last <= n, "schedule_create: section %d covers [%d, %d) of %d calls", k, s.first, s.last, n);
    // a section may only wait for sections issued before it: the schedule is issued in list order, so it cannot deadlock
    ADDHIP_REQUIRE(s.wait_before < k && s.wait_after < k, "schedule_create: section %d waits for a later section", k);
    if (s.wait_before >= 0) waited[s.wait_before] = 1;
    if (s.wait_after >= 0) waited[s.wait_after] = 1;
  }
  auto* sc = new addhip_schedule();
  sc->plan = const_cast<addhip_plan*>(plan);
  sc->plan->schedules += 1;
  sc->sections.assign(sections, sections + count);
  sc->num_streams = num_streams;
  sc->fork = nullptr;
  sc->done.assign(count, nullptr);
  sc->join.assign(num_streams > 1 ? num_streams - 1 : 0, nullptr);
  bool ok = hipEventCreateWithFlags(&sc->fork, hipEventDisableTiming) == hipSuccess;
  for (int32_t k = 0; ok && k < count; ++k)
    if (waited[k]) ok = hipEventCreateWithFlags(&sc->done[k], hipEventDisableTiming) == hipSuccess;
  for (size_t i = 0; ok && i < sc->join.size(); ++i) ok = hipEventCreateWithFlags(&sc->join[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    addhip::set_error("schedule_create: hipEventCreateWithFlags failed: %s", hipGetErrorString(hipGetLastError()));
    addhip_schedule_destroy(sc);
    return -2;
  }
  *out = sc;
  return 0;
}

extern "C" int addhip_schedule_destroy(addhip_schedule_t* sc) {
  if (!sc) return 0;
  if (sc->plan) sc->plan->schedules -= 1;
  if (sc->fork) (void)hipEventDestroy(sc->fork);
  for (hipEvent_t e : sc->done)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : sc->join)
    if (e) (void)hipEventDestroy(e);
  delete sc;
  return 0;
}

extern "C" int addhip_schedule_run(addhip_schedule_t* sc, void* const* streams, addhip_bucket_fn on_bucket, void* user) {
  ADDHIP_REQUIRE(sc && streams, "schedule_run: null argument");
  ADDHIP_REQUIRE(g_recording == nullptr, "schedule_run: called while recording");
  auto st = [&](int i) { return (hipStream_t)streams[i]; };
  // fork: the side streams start behind everything already enqueued on streams[0]
  ADDHIP_HIP(hipEventRecord(sc->fork, st(0)));
  for (int i = 1; i < sc->num_streams; ++i) ADDHIP_HIP(hipStreamWaitEvent(st(i), sc->fork, 0));
  int rc = 0;
  for (size_t k = 0; k < sc->sections.size() && rc == 0; ++k) {
    const addhip_section_t& s = sc->sections[k];
    if (s.wait_before >= 0) ADDHIP_HIP(hipStreamWaitEvent(st(s.stream), sc->done[s.wait_before], 0));
    rc = addhip_plan_run(sc->plan, s.first, s.last, streams[s.stream]);
    if (rc) break;  // (the join below still runs: the side streams were forked and may hold queued work)
    if (s.wait_after >= 0) ADDHIP_HIP(hipStreamWaitEvent(st(s.stream), sc->done[s.wait_after], 0));
    if (sc->done[k]) ADDHIP_HIP(hipEventRecord(sc->done[k], st(s.stream)));
    // the exchange step of this section's result (a gradient bucket's all-reduce) is the host's: it is told where in the issue order
    // and on which stream the data is final
    if (s.bucket >= 0 && on_bucket) on_bucket(user, s.bucket, streams[s.stream]);
  }
  // join: streams[0] continues behind every side stream -- also when a section failed, so that the caller's next work on streams[0]
  // cannot race with what the side streams already hold
  for (int i = 1; i < sc->num_streams; ++i) {
    ADDHIP_HIP(hipEventRecord(sc->join[i - 1], st(i)));
    ADDHIP_HIP(hipStreamWaitEvent(st(0), sc->join[i - 1], 0));
  }
  return rc;
}
