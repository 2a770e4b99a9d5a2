// Recording of C-ABI calls into an addhip_plan_t (include/addhip.h, "recorded plans"; plan.hip holds the plan itself).
//
// Between addhip_plan_record_begin and addhip_plan_record_end every stream-taking entry point called on the recording thread checks its
// arguments as usual and then, instead of launching, appends itself to the plan: the scalar arguments and device addresses by value,
// every parameter block (addhip_gemm_t, addhip_motion_t, ...) and host-side table as a COPY taken at record time.  addhip_plan_run
// replays the calls on a stream.  An entry point opts in with one line behind its argument checks:
//     ADDHIP_RECORDABLE(addhip_col_sum, X, M, N, ld, out, scale, accumulate);      (its arguments without the stream)
#pragma once
#include <functional>
#include <tuple>
#include <utility>
#include <vector>
#include "addhip.h"

namespace addhip {

// (plan.hip)
bool recording();
int record_push(const char* name, std::function<int(void*)> fn, const addhip_gemm_t* gemms, int n_gemms);
int record_size();               // calls in the plan being recorded on this thread (0 when not recording)
void record_truncate(int size);  // drop the calls recorded behind `size` (a composite entry point refused half way)

// how an argument is kept inside a recorded call: by value, except pointers to the ABI's parameter blocks, which are copied
template <class T> struct Held {
  T v;
  explicit Held(T x) : v(x) {}
  T get() const { return v; }
};
#define ADDHIP_HELD_BLOCK(TYPE)                              \
  template <> struct Held<const TYPE*> {                     \
    TYPE c;                                                  \
    bool null;                                               \
    explicit Held(const TYPE* p) : c(), null(p == nullptr) { \
      if (p) c = *p;                                         \
    }                                                        \
    const TYPE* get() const { return null ? nullptr : &c; }  \
  }
ADDHIP_HELD_BLOCK(addhip_motion_t);
ADDHIP_HELD_BLOCK(addhip_task_t);
ADDHIP_HELD_BLOCK(addhip_env_t);
ADDHIP_HELD_BLOCK(addhip_step_out_t);
ADDHIP_HELD_BLOCK(addhip_sampler_t);
ADDHIP_HELD_BLOCK(addhip_gather_t);
ADDHIP_HELD_BLOCK(addhip_rigid_model_t);
ADDHIP_HELD_BLOCK(addhip_rigid_dr_t);
ADDHIP_HELD_BLOCK(addhip_optimizer_t);
ADDHIP_HELD_BLOCK(addhip_actor_head_t);
#undef ADDHIP_HELD_BLOCK

// Q... = the entry point's parameter types (the trailing void* stream included), A... = the arguments but the stream
template <class... Q, class... A, size_t... I>
int record_impl(const char* name, int (*fn)(Q...), std::index_sequence<I...>, A... a) {
  using Params = std::tuple<Q...>;
  auto args = std::forward_as_tuple(a...);
  auto held = std::make_tuple(Held<std::tuple_element_t<I, Params>>(static_cast<std::tuple_element_t<I, Params>>(std::get<I>(args)))...);
  return record_push(name, [fn, held](void* stream) -> int { return fn(std::get<I>(held).get()..., stream); }, nullptr, 0);
}
template <class... Q, class... A>
int record_call(const char* name, int (*fn)(Q...), A... a) {
  static_assert(sizeof...(Q) == sizeof...(A) + 1, "ADDHIP_RECORDABLE lists every argument but the stream");
  return record_impl(name, fn, std::index_sequence_for<A...>{}, a...);
}

}  // namespace addhip

#define ADDHIP_RECORDABLE(fn, ...) \
  if (addhip::recording()) return addhip::record_call(#fn, &fn, __VA_ARGS__)
