// A C++ host without PyTorch: allocates the buffers of a small actor / critic / discriminator with hipMalloc, runs one optimiser step's loss
// sections (a) by calling the composite entry points directly on a stream and (b) from a recorded plan under the four-stream schedule with a
// bucket call-back, and compares the gradients of the two runs.  Build: hipcc -std=c++17 -I include step_host.cpp -L add-gym_amd -laddhip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "addhip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ADD(x) do { if (int rc_ = (x)) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, addhip_last_error()); return 3; } } while (0)

static std::mt19937 rng(7);
static std::vector<void*> g_allocs;
template <class T> static T* dev(size_t n, float scale = 0.f) {  // n elements: zeros, or N(0, scale) for floats
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(2); }
  g_allocs.push_back(p);
  std::vector<T> h(n, T(0));
  if (scale > 0.f) { std::normal_distribution<float> d(0.f, scale); for (auto& v : h) v = (T)d(rng); }
  (void)hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
  return (T*)p;
}

struct Net { addhip_mlp_t c; float* grads; size_t count; };
// parameters and gradients of a net in one flat buffer each (the layout a host would all-reduce)
static Net make_net(int in_dim, int in_ld, std::vector<int> hidden, int head_rows, int rows_cap, bool top_scratch) {
  Net n;
  memset(&n.c, 0, sizeof(n.c));
  size_t count = 0, off[16], k = 0;
  int prev = in_ld;
  for (int h : hidden) { off[k++] = count; count += (size_t)h * prev; off[k++] = count; count += h; prev = h; }
  off[k++] = count; count += (size_t)head_rows * prev; off[k++] = count; count += head_rows < 4 ? 4 : head_rows;
  float* P = dev<float>(count, 0.05f);
  n.grads = dev<float>(count);
  n.count = count;
  addhip_mlp_t& c = n.c;
  c.num_hidden = (int)hidden.size(); c.in_dim = in_dim; c.in_ld = in_ld; c.head_rows = head_rows; c.precision = ADDHIP_PREC_F32; c.rows_cap = rows_cap;
  size_t slab = 0;
  prev = in_ld;
  for (int i = 0; i < c.num_hidden; ++i) {
    const int h = hidden[i];
    c.hidden[i] = h;
    c.W[i] = P + off[2 * i]; c.b[i] = P + off[2 * i + 1]; c.gW[i] = n.grads + off[2 * i]; c.gb[i] = n.grads + off[2 * i + 1];
    c.h[i] = dev<float>((size_t)rows_cap * h); c.dz[i] = dev<float>((size_t)rows_cap * h); c.hbits[i] = dev<uint32_t>((size_t)rows_cap * ((h + 31) / 32));
    if ((size_t)h * prev > slab) slab = (size_t)h * prev;
    prev = h;
  }
  const int nh = c.num_hidden;
  c.Wh = P + off[2 * nh]; c.bh = P + off[2 * nh + 1]; c.gWh = n.grads + off[2 * nh]; c.gbh = n.grads + off[2 * nh + 1];
  c.slab_floats = (int64_t)(2 * 32 * slab);
  c.slabs = dev<float>((size_t)c.slab_floats);
  if (top_scratch) c.slabs_top = dev<float>((size_t)c.slab_floats);
  c.bias_replicas = dev<float>(16 * 1024); c.bias_replica_rows = 16;
  return n;
}

static int g_buckets = 0;
static void on_bucket(void* user, int32_t bucket, void* stream) {  // where a host issues ncclAllReduce(bucket's range, stream)
  (void)stream;
  static_cast<int*>(user)[g_buckets++] = bucket;
}

int main() {
  const int Mb = 2048, OBS = 264, OBS_LD = 272, DISC = 114, DISC_LD = 128;
  Net A = make_net(OBS, OBS_LD, {256, 256, 128}, 32, Mb + 1, false), C = make_net(OBS, OBS_LD, {256, 256, 128}, 1, Mb + 1, false),
      D = make_net(DISC, DISC_LD, {256, 128}, 1, Mb + 1, true);
  addhip_ppo_loss_t ppo;
  memset(&ppo, 0, sizeof(ppo));
  ppo.actor = &A.c; ppo.critic = &C.c; ppo.rows = Mb;
  ppo.norm_obs = dev<float>((size_t)Mb * OBS_LD, 1.f); ppo.norm_action = dev<float>((size_t)Mb * 32, 1.f); ppo.old_logp = dev<float>(Mb, 0.1f);
  ppo.adv = dev<float>(Mb, 1.f); ppo.tar_val = dev<float>(Mb, 1.f);
  { std::vector<float> ones(Mb, 1.f); float* m = dev<float>(Mb); (void)hipMemcpy(m, ones.data(), Mb * 4, hipMemcpyHostToDevice); ppo.rand_mask = m; }
  ppo.action_std = 0.05f; ppo.logp_const = 60.2f; ppo.ppo_clip_ratio = 0.2f; ppo.action_bound_weight = 10.f; ppo.critic_loss_weight = 1.f; ppo.grad_scale = 1.f;
  ppo.head_precision = ADDHIP_PREC_F32;
  ppo.mean = dev<float>((size_t)(Mb + 1) * 32); ppo.d_mean = dev<float>((size_t)(Mb + 1) * 32); ppo.dv = dev<float>(Mb); ppo.num_valid = dev<float>(4);
  ppo.stats = dev<float>(32);
  addhip_disc_loss_t dl;
  memset(&dl, 0, sizeof(dl));
  dl.disc = &D.c; dl.rows = Mb; dl.disc_dim = DISC;
  { float* nd = dev<float>((size_t)(Mb + 1) * DISC_LD, 1.f); (void)hipMemset(nd + (size_t)Mb * DISC_LD, 0, DISC_LD * 4); dl.norm_diff = nd; }  // row Mb: the zero difference
  dl.loss_scale = 5.f; dl.logit_reg = 0.01f; dl.grad_penalty = 2.f; dl.weight_decay = 1e-4f;
  dl.dlogit = dev<float>(Mb + 1); dl.a[1] = dev<float>((size_t)Mb * 128); dl.a[0] = dev<float>((size_t)Mb * 256); dl.g = dev<float>((size_t)Mb * DISC_LD);
  dl.G = dev<float>((size_t)Mb * DISC_LD); dl.e[0] = dev<float>((size_t)Mb * 256); dl.e[1] = dev<float>((size_t)Mb * 128); dl.stats = ppo.stats;

  hipStream_t st[4];
  for (auto& s : st) HIP(hipStreamCreate(&s));
  Net* nets[3] = {&A, &C, &D};
  std::vector<std::vector<float>> g_direct(3), g_plan(3);
  // (a) directly, on one stream
  for (Net* n : nets) ADD(addhip_fill_zero(n->grads, (int64_t)n->count, st[0]));
  ADD(addhip_ppo_loss_fwd_bwd(&ppo, nullptr, st[0]));
  ADD(addhip_disc_loss_fwd_bwd(&dl, nullptr, st[0]));
  HIP(hipStreamSynchronize(st[0]));
  for (int i = 0; i < 3; ++i) { g_direct[i].resize(nets[i]->count); HIP(hipMemcpy(g_direct[i].data(), nets[i]->grads, nets[i]->count * 4, hipMemcpyDeviceToHost)); }
  // (b) recorded, then replayed under the four-stream schedule
  addhip_plan_t* plan = nullptr;
  addhip_ppo_marks_t pm;
  addhip_disc_marks_t dm;
  ADD(addhip_plan_create(&plan));
  ADD(addhip_plan_record_begin(plan));
  int rc = addhip_ppo_loss_fwd_bwd(&ppo, &pm, nullptr);
  if (!rc) rc = addhip_disc_loss_fwd_bwd(&dl, &dm, nullptr);
  ADD(addhip_plan_record_end(plan));
  ADD(rc);
  addhip_section_t sec[10];
  if (addhip_update_schedule(0, &pm, &dm, sec, 10) != 10) return 4;
  addhip_schedule_t* step = nullptr;
  ADD(addhip_schedule_create(plan, sec, 10, 4, &step));
  int seen[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  void* streams[4] = {st[0], st[1], st[2], st[3]};
  for (int rep = 0; rep < 2; ++rep) {
    g_buckets = 0;
    HIP(hipMemsetAsync(ppo.stats, 0, 32 * 4, st[0]));
    for (Net* n : nets) ADD(addhip_fill_zero(n->grads, (int64_t)n->count, st[0]));
    ADD(addhip_schedule_run(step, streams, on_bucket, seen));
    HIP(hipStreamSynchronize(st[0]));
  }
  double worst = 0.0, norm = 0.0;
  for (int i = 0; i < 3; ++i) {
    g_plan[i].resize(nets[i]->count);
    HIP(hipMemcpy(g_plan[i].data(), nets[i]->grads, nets[i]->count * 4, hipMemcpyDeviceToHost));
    double mx = 0.0;
    for (float v : g_direct[i]) { if (!std::isfinite(v)) return 5; mx = std::fmax(mx, std::fabs(v)); }
    norm += mx;
    for (size_t k = 0; k < g_plan[i].size(); ++k) worst = std::fmax(worst, std::fabs(g_plan[i][k] - g_direct[i][k]) / mx);
  }
  printf("launches=%d buckets=%d order=%d,%d,%d,%d grad_scale=%.3e worst_rel_diff=%.3e\n", addhip_plan_size(plan), g_buckets, seen[0], seen[1], seen[2], seen[3], norm, worst);
  ADD(addhip_schedule_destroy(step));
  ADD(addhip_plan_destroy(plan));
  for (void* p : g_allocs) (void)hipFree(p);
  return worst < 1e-4 && norm > 0 ? 0 : 6;
}
