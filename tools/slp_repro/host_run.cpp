// Runs add-gym_amd/csrc/rigid.hip -- the source text itself, through shim/hip/hip_runtime.h -- on the CPU, one 64-thread workgroup at a
// time, so that the sanitizers can look at it (is the -O2/-O3 + SLP wrong answer undefined behaviour in the kernel?).
//   host_run <in.bin> <out.bin> <kernel: 4 = four lanes (LDS form), 5 = four lanes (register form), 1 = one lane>
// in.bin / out.bin are written / read by run_host_check.py.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <vector>
#if defined(__has_feature)
#if __has_feature(memory_sanitizer)
#include <sanitizer/msan_interface.h>
#define POISON(p, n) __msan_poison(p, n)
#endif
#endif
#ifndef POISON
#define POISON(p, n) ((void)0)
#endif

namespace hostsim {
thread_local Tid tid, bid, bdim, gdim;
pthread_barrier_t wg_barrier;
pthread_barrier_t quad_barrier[16];
float mailbox[64];
unsigned mailbox_u[64];
}  // namespace hostsim
#include <functional>
namespace addhip {
void set_error(const char*, ...) {}
bool recording() { return false; }  // (record.h: nothing records here)
}  // namespace addhip
namespace {
float lds[40000];  // one workgroup's LDS (the largest form needs 34 048 floats)
}

#include "addhip.h"
namespace addhip {
int record_push(const char*, std::function<int(void*)>, const addhip_gemm_t*, int) { return 0; }
}  // namespace addhip
#include "../../add-gym_amd/csrc/rigid.hip"

struct Launch {
  int kernel, block, blocks;
  addhip_rigid_model_t M;
  float *pose, *vel;
  const float* tgt;
  int n;
  unsigned char* flag;
  unsigned* bits;
};
static Launch g;

static void* lane_main(void* arg) {
  const int lane = (int)(intptr_t)arg;
  hostsim::tid = {(unsigned)lane, 0, 0};
  hostsim::bid = {(unsigned)g.block, 0, 0};
  hostsim::bdim = {64, 1, 1};
  hostsim::gdim = {(unsigned)g.blocks, 1, 1};
  if (g.kernel == 4) rigid_step4_kernel<false>(g.M, g.pose, g.vel, g.tgt, 32, g.n, g.flag, g.bits);
  else if (g.kernel == 5) rigid_step4_kernel<true>(g.M, g.pose, g.vel, g.tgt, 32, g.n, g.flag, g.bits);
  else rigid_step_kernel(g.M, g.pose, g.vel, g.tgt, 32, g.n, g.flag, g.bits);
  return nullptr;
}

template <typename T>
static std::vector<T> rd(FILE* f, size_t n) {
  std::vector<T> v(n);
  if (fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  const auto hdr = rd<int32_t>(f, 4);  // n, num_bodies, num_points, substeps
  const int n = hdr[0], nb = hdr[1], np = hdr[2];
  const auto sc = rd<float>(f, 9);     // dt gravity kc cd friction veps limit_stiffness max_torque limit_margin
  const auto mask = rd<uint32_t>(f, 1);
  const auto body = rd<float>(f, (size_t)nb * 32);
  const auto topo = rd<int32_t>(f, (size_t)nb * 8);
  const auto pts = rd<float>(f, (size_t)(np > 0 ? np : 1) * 4);
  const auto chains = rd<int32_t>(f, 64);
  auto pose = rd<float>(f, (size_t)n * 36);
  auto vel = rd<float>(f, (size_t)n * 36);
  const auto tgt = rd<float>(f, (size_t)n * 32);
  fclose(f);
  std::vector<unsigned char> flag(n);
  std::vector<unsigned> bits(n);
  g.kernel = atoi(argv[3]);
  g.M = addhip_rigid_model_t{nb, np, body.data(), topo.data(), pts.data(), sc[0], hdr[3], sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[7], sc[8], mask[0], nullptr,
                             g.kernel == 1 ? nullptr : chains.data()};
  g.pose = pose.data(); g.vel = vel.data(); g.tgt = tgt.data(); g.n = n; g.flag = flag.data(); g.bits = bits.data();
  const int per = g.kernel == 1 ? 64 : 16;
  g.blocks = (n + per - 1) / per;
  pthread_barrier_init(&hostsim::wg_barrier, nullptr, 64);
  for (auto& b : hostsim::quad_barrier) pthread_barrier_init(&b, nullptr, 4);
  for (g.block = 0; g.block < g.blocks; ++g.block) {
    POISON(lds, sizeof(lds));  // a fresh workgroup's LDS holds nothing
    pthread_t th[64];
    for (int l = 0; l < 64; ++l) pthread_create(&th[l], nullptr, lane_main, (void*)(intptr_t)l);
    for (int l = 0; l < 64; ++l) pthread_join(th[l], nullptr);
  }
  f = fopen(argv[2], "wb");
  fwrite(pose.data(), 4, pose.size(), f);
  fwrite(vel.data(), 4, vel.size(), f);
  fwrite(bits.data(), 4, bits.size(), f);
  fclose(f);
  return 0;
}
