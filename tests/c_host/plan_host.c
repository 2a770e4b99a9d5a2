/* A host in plain C (C99): records one optimiser step of the ADD agent into an addhip_plan_t through the composite entry points and lays the
 * four-stream schedule over it -- the steps of INTEGRATION.md section 2b, without Python and without a GPU (recording launches nothing; the
 * device addresses below are placeholders of the right alignment).  Prints what it recorded; tests/test_c_host.py checks the figures. */
#include <stdio.h>
#include <string.h>
#include "addhip.h"

#define DEV(n) ((void*)(uintptr_t)(0x100000000ull + (uint64_t)(n) * 0x1000000ull))   /* 16 MB apart, 16-byte aligned */

static void mlp(addhip_mlp_t* m, int in_dim, int in_ld, int nh, const int* hidden, int head_rows, int rows_cap, int base, int with_top_scratch) {
  memset(m, 0, sizeof(*m));
  m->num_hidden = nh; m->in_dim = in_dim; m->in_ld = in_ld; m->head_rows = head_rows; m->precision = ADDHIP_PREC_F32; m->rows_cap = rows_cap;
  for (int i = 0; i < nh; ++i) {
    m->hidden[i] = hidden[i];
    m->W[i] = DEV(base + 10 * i); m->b[i] = DEV(base + 10 * i + 1); m->gW[i] = DEV(base + 10 * i + 2); m->gb[i] = DEV(base + 10 * i + 3);
    m->h[i] = DEV(base + 10 * i + 4); m->dz[i] = DEV(base + 10 * i + 5); m->hbits[i] = DEV(base + 10 * i + 6);
  }
  m->Wh = DEV(base + 50); m->bh = DEV(base + 51); m->gWh = DEV(base + 52); m->gbh = DEV(base + 53);
  m->slabs = DEV(base + 54); m->slab_floats = 64ll << 20;
  if (with_top_scratch) m->slabs_top = DEV(base + 56);
  m->bias_replicas = DEV(base + 58); m->bias_replica_rows = 16;
}

int main(void) {
  const int Mb = 16384, h3[3] = {1024, 1024, 512}, h2[2] = {1024, 512};
  addhip_mlp_t actor, critic, disc;
  mlp(&actor, 264, 272, 3, h3, 32, Mb + 1, 100, 0);
  mlp(&critic, 264, 272, 3, h3, 1, Mb + 1, 200, 0);
  mlp(&disc, 114, 128, 2, h2, 1, Mb + 1, 300, 1);
  addhip_ppo_loss_t ppo;
  memset(&ppo, 0, sizeof(ppo));
  ppo.actor = &actor; ppo.critic = &critic; ppo.rows = Mb;
  ppo.norm_obs = DEV(1); ppo.norm_action = DEV(2); ppo.old_logp = DEV(3); ppo.adv = DEV(4); ppo.tar_val = DEV(5); ppo.rand_mask = DEV(6);
  ppo.action_std = 0.05f; ppo.logp_const = 60.2f; ppo.ppo_clip_ratio = 0.2f; ppo.action_bound_weight = 10.f; ppo.critic_loss_weight = 1.f; ppo.grad_scale = 1.f;
  ppo.head_precision = ADDHIP_PREC_F32;
  ppo.mean = DEV(7); ppo.d_mean = DEV(8); ppo.dv = DEV(9); ppo.num_valid = DEV(10); ppo.stats = DEV(11);
  addhip_disc_loss_t dl;
  memset(&dl, 0, sizeof(dl));
  dl.disc = &disc; dl.rows = Mb; dl.disc_dim = 114; dl.norm_diff = DEV(12); dl.loss_scale = 5.f; dl.logit_reg = 0.01f; dl.grad_penalty = 2.f; dl.weight_decay = 1e-4f;
  dl.dlogit = DEV(13); dl.a[1] = DEV(14); dl.a[0] = DEV(15); dl.g = DEV(16); dl.G = DEV(17); dl.e[0] = DEV(18); dl.e[1] = DEV(19); dl.stats = DEV(11);

  addhip_plan_t* plan = NULL;
  addhip_ppo_marks_t pm;
  addhip_disc_marks_t dm;
  if (addhip_plan_create(&plan) || addhip_plan_record_begin(plan)) return 1;
  int rc = addhip_ppo_loss_fwd_bwd(&ppo, &pm, NULL);
  if (!rc) rc = addhip_disc_loss_fwd_bwd(&dl, &dm, NULL);
  if (addhip_plan_record_end(plan) || rc) { fprintf(stderr, "recording failed: %s\n", addhip_last_error()); return 1; }
  addhip_section_t sec[10];
  const int ns = addhip_update_schedule(0, &pm, &dm, sec, 10);
  int gemms = 0;
  double flop = 0.0;
  for (int i = 0; i < addhip_plan_size(plan); ++i) {
    addhip_gemm_t g[ADDHIP_GEMM_MAX_GROUP];
    const int k = addhip_plan_call_gemms(plan, i, g, ADDHIP_GEMM_MAX_GROUP);
    for (int j = 0; j < k; ++j) flop += 2.0 * g[j].M * g[j].N * g[j].K;
    gemms += k > 0;
  }
  printf("version=%d launches=%d gemm_launches=%d gflop=%.1f sections=%d actor=[0,%d) early=%d critic_end=%d disc_end=%d first=%s last=%s\n", addhip_version(),
         addhip_plan_size(plan), gemms, flop / 1e9, ns, pm.actor_end, pm.actor_early, pm.launches, pm.launches + dm.launches, addhip_plan_call_name(plan, 0),
         addhip_plan_call_name(plan, addhip_plan_size(plan) - 1));
  /* a refused call leaves a message and the plan unchanged */
  ppo.rows = Mb + 2;
  addhip_plan_record_begin(plan);
  rc = addhip_ppo_loss_fwd_bwd(&ppo, &pm, NULL);
  addhip_plan_record_end(plan);
  printf("refused=%d message=%s\n", rc != 0, addhip_last_error());
  addhip_plan_destroy(plan);
  return 0;
}
