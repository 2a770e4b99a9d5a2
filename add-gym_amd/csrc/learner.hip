// Composite entry points (include/addhip.h, "composite entry points"): the MLP passes and the loss sections of one optimiser step,
// assembled from the library's own entry points.  Host code only -- every launch goes through an extern "C" function of this library, so a
// call made while a plan is being recorded (record.h) records those launches instead.
//
// Restates the Sequential stacks of PPOModel / ADDModel (learning/ppo_model.py:13-59, learning/add/add_model.py:12-46: Linear + ReLU, linear
// heads), PPOAgent._compute_actor_loss / _compute_critic_loss with their autograd backward (learning/ppo_agent.py:194-275,
// base_agent.py:522-546) and ADDAgent._compute_disc_loss (learning/add/add_agent.py:141-202, amp_agent.py:177-192) as explicit launches.
#include "common.h"
#include "record.h"

namespace {

// A composite entry point that fails after some of its launches were recorded (a check that sits between launches) leaves the plan as it
// found it: the plan's size at entry is restored on every non-zero return.
struct PlanGuard {
  int size = addhip::record_size();
  int done(int rc) const {
    if (rc != 0 && addhip::recording()) addhip::record_truncate(size);
    return rc;
  }
};

addhip_gemm_t gemm(int M, int N, int K, const void* A, int lda, int akc, const void* B, int ldb, int bkc, float* C, int ldc, int epilogue = ADDHIP_EPI_NONE,
                   const float* bias = nullptr) {
  addhip_gemm_t g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K;
  g.A = static_cast<const float*>(A); g.lda = lda; g.a_kcontig = akc;
  g.B = static_cast<const float*>(B); g.ldb = ldb; g.b_kcontig = bkc;
  g.C = C; g.ldc = ldc;
  g.epilogue = epilogue;
  g.bias = bias;
  g.split_k = 1;
  g.alpha = 1.0f;
  g.precision = ADDHIP_PREC_F32;
  return g;
}

// K slices of a weight-gradient GEMM dW[out, in] over `rows` rows: enough 128x128 tiles x slices to fill the chip, at least 256 rows a slice
int split_k_for(int out_dim, int in_dim, int64_t rows) {
  const int tiles = ((out_dim + 127) / 128) * ((in_dim + 127) / 128);
  int s = (512 + tiles - 1) / tiles;
  s = s < 1 ? 1 : (s > 32 ? 32 : s);
  while (s > 1 && rows / s < 256) s /= 2;
  return s;
}

// 16-bit storage of the net's GEMM operands: 0 = none (fp32 operands), ADDHIP_STORE_BF16 (bf16 storage), ADDHIP_STORE_BF16X3 (plane storage)
// (storage 0 with precision ADDHIP_PREC_BF16: fp32 operands cut to bf16 on the way into LDS, one product per term -- no *16 buffers)
int store_fmt(const addhip_mlp_t& n) { return n.storage == ADDHIP_STORE_BF16X3 ? ADDHIP_STORE_BF16X3 : n.storage == ADDHIP_STORE_BF16 ? ADDHIP_STORE_BF16 : 0; }
bool storage16(const addhip_mlp_t& n) { return store_fmt(n) != 0; }  // (either format: the *16 buffers are the operands)
void set_prec(addhip_gemm_t& g, const addhip_mlp_t& n) {
  g.precision = n.precision;
  g.operands_bf16 = store_fmt(n);
  g.c16_planes = store_fmt(n);
}

// ADDHIP_PREC_F16X2: the net's tracked-maximum slots of tensor t (include/addhip.h: ADDHIP_MLP_AMAX_*), or NULL in every other mode
constexpr int AMAX_H = ADDHIP_MLP_AMAX_H, AMAX_DZ = ADDHIP_MLP_AMAX_DZ, AMAX_A = ADDHIP_MLP_AMAX_A, AMAX_G = ADDHIP_MLP_AMAX_G, AMAX_E = ADDHIP_MLP_AMAX_E;
bool f16x2(const addhip_mlp_t& n) { return n.precision == ADDHIP_PREC_F16X2 && n.amax && n.w_amax; }
uint32_t* amax_of(const addhip_mlp_t& n, int t) { return f16x2(n) ? n.amax + (size_t)t * ADDHIP_AMAX_SLOTS : nullptr; }
void set_amax(addhip_gemm_t& g, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  g.a_amax = a;
  g.b_amax = b;
  g.amax_out = out;
}

// A few rows past a multiple of 128 (the discriminator's extra zero-difference sample: Mb + 1 rows) would cost a whole extra round of
// 128-row tiles: they go into a second, tiny launch instead.  (Not with bf16 storage: its GEMM keeps 4 workgroups per CU in flight, so
// eight more tiles cost less than the launch, which sits on the discriminator's serial chain.)
struct Chunks { int n; int64_t r0[2], cnt[2]; };
Chunks row_chunks(const addhip_mlp_t& net, int64_t rows) {
  const int64_t rem = rows % 128;
  if (rows > 128 && rem > 0 && rem <= 8 && !storage16(net)) return {2, {0, rows - rem}, {rows - rem, rem}};
  return {1, {0, 0}, {rows, 0}};
}

// mask of the backward pass through the ReLU of `layer` for rows [r0, r0 + cnt): the sign bits where the forward pass wrote them (row
// chunks of more than 8 rows), the fp32 activations otherwise
void set_mask(addhip_gemm_t& g, const addhip_mlp_t& net, int layer, int64_t r0, int64_t cnt, bool bits_valid) {
  const int h = net.hidden[layer];
  if (bits_valid && (cnt > 8 || storage16(net))) {
    const int ldb = (h + 31) / 32;
    g.mask_bits = net.hbits[layer] + r0 * ldb;
    g.ldbits = ldb;
  } else {
    g.mask = net.h[layer] + r0 * h;
    g.ldmask = h;
  }
}

// column sums behind a gradient: by float atomics, or (net.deterministic) in a fixed order through the net's scratch
int col_sum(const addhip_mlp_t& net, const float* X, int M, int N, int ld, float* out, float scale, int accumulate, void* stream) {
  if (net.deterministic) return addhip_col_sum_ordered(X, M, N, ld, out, scale, accumulate, net.ordered_scratch, stream);
  return addhip_col_sum(X, M, N, ld, out, scale, accumulate, stream);
}
float* ordered_of(const addhip_mlp_t& net) { return net.deterministic ? net.ordered_scratch : nullptr; }

int check_net(const addhip_mlp_t* net, int64_t rows, const char* who) {
  ADDHIP_REQUIRE(net, "%s: null net", who);
  ADDHIP_REQUIRE(net->num_hidden >= 1 && net->num_hidden <= ADDHIP_MLP_MAX_HIDDEN, "%s: 1..%d hidden layers", who, ADDHIP_MLP_MAX_HIDDEN);
  ADDHIP_REQUIRE(rows > 0 && rows <= net->rows_cap, "%s: %lld rows, workspace holds %d", who, (long long)rows, net->rows_cap);
  ADDHIP_REQUIRE(net->in_ld >= net->in_dim && net->in_dim > 0, "%s: bad input width", who);
  ADDHIP_REQUIRE(net->storage == 0 || (net->storage == ADDHIP_STORE_BF16 && net->precision == ADDHIP_PREC_BF16) || net->storage == ADDHIP_STORE_BF16X3,
                 "%s: storage is 0, ADDHIP_STORE_BF16 (with precision ADDHIP_PREC_BF16) or ADDHIP_STORE_BF16X3", who);
  for (int i = 0; i < net->num_hidden; ++i) {
    ADDHIP_REQUIRE(net->hidden[i] > 0 && net->W[i] && net->b[i] && net->hbits[i], "%s: layer %d incomplete", who, i);
    if (storage16(*net)) ADDHIP_REQUIRE(net->W16[i] && net->h16[i] && net->dz16[i], "%s: layer %d lacks its bf16 / plane-storage buffers", who, i);
    else ADDHIP_REQUIRE(net->h[i] && net->dz[i], "%s: layer %d lacks its fp32 buffers", who, i);
  }
  ADDHIP_REQUIRE(!storage16(*net) || net->h[net->num_hidden - 1], "%s: the last hidden layer is kept in fp32 too (loss heads)", who);
  if (net->deterministic)
    ADDHIP_REQUIRE(net->ordered_scratch && (!net->slabs || (net->bias_replicas && (int64_t)net->bias_replica_rows * 32 >= rows)),
                   "%s: deterministic reductions need ordered_scratch and one bias replica row per 32-row block (%d rows hold %lld)", who,
                   net->bias_replica_rows, (long long)rows);
  return 0;
}

#define LAUNCH(call)                \
  do {                              \
    if (int rc_ = (call)) return rc_; \
    ++launches;                     \
  } while (0)

int refresh_transposed(const addhip_mlp_t& net, int& launches, void* stream) {
  if (storage16(net) && net.t_count > 0)
    LAUNCH(addhip_shadow_refresh(net.flat_params, nullptr, net.flat_trans16, net.flat_count, net.t_offset, net.t_rows, net.t_cols, net.t_count, store_fmt(net), stream));
  return 0;
}

int forward(const addhip_mlp_t& net, const float* x, const uint16_t* x16, int64_t rows, const float* a_mean, const float* a_std, bool sign_bits, int& launches,
            void* stream, const uint32_t* x_amax = nullptr) {
  const int n = net.num_hidden;
  const bool s16 = storage16(net);
  // the fp16 split: this pass (and the loss / backward kernels behind it) rewrites the net's tracked maxima -- they start from zero
  if (f16x2(net)) LAUNCH(addhip_fill_zero(reinterpret_cast<float*>(net.amax), ADDHIP_MLP_AMAX_TENSORS * ADDHIP_AMAX_SLOTS, stream));
  const char* prev = reinterpret_cast<const char*>(s16 ? static_cast<const void*>(x16) : static_cast<const void*>(x));
  const int esz = s16 ? 2 * store_fmt(net) : 4;  // bytes per value of the operand rows
  int ld = net.in_ld, k = net.in_ld;
  for (int i = 0; i < n; ++i) {
    const int h = net.hidden[i];
    const Chunks ch = row_chunks(net, rows);
    for (int c = 0; c < ch.n; ++c) {
      const int64_t r0 = ch.r0[c], cnt = ch.cnt[c];
      float* C = s16 ? (i == n - 1 ? net.h[i] + r0 * h : nullptr) : net.h[i] + r0 * h;
      addhip_gemm_t g = gemm((int)cnt, h, k, prev + (size_t)esz * r0 * ld, ld, 1, s16 ? static_cast<const void*>(net.W16[i]) : static_cast<const void*>(net.W[i]), k, 1,
                             C, h, ADDHIP_EPI_BIAS_RELU, net.b[i]);
      set_prec(g, net);
      if (s16) {
        g.C16 = net.h16[i] + store_fmt(net) * r0 * h;
        g.ldc16 = h;
      } else if (i == 0) {
        g.a_mean = a_mean;
        g.a_std = a_std;
      }
      if (sign_bits && (s16 || cnt > 8)) {
        g.relu_bits = net.hbits[i] + r0 * ((h + 31) / 32);
        g.ldbits = (h + 31) / 32;
      }
      if (f16x2(net)) set_amax(g, i == 0 ? x_amax : amax_of(net, AMAX_H + i - 1), net.w_amax, amax_of(net, AMAX_H + i));
      LAUNCH(addhip_gemm_f32(&g, stream));
    }
    prev = reinterpret_cast<const char*>(s16 ? static_cast<const void*>(net.h16[i]) : static_cast<const void*>(net.h[i]));
    ld = h;
    k = h;
  }
  return 0;
}

int backward(const addhip_mlp_t& net, const float* x, const uint16_t* x16, int64_t rows, const addhip_extra_dw_t* extra, int flags, addhip_mlp_marks_t* marks,
             int& launches, void* stream, const uint32_t* x_amax = nullptr) {
  const int n = net.num_hidden;
  const bool s16 = storage16(net);
  const int esz = s16 ? 2 * store_fmt(net) : 4;
  const bool zeroed = flags & ADDHIP_BWD_GRADS_ZEROED, bits_valid = flags & ADDHIP_BWD_SIGN_BITS;
  const int base = launches;
  auto dz_ptr = [&](int i) { return reinterpret_cast<const char*>(s16 ? static_cast<const void*>(net.dz16[i]) : static_cast<const void*>(net.dz[i])); };
  if (s16 && !(flags & ADDHIP_BWD_TOP_CAST_DONE)) {
    if (store_fmt(net) == ADDHIP_STORE_BF16X3) LAUNCH(addhip_to_bf16x3(net.dz[n - 1], net.dz16[n - 1], rows, net.hidden[n - 1], net.hidden[n - 1], net.hidden[n - 1], stream));
    else LAUNCH(addhip_to_bf16(net.dz[n - 1], net.dz16[n - 1], rows, net.hidden[n - 1], net.hidden[n - 1], net.hidden[n - 1], stream));
  }
  for (int i = n - 1; i >= 0; --i) {
    const int out_d = net.hidden[i], in_ld = i == 0 ? net.in_ld : net.hidden[i - 1];
    const void* inp = i == 0 ? (s16 ? static_cast<const void*>(x16) : static_cast<const void*>(x))
                             : (s16 ? static_cast<const void*>(net.h16[i - 1]) : static_cast<const void*>(net.h[i - 1]));
    const int s = split_k_for(out_d, in_ld, rows);
    const int64_t slab = (int64_t)out_d * in_ld;
    float* slabs = (net.slabs_top && i == n - 1) ? net.slabs_top : net.slabs;
    const bool has_extra = extra && extra[i].A;
    ADDHIP_REQUIRE((has_extra ? 2 : 1) * (int64_t)s * slab <= net.slab_floats, "mlp_backward: split-K scratch holds %lld floats, layer %d needs %lld",
                   (long long)net.slab_floats, i, (long long)((has_extra ? 2 : 1) * (int64_t)s * slab));
    if (marks) marks->dw_first[i] = launches - base;
    {  // dW = dz^T x  (both operands row-contiguous: m/n-contiguous for this product), K = rows cut into s slices
      addhip_gemm_t g = gemm(out_d, in_ld, (int)rows, dz_ptr(i), out_d, 0, inp, in_ld, 0, slabs, in_ld);
      g.split_k = s;
      set_prec(g, net);
      if (f16x2(net)) set_amax(g, amax_of(net, AMAX_DZ + i), i == 0 ? x_amax : amax_of(net, AMAX_H + i - 1), nullptr);
      LAUNCH(addhip_gemm_f32(&g, stream));
    }
    int total = s;
    if (has_extra) {
      addhip_gemm_t g = gemm(out_d, in_ld, (int)extra[i].rows, extra[i].A, extra[i].lda, 0, extra[i].B, extra[i].ldb, 0, slabs + (int64_t)s * slab, in_ld);
      g.split_k = s;
      set_prec(g, net);
      if (f16x2(net)) set_amax(g, extra[i].a_amax, extra[i].b_amax, nullptr);
      LAUNCH(addhip_gemm_f32(&g, stream));
      total = 2 * s;
    }
    // the combine of this layer's weight gradient; with it (replicated bias sums) the bias gradient the dX GEMM above left spread over the
    // replica rows -- summed into gb[i] and cleared for the next step in the same launch
    const bool reps = net.bias_replicas && net.bias_replica_rows > 1;
    if (reps && (i < n - 1 || (flags & ADDHIP_BWD_TOP_BIAS_REPLICAS)) && slab % 4 == 0)
      LAUNCH(addhip_slab_reduce_pair(slabs, total, slab, net.gW[i], slab, 1.0f, (flags & ADDHIP_BWD_ACCUMULATE_DW) ? 1 : 0, net.bias_replicas, net.bias_replica_rows,
                                     out_d, net.gb[i], out_d, 1, 1, stream));
    else
      LAUNCH(addhip_slab_reduce(slabs, total, slab, net.gW[i], slab, 1.0f, (flags & ADDHIP_BWD_ACCUMULATE_DW) ? 1 : 0, stream));
    if (marks) marks->dw_last[i] = launches - base;
    if (i == n - 1 && !(flags & ADDHIP_BWD_TOP_BIAS_DONE)) {
      ADDHIP_REQUIRE(net.dz[i], "mlp_backward: the top bias gradient is summed from the fp32 dz[last]");
      LAUNCH(col_sum(net, net.dz[i], (int)rows, out_d, out_d, net.gb[i], 1.0f, zeroed ? 1 : 0, stream));
    }
    // every gradient of this net except W[0] / b[0] is final here (b[1] came with the dX GEMM of layer 2 -- or, for a two-layer net
    // whose caller left the top bias to this pass, with the column sum just above; the head's with the loss kernels): an early bucket
    // for the data-parallel exchange
    if (marks && i == 1) marks->early = launches - base;
    if (i > 0) {
      const int prev_d = net.hidden[i - 1];
      if (!zeroed) LAUNCH(addhip_fill_zero(net.gb[i - 1], prev_d, stream));
      const Chunks ch = row_chunks(net, rows);
      for (int c = 0; c < ch.n; ++c) {
        const int64_t r0 = ch.r0[c], cnt = ch.cnt[c];
        // dX = dz W: the fp32 path reads W[out, in] n-contiguously; bf16 storage reads the transposed shadow W^T[in, out] k-contiguously
        addhip_gemm_t g = s16 ? gemm((int)cnt, prev_d, out_d, dz_ptr(i) + (size_t)esz * r0 * out_d, out_d, 1, net.W16t[i], out_d, 1, nullptr, prev_d, ADDHIP_EPI_MASK)
                              : gemm((int)cnt, prev_d, out_d, dz_ptr(i) + (size_t)esz * r0 * out_d, out_d, 1, net.W[i], prev_d, 0, net.dz[i - 1] + r0 * prev_d,
                                     prev_d, ADDHIP_EPI_MASK);
        ADDHIP_REQUIRE(!s16 || net.W16t[i], "mlp_backward: layer %d lacks its transposed bf16 shadow", i);
        g.colsum = net.gb[i - 1];
        // (the big row chunk only: a few-row chunk adds straight to gb, which the paired combine of layer i-1 then adds the replicas to)
        if (reps && cnt > 8 && ((int64_t)net.hidden[i - 1] * (i - 1 == 0 ? net.in_ld : net.hidden[i - 2])) % 4 == 0) {
          g.colsum = net.bias_replicas;
          g.colsum_replicas = net.bias_replica_rows;
          g.ldcs = prev_d;
        }
        set_prec(g, net);
        if (s16) {
          g.C16 = net.dz16[i - 1] + store_fmt(net) * r0 * prev_d;
          g.ldc16 = prev_d;
        }
        set_mask(g, net, i - 1, r0, cnt, bits_valid);
        if (f16x2(net)) set_amax(g, amax_of(net, AMAX_DZ + i), net.w_amax, amax_of(net, AMAX_DZ + i - 1));
        LAUNCH(addhip_gemm_f32(&g, stream));
      }
    }
  }
  if (marks) marks->launches = launches - base;
  return 0;
}

}  // namespace

extern "C" int addhip_mlp_forward(const addhip_mlp_t* net, const float* x, const uint16_t* x16, int64_t rows, const float* a_mean, const float* a_std,
                                  int32_t sign_bits, const uint32_t* x_amax, void* stream) {
  if (int rc = check_net(net, rows, "mlp_forward")) return rc;
  ADDHIP_REQUIRE(storage16(*net) ? (x16 && !a_mean) : (x != nullptr), "mlp_forward: bf16 storage takes x16 (already normalised), the other modes x");
  ADDHIP_REQUIRE((a_mean == nullptr) == (a_std == nullptr), "mlp_forward: a_mean and a_std come together");
  int launches = 0;
  const PlanGuard guard;
  return guard.done(forward(*net, x, x16, rows, a_mean, a_std, sign_bits != 0, launches, stream, x_amax));
}

extern "C" int addhip_mlp_backward(const addhip_mlp_t* net, const float* x, const uint16_t* x16, int64_t rows, const addhip_extra_dw_t* extra, int32_t flags,
                                   addhip_mlp_marks_t* marks, const uint32_t* x_amax, void* stream) {
  if (int rc = check_net(net, rows, "mlp_backward")) return rc;
  ADDHIP_REQUIRE(storage16(*net) ? (x16 != nullptr) : (x != nullptr), "mlp_backward: input rows missing");
  ADDHIP_REQUIRE(net->slabs && net->slab_floats > 0, "mlp_backward: split-K scratch missing");
  ADDHIP_REQUIRE(!storage16(*net) || (flags & ADDHIP_BWD_GRADS_ZEROED), "mlp_backward: bf16 storage expects the gradient buffer zeroed by the caller");
  for (int i = 0; i < net->num_hidden; ++i) ADDHIP_REQUIRE(net->gW[i] && net->gb[i], "mlp_backward: gradient of layer %d missing", i);
  if (marks) memset(marks, 0, sizeof(*marks));
  int launches = 0;
  const PlanGuard guard;
  return guard.done(backward(*net, x, x16, rows, extra, flags, marks, launches, stream, x_amax));
}

static int ppo_loss_impl(const addhip_ppo_loss_t* d, addhip_ppo_marks_t* marks, void* stream) {
  ADDHIP_REQUIRE(d && d->actor && d->critic, "ppo_loss_fwd_bwd: null argument");
  const addhip_mlp_t &A = *d->actor, &Cn = *d->critic;
  const int Mb = d->rows;
  if (int rc = check_net(&A, Mb, "ppo_loss_fwd_bwd (actor)")) return rc;
  if (int rc = check_net(&Cn, Mb, "ppo_loss_fwd_bwd (critic)")) return rc;
  // HR: rows of the actor's head matrix -- 32 (29 action means + padding), or 64 when the log-std is a second head (actor_std_type VARIABLE:
  // rows 32..60; the two heads are one 64-wide product in every launch below, the loss kernel reads the log-std columns behind the mean's)
  const int HR = A.head_rows;
  ADDHIP_REQUIRE((HR == 32 || HR == 64) && Cn.head_rows == 1, "ppo_loss_fwd_bwd: the actor's head is 32 rows (29 + padding) or 64 (mean | log-std), the critic's 1");
  ADDHIP_REQUIRE(HR == 32 || !d->dist, "ppo_loss_fwd_bwd: a log-std head (64 head rows) and a log-std vector (dist) exclude each other");
  ADDHIP_REQUIRE(d->norm_obs && d->norm_action && d->old_logp && d->adv && d->tar_val && d->rand_mask, "ppo_loss_fwd_bwd: minibatch rows missing");
  ADDHIP_REQUIRE(d->mean && d->d_mean && d->dv && d->num_valid && d->stats, "ppo_loss_fwd_bwd: workspace missing");
  ADDHIP_REQUIRE(A.Wh && A.bh && A.gWh && A.gbh && Cn.Wh && Cn.bh && Cn.gWh && Cn.gbh && A.slabs && Cn.slabs, "ppo_loss_fwd_bwd: head parameters / scratch missing");
  const bool s16 = storage16(A);
  ADDHIP_REQUIRE(storage16(Cn) == s16 && (!s16 || d->norm_obs16), "ppo_loss_fwd_bwd: both nets in one storage mode (bf16 storage: norm_obs16 too)");
  const int nA = A.num_hidden, nC = Cn.num_hidden, hA = A.hidden[nA - 1], hC = Cn.hidden[nC - 1];
  ADDHIP_REQUIRE(32LL * HR * hA <= A.slab_floats, "ppo_loss_fwd_bwd: the actor's split-K scratch is smaller than its head gradient's 32 slabs");
  const int bwd = ADDHIP_BWD_GRADS_ZEROED | ADDHIP_BWD_TOP_BIAS_DONE | ADDHIP_BWD_TOP_CAST_DONE | ADDHIP_BWD_SIGN_BITS;
  // the actor's top bias gradient comes as column sums of the head's dz GEMM: replicated like the dX GEMMs' (same-line atomics serialise)
  const bool top_reps = A.bias_replicas && A.bias_replica_rows > 1 && ((int64_t)hA * (nA > 1 ? A.hidden[nA - 2] : A.in_ld)) % 4 == 0;
  int launches = 0;
  addhip_mlp_marks_t mk;
  // ---- actor (ppo_agent.py:194-232, 247-275)
  if (int rc = forward(A, d->norm_obs, d->norm_obs16, Mb, nullptr, nullptr, true, launches, stream, d->norm_obs_amax)) return rc;
  if (int rc = refresh_transposed(A, launches, stream)) return rc;  // beside the other nets' GEMMs (the optimiser step wrote the flat shadow)
  // The head section: one launch (csrc/actor_head.hip: head forward, loss, gradient, backward step into the last hidden layer) where the
  // last hidden layer is 128 / 256 / 512 wide; the three 32-wide GEMMs + loss + column sums otherwise.
  const int head_slabs = addhip_actor_head_slabs(Mb);
  const bool fused = HR == 32 && (hA == 128 || hA == 256 || hA == 512) && (int64_t)head_slabs * ADDHIP_ACTOR_HEAD_SLAB(hA) <= A.slab_floats;
  ADDHIP_REQUIRE(!d->dist || d->g_logstd, "ppo_loss_fwd_bwd: a trainable log-std (dist) needs its gradient g_logstd");
  const bool planes = store_fmt(A) == ADDHIP_STORE_BF16X3;
  LAUNCH(addhip_count_mask(d->rand_mask, Mb, d->num_valid, stream));
  if (fused) {
    addhip_actor_head_t h;
    memset(&h, 0, sizeof(h));
    h.rows = Mb; h.hidden = hA;
    h.H = A.h[nA - 1]; h.Wh = A.Wh; h.bh = A.bh;
    h.norm_action = d->norm_action; h.old_logp = d->old_logp; h.adv = d->adv; h.rand_mask = d->rand_mask; h.n_valid = d->num_valid;
    h.action_std = d->action_std; h.logp_const = d->logp_const; h.dist = d->dist; h.clip_ratio = d->ppo_clip_ratio; h.bound_weight = d->action_bound_weight;
    h.reg_weight = d->action_reg_weight; h.loss_scale = d->grad_scale;
    h.dz = s16 ? nullptr : A.dz[nA - 1]; h.dz16 = s16 ? A.dz16[nA - 1] : nullptr; h.planes16 = store_fmt(A);
    h.slabs = A.slabs; h.num_slabs = head_slabs;
    h.gb_top = top_reps ? A.bias_replicas : A.gb[nA - 1]; h.gb_replicas = top_reps ? A.bias_replica_rows : 1; h.ld_gb = hA;
    h.stats = d->stats;
    h.amax = amax_of(A, AMAX_DZ + nA - 1);
    LAUNCH(addhip_actor_head(&h, stream));
    const int64_t stride = ADDHIP_ACTOR_HEAD_SLAB(hA);
    const bool ls_follows = d->dist && d->g_logstd == A.gbh + 32;
    if (A.gbh == A.gWh + 32LL * hA) {  // (the flat layout places a head's bias behind its weight, a trainable log-std behind the bias)
      LAUNCH(addhip_slab_reduce(A.slabs, head_slabs, stride, A.gWh, 32LL * hA + (ls_follows ? 64 : 32), 1.0f, 0, stream));
    } else {
      LAUNCH(addhip_slab_reduce(A.slabs, head_slabs, stride, A.gWh, 32LL * hA, 1.0f, 0, stream));
      LAUNCH(addhip_slab_reduce(A.slabs + 32LL * hA, head_slabs, stride, A.gbh, ls_follows ? 64 : 32, 1.0f, 0, stream));
    }
    if (d->dist && !ls_follows) LAUNCH(addhip_slab_reduce(A.slabs + 32LL * hA + 32, head_slabs, stride, d->g_logstd, 32, 1.0f, 0, stream));
  } else {
    {
      addhip_gemm_t g = gemm(Mb, HR, hA, A.h[nA - 1], hA, 1, A.Wh, hA, 1, d->mean, HR, ADDHIP_EPI_BIAS, A.bh);
      g.precision = d->head_precision;
      LAUNCH(addhip_gemm_f32(&g, stream));
    }
    LAUNCH(addhip_actor_loss(d->mean, d->norm_action, d->old_logp, d->adv, d->rand_mask, Mb, d->action_std, d->logp_const, d->dist, d->ppo_clip_ratio,
                             d->action_bound_weight, d->action_reg_weight, d->grad_scale, d->num_valid, d->d_mean, d->g_logstd, d->stats, HR,
                             HR == 64 ? d->mean + 32 : nullptr, HR == 64 ? d->action_entropy_weight : 0.f, stream));
    {  // head weight gradient: d_mean^T h over Mb rows, 32 K slices
      addhip_gemm_t g = gemm(HR, hA, Mb, d->d_mean, HR, 0, A.h[nA - 1], hA, 0, A.slabs, hA);
      g.split_k = 32;
      g.precision = d->head_precision;
      LAUNCH(addhip_gemm_f32(&g, stream));
    }
    LAUNCH(addhip_slab_reduce(A.slabs, 32, (int64_t)HR * hA, A.gWh, (int64_t)HR * hA, 1.0f, 0, stream));
    LAUNCH(col_sum(A, d->d_mean, Mb, HR, HR, A.gbh, 1.0f, 1, stream));
    {  // dz[last] = (d_mean Wh) * relu'(h[last]); bf16 storage: written as bf16 directly
      // (plane storage: an fp32-operand GEMM cannot write planes -- it leaves the fp32 dz and the backward pass splits it first)
      addhip_gemm_t g = gemm(Mb, hA, HR, d->d_mean, HR, 1, A.Wh, hA, 0, s16 && !planes ? nullptr : A.dz[nA - 1], hA, ADDHIP_EPI_MASK);
      g.colsum = A.gb[nA - 1];
      if (top_reps) {  // spread over the replica rows like the dX GEMMs' sums; the top layer's combine folds them into gb[last]
        g.colsum = A.bias_replicas;
        g.colsum_replicas = A.bias_replica_rows;
        g.ldcs = hA;
      }
      g.precision = d->head_precision;
      if (s16 && !planes) {
        g.C16 = A.dz16[nA - 1];
        g.ldc16 = hA;
      }
      set_mask(g, A, nA - 1, 0, Mb, true);
      g.amax_out = amax_of(A, AMAX_DZ + nA - 1);  // (d_mean's own maximum is not tracked: this launch runs the exact bf16 split or the fp32 MFMA)
      LAUNCH(addhip_gemm_f32(&g, stream));
    }
  }
  const bool cast_top = planes && !fused;  // (the unfused path leaves fp32 dz in plane-storage mode: the backward pass splits it)
  int at = launches;
  if (int rc = backward(A, d->norm_obs, d->norm_obs16, Mb, nullptr, (cast_top ? bwd & ~ADDHIP_BWD_TOP_CAST_DONE : bwd) | (top_reps ? ADDHIP_BWD_TOP_BIAS_REPLICAS : 0), &mk,
                        launches, stream, d->norm_obs_amax))
    return rc;
  const int actor_early = at + mk.early, actor_end = launches;
  // ---- critic (ppo_agent.py:234-245, base_agent.py:522-546)
  if (int rc = forward(Cn, d->norm_obs, d->norm_obs16, Mb, nullptr, nullptr, true, launches, stream, d->norm_obs_amax)) return rc;
  if (int rc = refresh_transposed(Cn, launches, stream)) return rc;
  LAUNCH(addhip_critic_head(Cn.h[nC - 1], hC, hC, Mb, Cn.Wh, Cn.bh, d->tar_val, d->critic_loss_weight * d->grad_scale, nullptr, d->dv, d->stats + 8, stream));
  LAUNCH(addhip_head_backward(d->dv, Cn.Wh, Cn.h[nC - 1], hC, hC, Mb, s16 ? nullptr : Cn.dz[nC - 1], s16 ? Cn.dz16[nC - 1] : nullptr, store_fmt(Cn), Cn.gWh,
                              Cn.gbh, Cn.gb[nC - 1], amax_of(Cn, AMAX_DZ + nC - 1), ordered_of(Cn), stream));
  at = launches;
  if (int rc = backward(Cn, d->norm_obs, d->norm_obs16, Mb, nullptr, bwd, &mk, launches, stream, d->norm_obs_amax)) return rc;
  if (marks) *marks = addhip_ppo_marks_t{launches, actor_end, actor_early, at + mk.early};
  return 0;
}

static int disc_loss_impl(const addhip_disc_loss_t* d, addhip_disc_marks_t* marks, void* stream) {
  ADDHIP_REQUIRE(d && d->disc, "disc_loss_fwd_bwd: null argument");
  const addhip_mlp_t& D = *d->disc;
  const int Mb = d->rows, Md = Mb + 1;
  if (int rc = check_net(&D, Md, "disc_loss_fwd_bwd")) return rc;
  const int n = D.num_hidden, last = n - 1;
  ADDHIP_REQUIRE(D.head_rows == 1 && D.Wh && D.bh && D.gWh && D.gbh && D.slabs && D.slabs_top, "disc_loss_fwd_bwd: head parameters / both split-K scratches missing");
  const bool s16 = storage16(D);
  ADDHIP_REQUIRE(d->norm_diff && d->dlogit && d->g && d->e[last] && d->stats && d->disc_dim > 0 && d->disc_dim <= D.in_ld, "disc_loss_fwd_bwd: buffers missing");
  ADDHIP_REQUIRE(s16 ? (d->norm_diff16 && d->G16 != nullptr) : (d->G != nullptr), "disc_loss_fwd_bwd: the penalty chain's buffers for this storage mode are missing");
  for (int i = 0; i < n; ++i)
    ADDHIP_REQUIRE(s16 ? (d->a16[i] && (i == last || d->e16[i]) && D.W16t[i]) : (d->a[i] && d->e[i]), "disc_loss_fwd_bwd: the penalty chain's buffers of layer %d are missing", i);
  const int DS = D.in_ld, dl = D.hidden[last];
  const float ls = d->loss_scale, wd = d->weight_decay;
  int launches = 0;
  // L2 terms (add_agent.py:161-164, 181-186): logit regularisation on the head weights, weight decay on every weight.  They go into the
  // freshly zeroed gradient FIRST (the weight gradients are added to them by the split-K combines), where they run beside the other
  // nets' GEMMs instead of alone at the end of the step's longest chain.
  for (int i = 0; i < n; ++i)
    LAUNCH(addhip_l2_grad(D.W[i], D.gW[i], (int64_t)D.hidden[i] * (i == 0 ? DS : D.hidden[i - 1]), 2.0f * ls * wd, d->stats + 24, stream));
  LAUNCH(addhip_l2_grad(D.Wh, D.gWh, (int64_t)dl, 2.0f * ls * (wd + d->logit_reg), d->stats + 25, stream));
  if (int rc = forward(D, d->norm_diff, d->norm_diff16, Md, nullptr, nullptr, true, launches, stream, d->norm_diff_amax)) return rc;
  if (int rc = refresh_transposed(D, launches, stream)) return rc;
  const int m_head = launches;
  // logit loss on Mb agent rows (negative) and the demo... rows of h[last]: row Mb = the zero-difference sample (positive)
  LAUNCH(addhip_disc_head(D.h[last], dl, dl, Mb, D.h[last] + (size_t)Mb * dl, D.Wh, D.bh, ls, d->dlogit, d->dlogit + Mb, d->stats + 12, stream));
  LAUNCH(addhip_head_backward(d->dlogit, D.Wh, D.h[last], dl, dl, Md, s16 ? nullptr : D.dz[last], s16 ? D.dz16[last] : nullptr, store_fmt(D), D.gWh, D.gbh, D.gb[last],
                              amax_of(D, AMAX_DZ + last), ordered_of(D), stream));
  const int m_chain = launches;
  // gradient penalty (hand-derived double backward of add_agent.py:166-178; include/addhip.h has the chain):  a[last] = w_head * m[last],
  // down to g = a[0] W[0];  G = d penalty / d g;  then e[0] = (G W[0]^T) * m[0] up to e[last]
  LAUNCH(addhip_bcast_mask(D.Wh, D.h[last], dl, dl, Mb, s16 ? nullptr : d->a[last], s16 ? d->a16[last] : nullptr, store_fmt(D), amax_of(D, AMAX_A + last), stream));
  addhip_extra_dw_t extra[ADDHIP_MLP_MAX_HIDDEN];
  memset(extra, 0, sizeof(extra));
  auto chain = [&](addhip_gemm_t g, int mask_layer, int t_in = -1, int t_out = -1) -> int {  // t_in / t_out: tracked-maximum slots of A / the result
    set_prec(g, D);
    if (mask_layer >= 0) set_mask(g, D, mask_layer, 0, Mb, true);
    if (f16x2(D)) set_amax(g, t_in >= 0 ? amax_of(D, t_in) : nullptr, D.w_amax, t_out >= 0 ? amax_of(D, t_out) : nullptr);
    return addhip_gemm_f32(&g, stream);
  };
  // (storage modes: a[last] and G are written as 16-bit rows by their kernels, a[i] / e[i < last] leave their GEMMs as such, g and e[last] as fp32)
  for (int i = last; i >= 1; --i) {  // a[i-1] = (a[i] W[i]) * m[i-1]
    const int di = D.hidden[i], dp = D.hidden[i - 1];
    if (s16) {
      addhip_gemm_t gg = gemm(Mb, dp, di, d->a16[i], di, 1, D.W16t[i], di, 1, nullptr, dp, ADDHIP_EPI_MASK);
      gg.C16 = d->a16[i - 1]; gg.ldc16 = dp;
      LAUNCH(chain(gg, i - 1));
    } else {
      LAUNCH(chain(gemm(Mb, dp, di, d->a[i], di, 1, D.W[i], dp, 0, d->a[i - 1], dp, ADDHIP_EPI_MASK), i - 1, AMAX_A + i, AMAX_A + i - 1));
    }
  }
  const int d0 = D.hidden[0];
  if (s16) {
    LAUNCH(chain(gemm(Mb, DS, d0, d->a16[0], d0, 1, D.W16t[0], d0, 1, d->g, DS), -1));
    LAUNCH(addhip_grad_penalty(d->g, DS, d->disc_dim, Mb, ls * d->grad_penalty, nullptr, d->G16, store_fmt(D), d->stats + 20, nullptr, stream));
  } else {
    LAUNCH(chain(gemm(Mb, DS, d0, d->a[0], d0, 1, D.W[0], DS, 0, d->g, DS), -1, AMAX_A + 0));
    LAUNCH(addhip_grad_penalty(d->g, DS, d->disc_dim, Mb, ls * d->grad_penalty, d->G, nullptr, 0, d->stats + 20, amax_of(D, AMAX_G), stream));
  }
  for (int i = 0; i < n; ++i) {  // e[0] = (G W[0]^T) * m[0];  e[i] = (e[i-1] W[i]^T) * m[i]
    const int di = D.hidden[i], kin = i == 0 ? DS : D.hidden[i - 1];
    if (s16) {
      addhip_gemm_t gg = gemm(Mb, di, kin, i == 0 ? d->G16 : d->e16[i - 1], kin, 1, D.W16[i], kin, 1, i == last ? d->e[i] : nullptr, di, ADDHIP_EPI_MASK);
      if (i < last) { gg.C16 = d->e16[i]; gg.ldc16 = di; }
      LAUNCH(chain(gg, i));
      extra[i] = {d->a16[i], di, i == 0 ? d->G16 : d->e16[i - 1], kin, Mb, nullptr, nullptr};
    } else {
      LAUNCH(chain(gemm(Mb, di, kin, i == 0 ? d->G : d->e[i - 1], kin, 1, D.W[i], kin, 1, d->e[i], di, ADDHIP_EPI_MASK), i, i == 0 ? AMAX_G : AMAX_E + i - 1,
                   i < last ? AMAX_E + i : -1));
      extra[i] = {d->a[i], di, i == 0 ? d->G : d->e[i - 1], kin, Mb, amax_of(D, AMAX_A + i), amax_of(D, i == 0 ? AMAX_G : AMAX_E + i - 1)};
    }
  }
  LAUNCH(col_sum(D, d->e[last], Mb, dl, dl, D.gWh, 1.0f, 1, stream));
  const int m_bwd = launches;
  addhip_mlp_marks_t mk;
  if (int rc = backward(D, d->norm_diff, d->norm_diff16, Md, extra,
                        ADDHIP_BWD_GRADS_ZEROED | ADDHIP_BWD_TOP_BIAS_DONE | ADDHIP_BWD_ACCUMULATE_DW | ADDHIP_BWD_TOP_CAST_DONE | ADDHIP_BWD_SIGN_BITS, &mk, launches,
                        stream, d->norm_diff_amax))
    return rc;
  if (marks) *marks = addhip_disc_marks_t{launches, m_head, m_chain, m_bwd, m_bwd + mk.dw_first[last], m_bwd + mk.dw_last[last]};
  return 0;
}

extern "C" int addhip_ppo_loss_fwd_bwd(const addhip_ppo_loss_t* d, addhip_ppo_marks_t* marks, void* stream) {
  const PlanGuard guard;  // (a call refused between two of its launches leaves a plan being recorded unchanged)
  return guard.done(ppo_loss_impl(d, marks, stream));
}
extern "C" int addhip_disc_loss_fwd_bwd(const addhip_disc_loss_t* d, addhip_disc_marks_t* marks, void* stream) {
  const PlanGuard guard;
  return guard.done(disc_loss_impl(d, marks, stream));
}

extern "C" int addhip_update_schedule(int32_t base, const addhip_ppo_marks_t* ppo, const addhip_disc_marks_t* disc, addhip_section_t* out, int32_t capacity) {
  ADDHIP_REQUIRE(ppo && disc && out && capacity >= 10 && base >= 0, "update_schedule: bad arguments (10 sections)");
  const int a0 = base, ea = base + ppo->actor_early, end_a = base + ppo->actor_end, ec = base + ppo->critic_early, end_ac = base + ppo->launches;
  const int d0 = end_ac, d_head = d0 + disc->head, d_gp = d0 + disc->chain, d_bwd = d0 + disc->backward, dw_first = d0 + disc->top_dw_first,
            dw_last = d0 + disc->top_dw_last, end_d = d0 + disc->launches;
  // Actor and critic hand over everything but their first layers as soon as it is final, then their two first layers as ONE bucket as
  // soon as both are; the collectives are issued in the order they become ready, because one communicator runs them in issue order, and
  // only the discriminator's -- the last section to finish -- is left with nothing to hide behind.  The discriminator's section is the
  // longest chain of the step, so its independent pieces run side by side on two streams: the logit loss and its backward step through
  // the head (stream 3) beside the gradient-penalty chain (stream 2), then the top layer's weight gradient (stream 3, own split-K
  // scratch) beside the dX GEMM and the first layer's.  {stream, first, last, wait_before, wait_after, bucket}
  const addhip_section_t s[10] = {
      {0, a0, ea, -1, -1, 0},            // 0: actor up to its early mark -> bucket 0
      {1, end_a, ec, -1, -1, 1},         // 1: critic likewise -> bucket 1
      {1, ec, end_ac, -1, -1, -1},       // 2: the critic's first layer
      {0, ea, end_a, -1, 2, 3},          // 3: the actor's first layer -> bucket 3 (both first layers) once section 2 is in too
      {2, d0, d_head, -1, -1, -1},       // 4: L2 terms, forward
      {3, d_head, d_gp, 4, -1, -1},      // 5: logit loss, head backward -> top dz
      {2, d_gp, d_bwd, -1, -1, -1},      // 6: gradient-penalty chain
      {3, d_bwd, dw_first, -1, -1, -1},  // 7: (bf16 storage without a pre-cast top gradient: its rounding)
      {3, dw_first, dw_last, 6, -1, -1}, // 8: top-layer weight gradient (needs the chain's a2 / e1)
      {2, dw_last, end_d, 7, 8, 2},      // 9: dX, first-layer weight gradient -> bucket 2 once section 8 is in too
  };
  for (int i = 0; i < 10; ++i) out[i] = s[i];
  return 10;
}
