"""Learning core of the hot path: normalisers, actor step, TD(lambda) + advantage, the PPO /
critic / ADD-discriminator losses and AdamW (oracle restatement of add_gym/learning/*.py).
numpy for the scans and statistics; torch-CPU fp32 autograd for the MLP losses because the
algorithm there *is* reverse-mode differentiation (incl. the double backward of the
gradient penalty).  Test infrastructure."""
import math

import numpy as np
import torch

F = np.float32
DONE_NULL, DONE_FAIL, DONE_SUCC, DONE_TIME = 0, 1, 2, 3

# parameter tensors in optimiser order = registration order of the trainable parameters of
# ADDModel (ppo_model.py:36-59, add_model.py:29-46, base_agent.py:266-269)
PARAM_SHAPES = [
    ("_model._actor_layers.0.weight", (1024, 264)), ("_model._actor_layers.0.bias", (1024,)),
    ("_model._actor_layers.2.weight", (1024, 1024)), ("_model._actor_layers.2.bias", (1024,)),
    ("_model._actor_layers.4.weight", (512, 1024)), ("_model._actor_layers.4.bias", (512,)),
    ("_model._action_dist._mean_net.weight", (29, 512)), ("_model._action_dist._mean_net.bias", (29,)),
    ("_model._critic_layers.0.weight", (1024, 264)), ("_model._critic_layers.0.bias", (1024,)),
    ("_model._critic_layers.2.weight", (1024, 1024)), ("_model._critic_layers.2.bias", (1024,)),
    ("_model._critic_layers.4.weight", (512, 1024)), ("_model._critic_layers.4.bias", (512,)),
    ("_model._critic_out.weight", (1, 512)), ("_model._critic_out.bias", (1,)),
    ("_model._disc_layers.0.weight", (1024, 114)), ("_model._disc_layers.0.bias", (1024,)),
    ("_model._disc_layers.2.weight", (512, 1024)), ("_model._disc_layers.2.bias", (512,)),
    ("_model._disc_logits.weight", (1, 512)), ("_model._disc_logits.bias", (1,)),
]


# hidden layer widths of the reference's net modules (nets/fc_*layers_*units.py: layer_sizes), by module name (nets/net_builder.py:5-11)
NET_LAYERS = {"fc_3layers_1024units": [1024, 1024, 512], "fc_2layers_1024units": [1024, 512], "fc_2layers_512units": [512, 256],
              "fc_2layers_256units": [256, 128], "fc_2layers_128units": [128, 64], "fc_2layers_64units": [64, 32]}


def param_shapes(nets=None, obs_dim=264, disc_dim=114):
    """PARAM_SHAPES for a choice of net modules {"actor_net": ..., "critic_net": ..., "disc_net": ...} (None: add_g1.yaml's)."""
    nets = {"actor_net": "fc_3layers_1024units", "critic_net": "fc_3layers_1024units", "disc_net": "fc_2layers_1024units", **(nets or {})}
    out = []
    for key, prefix, head, in_dim, head_dim in (("actor_net", "_model._actor_layers", "_model._action_dist._mean_net", obs_dim, 29),
                                                ("critic_net", "_model._critic_layers", "_model._critic_out", obs_dim, 1),
                                                ("disc_net", "_model._disc_layers", "_model._disc_logits", disc_dim, 1)):
        prev = in_dim
        for i, h in enumerate(NET_LAYERS[nets[key]]):
            out += [(f"{prefix}.{2 * i}.weight", (h, prev)), (f"{prefix}.{2 * i}.bias", (h,))]
            prev = h
        out += [(f"{head}.weight", (head_dim, prev)), (f"{head}.bias", (head_dim,))]
    return out


LOGSTD_KEY = "_model._action_dist._logstd_net"                     # actor_std_type CONSTANT: a trainable vector
LOGSTD_W, LOGSTD_B = LOGSTD_KEY + ".weight", LOGSTD_KEY + ".bias"    # actor_std_type VARIABLE: a second linear head on the actor's last layer


def synth_params(seed, obs_dim=264, disc_dim=114, bias_scale=0.02, nets=None, logstd=False):
    """Deterministic parameter set (numpy legacy RandomState: stable across versions) used by
    fixtures instead of shipping 17 MB of weights.  Same distributions as the reference's init
     except that biases are non-zero so that bias gradients/ReLU masks are exercised."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shape in (PARAM_SHAPES if nets is None and (obs_dim, disc_dim) == (264, 114) else param_shapes(nets, obs_dim, disc_dim)):
        shape = tuple(obs_dim if (s == 264) else disc_dim if (s == 114) else s for s in shape)
        if name.endswith("weight"):
            fan_in = shape[1]
            bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), ..)
            if "_mean_net" in name:
                bound = 0.01
            if "_disc_logits" in name:
                bound = 1.0
            out[name] = rng.uniform(-bound, bound, size=shape).astype(F)
        else:
            out[name] = (rng.uniform(-1, 1, size=shape) * bias_scale).astype(F)
    if logstd:  # a trainable log-std, different per action dimension, at its place in the registration order (before the mean head):
        # True / "constant": a vector (actor_std_type CONSTANT); "variable": a linear head on the last hidden layer (VARIABLE:
        # distribution_gaussian_diag.py:38-43 -- weights large enough here for the per-sample part to matter)
        r2 = np.random.RandomState(seed + 7919)
        ls = (np.log(0.05) + r2.uniform(-0.4, 0.4, size=29)).astype(F)
        ordered = {}
        for k, v in out.items():
            # (registration order: a Parameter of the distribution module precedes its sub-modules' -- CONSTANT's vector comes before the
            # mean head, VARIABLE's linear head, a sub-module created after the mean head, behind it)
            if k == "_model._action_dist._mean_net.weight" and logstd != "variable":
                ordered[LOGSTD_KEY] = ls
            ordered[k] = v
            if k == "_model._action_dist._mean_net.bias" and logstd == "variable":
                ordered[LOGSTD_W] = r2.uniform(-0.02, 0.02, size=out["_model._action_dist._mean_net.weight"].shape).astype(F)
                ordered[LOGSTD_B] = ls
        out = ordered
    return out


# ---------------------------------------------------------------- normalisers
class Normalizer:
    # normalizer.py:7-162
    def __init__(self, dim, mean=None, std=None, min_std=1e-4):
        self.count = 0
        self.mean = np.zeros(dim, F) if mean is None else np.asarray(mean, F).copy()
        self.std = np.ones(dim, F) if std is None else np.asarray(std, F).copy()
        self.mean_sq = None
        self.min_var = F(min_std * min_std)
        self.new_count, self.new_sum, self.new_sum_sq = 0, np.zeros(dim, F), np.zeros(dim, F)

    def record(self, x):
        x = np.asarray(x, F).reshape(-1, self.mean.shape[0])
        self.new_count += x.shape[0]
        self.new_sum = (self.new_sum + np.sum(x, axis=0, dtype=F)).astype(F)
        self.new_sum_sq = (self.new_sum_sq + np.sum(x * x, axis=0, dtype=F)).astype(F)

    def update(self):
        if self.mean_sq is None:
            self.mean_sq = (self.std * self.std + self.mean * self.mean).astype(F)
        if self.new_count == 0:
            return
        new_mean = self.new_sum / F(self.new_count)
        new_mean_sq = self.new_sum_sq / F(self.new_count)
        total = self.count + self.new_count
        w_old = F(self.count) / F(total)
        w_new = F(float(self.new_count)) / F(total)
        self.mean = (w_old * self.mean + w_new * new_mean).astype(F)
        self.mean_sq = (w_old * self.mean_sq + w_new * new_mean_sq).astype(F)
        self.count = total
        var = np.maximum(self.mean_sq - self.mean * self.mean, self.min_var)
        self.std = np.sqrt(var).astype(F)
        self.new_count = 0
        self.new_sum[:] = 0
        self.new_sum_sq[:] = 0

    def normalize(self, x):
        return ((np.asarray(x, F) - self.mean) / self.std).astype(F)

    def unnormalize(self, x):
        return (np.asarray(x, F) * self.std + self.mean).astype(F)


class DiffNormalizer:
    # diff_normalizer.py:6-86
    def __init__(self, dim, min_diff=1e-4):
        self.count = 0
        self.mean_abs = np.ones(dim, F)
        self.min_diff = F(min_diff)
        self.new_count, self.new_sum_abs = 0, np.zeros(dim, F)

    def record(self, x):
        x = np.asarray(x, F).reshape(-1, self.mean_abs.shape[0])
        self.new_count += x.shape[0]
        self.new_sum_abs = (self.new_sum_abs + np.sum(np.abs(x), axis=0, dtype=F)).astype(F)

    def update(self):
        new_mean = self.new_sum_abs / F(self.new_count)
        total = self.count + self.new_count
        w_old = F(self.count) / F(total)
        w_new = F(float(self.new_count)) / F(total)
        self.mean_abs = (w_old * self.mean_abs + w_new * new_mean).astype(F)
        self.count = total
        self.new_count = 0
        self.new_sum_abs[:] = 0

    def normalize(self, x):
        return (np.asarray(x, F) / np.maximum(self.mean_abs, self.min_diff)).astype(F)


# ---------------------------------------------------------------- TD(lambda), advantage
def td_lambda_return(r, next_vals, done, discount, td_lambda):
    # base_agent.py:624-647
    r, next_vals = np.asarray(r, F), np.asarray(next_vals, F)
    reset = (done != DONE_NULL).astype(F)
    ret = np.zeros_like(r)
    ret[-1] = r[-1] + F(discount) * next_vals[-1]
    for i in range(r.shape[0] - 2, -1, -1):
        lam = F(td_lambda) * (F(1.0) - reset[i])
        ret[i] = r[i] + F(discount) * ((F(1.0) - lam) * next_vals[i] + lam * ret[i + 1])
    return ret.astype(F)


def advantages(ret, vals, rand_mask, adv_clip):
    # ppo_agent.py:141-153: unbiased std over samples with random actions
    adv = (ret - vals).astype(F)
    sel = adv.reshape(-1)[(rand_mask == 1.0).reshape(-1)]
    mean = np.mean(sel, dtype=np.float64)
    std = np.sqrt(np.sum((sel.astype(np.float64) - mean) ** 2) / (sel.size - 1))
    mean, std = F(mean), F(std)
    norm = (adv - mean) / np.maximum(std, F(1e-5))
    return np.clip(norm, -F(adv_clip), F(adv_clip)).astype(F), mean, std


def disc_reward(logits, scale):
    # amp_agent.py:194-206
    logits = np.asarray(logits, F)
    prob = F(1) / (F(1) + np.exp(-logits))
    return (-np.log(np.maximum(F(1) - prob, F(0.0001))) * F(scale)).astype(F)


# ---------------------------------------------------------------- model (torch CPU fp32)
class Model:
    """actor 264-1024-1024-512-29, critic ..-1, disc 114-1024-512-1, ReLU."""

    def __init__(self, params, action_std=0.05):
        self.p = {k: torch.tensor(np.asarray(v, F)).to(_DTYPE).requires_grad_(True) for k, v in params.items()}
        # distribution_gaussian_diag.py:24-31, 63-67: fixed fp32 logstd vector, std = exp(logstd).  A parameter set that holds LOGSTD_KEY is
        # the CONSTANT std type (:32-37): the log-std is one more trainable tensor, registered before the mean head
        fixed = torch.full((29,), float(np.log(action_std)), dtype=torch.float32)
        self._std_fixed = torch.exp(fixed).to(_DTYPE)
        self._logstd_fixed = fixed.to(_DTYPE)

    @property
    def logstd(self):
        return self.p[LOGSTD_KEY] if LOGSTD_KEY in self.p else self._logstd_fixed

    @property
    def std(self):
        return torch.exp(self.p[LOGSTD_KEY]) if LOGSTD_KEY in self.p else self._std_fixed

    def dist(self, x):
        """(mean, logstd, std) of the action distribution for normalised observations x (DistributionGaussianDiagBuilder.forward,
        distribution_gaussian_diag.py:47-58): logstd / std are per row for the VARIABLE type, vectors otherwise."""
        h = self._mlp(x, "_model._actor_layers")
        mean = torch.nn.functional.linear(h, self.p["_model._action_dist._mean_net.weight"], self.p["_model._action_dist._mean_net.bias"])
        if LOGSTD_W in self.p:
            ls = torch.nn.functional.linear(h, self.p[LOGSTD_W], self.p[LOGSTD_B])
            return mean, ls, torch.exp(ls)
        return mean, self.logstd, self.std

    def names(self):
        return list(self.p)  # (registration order: the order of PARAM_SHAPES / param_shapes())

    def _mlp(self, x, prefix, idxs=None):
        # (every hidden layer the parameter set holds for this stack: Sequential indices 0, 2, 4, ...)
        for i in sorted(int(k[len(prefix) + 1:].split(".")[0]) for k in self.p if k.startswith(prefix + ".") and k.endswith(".weight")):
            x = torch.relu(torch.nn.functional.linear(x, self.p[f"{prefix}.{i}.weight"], self.p[f"{prefix}.{i}.bias"]))
        return x

    def actor_mean(self, x):
        h = self._mlp(x, "_model._actor_layers", (0, 2, 4))
        return torch.nn.functional.linear(h, self.p["_model._action_dist._mean_net.weight"], self.p["_model._action_dist._mean_net.bias"])

    def critic(self, x):
        h = self._mlp(x, "_model._critic_layers", (0, 2, 4))
        return torch.nn.functional.linear(h, self.p["_model._critic_out.weight"], self.p["_model._critic_out.bias"]).squeeze(-1)

    def disc(self, x):
        h = self._mlp(x, "_model._disc_layers", (0, 2))
        return torch.nn.functional.linear(h, self.p["_model._disc_logits.weight"], self.p["_model._disc_logits.bias"]).squeeze(-1)

    def log_prob(self, mean, a, logstd=None, std=None):
        # distribution_gaussian_diag.py:90-94
        d = mean.shape[-1]
        logstd, std = (self.logstd, self.std) if logstd is None else (logstd, std)
        logp = -0.5 * torch.sum(torch.square((a - mean) / std), dim=-1)
        logp = logp + (-0.5 * d * np.log(2.0 * np.pi) - torch.sum(torch.broadcast_to(logstd, mean.shape), dim=-1))
        return logp


_DTYPE = torch.float32


class float64_mode:
    """with float64_mode(): Model(...) / compute_loss(...) run the same fp32 inputs in double precision -- the exact-arithmetic
    reference against which fp32 rounding (the HIP kernels' and torch-CPU's alike) is measured."""

    def __enter__(self):
        global _DTYPE
        _DTYPE = torch.float64

    def __exit__(self, *a):
        global _DTYPE
        _DTYPE = torch.float32


def t32(x):
    return torch.tensor(np.asarray(x, F)).to(_DTYPE)


def actor_step(model, obs_norm, a_norm, obs, noise, rand_mask=None):
    """ppo_agent.py:72-104 (TRAIN mode) given the N(0,1) draw: returns action, logp, norm_a."""
    with torch.no_grad():
        mean, ls, std = model.dist(t32(obs_norm.normalize(obs)))
        norm_a = mean + std * t32(noise)
        if rand_mask is not None:
            norm_a = torch.where(t32(rand_mask)[:, None] == 1.0, norm_a, mean)
        logp = model.log_prob(mean, norm_a, ls, std)
    a = a_norm.unnormalize(norm_a.numpy())
    return a, logp.numpy(), norm_a.numpy()


class LossCfg:
    # configs/agent/add_g1.yaml:15-38
    def __init__(self, **kw):
        self.ppo_clip_ratio = 0.2
        self.action_bound_weight = 10.0
        self.action_entropy_weight = 0.0
        self.action_reg_weight = 0.0
        self.critic_loss_weight = 1.0
        self.disc_loss_weight = 0.5
        self.disc_logit_reg = 0.01
        self.disc_grad_penalty = 20.0
        self.disc_weight_decay = 0.0001
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def compute_loss(model, cfg, batch):
    """amp_agent.py:98-114 + ppo_agent.py:194-275 + add_agent.py:141-202.
    batch: norm_obs, norm_action, a_logp, adv, tar_val, rand_action_mask, norm_diff (numpy)."""
    info = {}
    norm_obs = t32(batch["norm_obs"])
    # critic (ppo_agent.py:209-219)
    pred = model.critic(norm_obs)
    critic_loss = torch.mean(torch.square(t32(batch["tar_val"]) - pred))
    # actor (ppo_agent.py:221-275); only samples with random actions
    m = t32(batch["rand_action_mask"]) == 1.0
    mean, ls, std = model.dist(norm_obs[m])
    logp = model.log_prob(mean, t32(batch["norm_action"])[m], ls, std)
    ratio = torch.exp(logp - t32(batch["a_logp"])[m])
    adv = t32(batch["adv"])[m]
    l0 = adv * ratio
    l1 = adv * torch.clamp(ratio, 1.0 - cfg.ppo_clip_ratio, 1.0 + cfg.ppo_clip_ratio)
    actor_loss = -torch.mean(torch.minimum(l0, l1))
    info["clip_frac"] = torch.mean((torch.abs(ratio - 1.0) > cfg.ppo_clip_ratio).float()).item()
    info["imp_ratio"] = torch.mean(ratio).item()
    # base_agent.py:522-546 (bounds are finite for the G1)
    vmin = torch.clamp_max(mean + 1.0, 0.0)
    vmax = torch.clamp_min(mean - 1.0, 0.0)
    bound = torch.mean(torch.sum(torch.square(vmin), dim=-1) + torch.sum(torch.square(vmax), dim=-1))
    actor_loss = actor_loss + cfg.action_bound_weight * bound
    info["action_bound_loss"] = bound.item()
    if cfg.action_entropy_weight != 0:  # ppo_agent.py:262-266, distribution_gaussian_diag.py:96-99
        ent = torch.sum(torch.broadcast_to(ls, mean.shape), dim=-1) + 0.5 * mean.shape[-1] * np.log(2.0 * np.pi * np.e)
        actor_loss = actor_loss - cfg.action_entropy_weight * torch.mean(ent)
        info["action_entropy"] = torch.mean(ent).item()
    if cfg.action_reg_weight != 0:      # ppo_agent.py:268-272, distribution_gaussian_diag.py:113-116
        reg = torch.mean(torch.sum(torch.square(mean), dim=-1))
        actor_loss = actor_loss + cfg.action_reg_weight * reg
        info["action_reg_loss"] = reg.item()
    # discriminator (add_agent.py:141-202)
    nd = t32(batch["norm_diff"]).requires_grad_(True)
    pos_logit = model.disc(torch.zeros(1, nd.shape[1], dtype=nd.dtype))
    neg_logit = model.disc(nd)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    loss_pos = bce(pos_logit, torch.full_like(pos_logit, 0.9))  # amp_agent.py:182-185
    loss_neg = bce(neg_logit, torch.full_like(neg_logit, 0.1))  # amp_agent.py:177-180
    disc_loss = 0.5 * (loss_pos + loss_neg)
    w_logit = model.p["_model._disc_logits.weight"]
    logit_loss = torch.sum(torch.square(w_logit))
    disc_loss = disc_loss + cfg.disc_logit_reg * logit_loss
    grad = torch.autograd.grad(neg_logit, nd, grad_outputs=torch.ones_like(neg_logit), create_graph=True)[0]
    gnorm = torch.sqrt(torch.sum(torch.square(grad), dim=-1) + 1e-8)
    gp = torch.mean(torch.square(gnorm - 1))
    disc_loss = disc_loss + cfg.disc_grad_penalty * gp
    if cfg.disc_weight_decay != 0:
        ws = [v for k, v in model.p.items() if k.startswith("_model._disc_layers.") and k.endswith(".weight")] + [w_logit]  # add_agent.py:181-186
        disc_loss = disc_loss + cfg.disc_weight_decay * sum(torch.sum(torch.square(w)) for w in ws)
    loss = actor_loss + cfg.critic_loss_weight * critic_loss + cfg.disc_loss_weight * disc_loss
    info.update(
        loss=loss.item(), actor_loss=actor_loss.item(), critic_loss=critic_loss.item(), disc_loss=disc_loss.item(),
        disc_grad_penalty=gp.item(), disc_logit_loss=logit_loss.item(),
        disc_pos_acc=torch.mean((pos_logit > 0).float()).item(), disc_neg_acc=torch.mean((neg_logit < 0).float()).item(),
        disc_pos_logit=torch.mean(pos_logit).item(), disc_neg_logit=torch.mean(neg_logit).item(),
    )
    return loss, info


class AdamW:
    """torch.optim.AdamW(lr, betas=(0.9,0.999), eps=1e-8, weight_decay) as MPOptimizer builds
    it (mp_optimizer.py:14-40; grad clipping is off)."""

    def __init__(self, model, lr=1e-4, weight_decay=0.0):
        self.model, self.lr, self.wd = model, lr, weight_decay
        self.b1, self.b2, self.eps = 0.9, 0.999, 1e-8
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in model.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in model.p.items()}

    def step(self, loss):
        names = self.model.names()
        grads = torch.autograd.grad(loss, [self.model.p[n] for n in names])
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2_sqrt = math.sqrt(1 - self.b2 ** self.t)
        with torch.no_grad():
            for n, g in zip(names, grads):
                p = self.model.p[n]
                p.mul_(1 - self.lr * self.wd)
                self.m[n].lerp_(g, 1 - self.b1)
                self.v[n].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (self.v[n].sqrt() / bc2_sqrt).add_(self.eps)
                p.addcdiv_(self.m[n], denom, value=-(self.lr / bc1))
        return {n: g.numpy() for n, g in zip(names, grads)}


class SampleStream:
    """experience_buffer.py:92-113: minibatch indices are consecutive slices of a permutation
    that is redrawn (a fresh randperm) when it runs out; `perms` supplies the draws."""

    def __init__(self, perms):
        self.perms = [np.asarray(p, np.int64) for p in perms]
        self.k = 0
        self.buf = self.perms[0]
        self.head = 0

    def reset(self):
        self.k += 1
        self.buf = self.perms[self.k]
        self.head = 0

    def sample(self, n, sample_count):
        L = self.buf.shape[0]
        if self.head + n <= L:
            idx = self.buf[self.head:self.head + n]
            self.head += n
        else:
            idx0 = self.buf[self.head:]
            rem = n - (L - self.head)
            self.reset()
            idx = np.concatenate([idx0, self.buf[:rem]])
            self.head = rem
        return np.remainder(idx, sample_count)
