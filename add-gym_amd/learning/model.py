"""Parameters of the actor / critic / discriminator MLPs in ONE flat fp32 device buffer
(+ same-layout gradient and AdamW moment buffers), with the call plans that run their forward
and backward passes through libaddhip (fp32 MFMA GEMMs + fused epilogues).

Architecture = the reference's ADDModel for configs/agent/add_g1.yaml (ppo_model.py:36-59,
add_model.py:29-46, nets/fc_3layers_1024units.py, nets/fc_2layers_1024units.py):
  actor  obs -> 1024 -> 1024 -> 512 -> 29 (ReLU; fixed logstd)      critic obs -> 1024 -> 1024 -> 512 -> 1
  disc   disc_obs -> 1024 -> 512 -> 1
Device layout differences (all invisible in checkpoints): rows of the 29-wide actor head are padded
to 32, input columns to the row strides of the obs buffers (264 -> 272, 114 -> 128), 1-wide heads are stored as vectors.
"""
import math

import torch

from .. import _lib as L
from ..hotpath import gemm

NET_SIZES = {"fc_3layers_1024units": [1024, 1024, 512], "fc_2layers_1024units": [1024, 512],
             "fc_2layers_512units": [512, 256], "fc_2layers_256units": [256, 128], "fc_2layers_128units": [128, 64],
             "fc_2layers_64units": [64, 32]}  # nets/*.py layer_sizes


def build_net_sizes(name):
    """nets/net_builder.py:5-11: networks are chosen by module name."""
    if name not in NET_SIZES:
        raise ValueError(f"Unsupported net: {name}")
    return NET_SIZES[name]


class Mlp:
    """One network: hidden Linear+ReLU stack and a linear head, as views into the flat buffers."""

    def __init__(self, name, in_dim, in_ld, hidden, head_dim, heads=1):
        self.name, self.in_dim, self.in_ld, self.hidden, self.head_dim = name, in_dim, in_ld, list(hidden), head_dim
        # 29-wide head padded to 32 rows; 1-wide head = vector; heads = 2 (actor_std_type VARIABLE): the log-std head's 32 rows behind the mean head's
        self.head_rows = 32 * heads if head_dim > 1 else 1
        self.specs = []  # (key, shape)
        prev = in_ld
        for i, h in enumerate(hidden):
            self.specs += [(f"W{i}", (h, prev)), (f"b{i}", (h,))]
            prev = h
        self.specs += [("Wh", (self.head_rows, prev)), ("bh", (max(4, self.head_rows),))]


class Model:
    def __init__(self, model_cfg, obs_dim, obs_ld, disc_dim, disc_ld, device, seed=0):
        self.device = device
        self.std_type = str(model_cfg.get("actor_std_type", "FIXED"))
        self.actor = Mlp("actor", obs_dim, obs_ld, build_net_sizes(model_cfg["actor_net"]), L.NUM_DOF, heads=2 if self.std_type == "VARIABLE" else 1)
        self.critic = Mlp("critic", obs_dim, obs_ld, build_net_sizes(model_cfg["critic_net"]), 1)
        self.disc = Mlp("disc", disc_dim, disc_ld, build_net_sizes(model_cfg["disc_net"]), 1)
        self.nets = [self.actor, self.critic, self.disc]
        self.action_std = float(model_cfg["action_std"])
        # distribution_gaussian_diag.py:19-45.  FIXED: the log-std is a constant (a scalar argument of the kernels).  CONSTANT: one trainable
        # log-std per action dimension -- 32 floats behind the actor head's bias in the flat buffers (gradient, AdamW state, the actor's
        # exchange bucket all follow from that), from which addhip_dist_refresh derives the kernels' `dist` vector after every change.
        # VARIABLE: the log-std is a second linear head on the actor's last layer -- rows 32..60 of a 64-row head matrix (bias likewise), so that the
        # two heads are ONE 64-wide product everywhere (rollout head GEMM, weight gradient, dz); the kernels read the per-sample log-std columns
        # behind the mean's (addhip_actor_sample / addhip_actor_loss: logstd_rows)
        if self.std_type not in ("FIXED", "CONSTANT", "VARIABLE"):
            raise ValueError("actor_std_type must be FIXED, CONSTANT or VARIABLE")
        if self.std_type == "CONSTANT":
            self.actor.specs.append(("logstd", (32,)))
        self.init_output_scale = float(model_cfg["actor_init_output_scale"])
        # Flat layout, chosen for the data-parallel exchange (each bucket one contiguous range, in the order the gradients
        # become final during a backward pass): [actor W1.. head][critic W1.. head][discriminator][actor W0 b0][critic W0 b0]
        off = 0
        self.offsets = {}

        def place(net, keys):
            nonlocal off
            first = off
            for key, shape in net.specs:
                if key in keys:
                    self.offsets[(net.name, key)] = (off, shape)
                    off += (math.prod(shape) + 7) // 8 * 8  # 32-byte steps: every tensor is 16-byte aligned in the bf16 shadow too
            return first, off

        first_layer = ("W0", "b0")
        rest = lambda net: [k for k, _ in net.specs if k not in first_layer]
        self.bucket_ranges = {"actor_tail": place(self.actor, rest(self.actor)), "critic_tail": place(self.critic, rest(self.critic)),
                              "disc": place(self.disc, [k for k, _ in self.disc.specs])}
        a0 = place(self.actor, first_layer)
        c0 = place(self.critic, first_layer)
        self.bucket_ranges["first_layers"] = (a0[0], c0[1])
        self.count = off
        self.params = torch.zeros(off, device=device)
        self.grads = torch.zeros(off, device=device)
        self.exp_avg = torch.zeros(off, device=device)
        self.exp_avg_sq = torch.zeros(off, device=device)
        self.opt_step = 0
        self.params16 = None  # bf16 shadow of params (agent.matmul_precision = bf16), refreshed after every optimiser step
        self.w_amax = None    # tracked maximum |parameter| (agent.matmul_precision = f16x2: the scale of every weight operand), likewise
        self.dist = torch.zeros(L.DIST_FLOATS, device=device) if self.std_type == "CONSTANT" else None
        self.logstd_ones = torch.zeros(32, device=device)  # (d entropy / d logstd: ones on the 29 action dimensions)
        self.logstd_ones[:L.NUM_DOF] = 1.0
        self._init_params(seed)
        self.refresh_dist()
        # distribution_gaussian_diag.py:24-31, 63-94: fp32 logstd vector -> std and the log-prob constant
        logstd = torch.full((L.NUM_DOF,), float(math.log(self.action_std)), dtype=torch.float32)
        self.std32 = float(torch.exp(logstd)[0].item())
        self.logp_const = float((-0.5 * L.NUM_DOF * math.log(2.0 * math.pi) - torch.sum(logstd)).item())
        self.entropy = float((torch.sum(logstd) + 0.5 * L.NUM_DOF * math.log(2.0 * math.pi * math.e)).item())  # :96-99

    # ---- views
    def view(self, net, key, buf=None):
        off, shape = self.offsets[(net, key)]
        buf = self.params if buf is None else buf
        return buf[off:off + math.prod(shape)].view(shape)

    def p(self, net, key, buf=None):
        off, _ = self.offsets[(net, key)]
        base = (self.params if buf is None else buf).data_ptr()
        return base + 4 * off

    def g(self, net, key):
        return self.p(net, key, self.grads)

    def enable_shadow(self, planes=L.STORE_BF16):
        """16-bit shadows of the parameters: params16 in the flat layout (W[out,in]: what the forward and weight-gradient GEMMs read) and
        params16t holding W^T[in,out] of every hidden weight past a net's first layer (what the backward dX GEMMs and the gradient-
        penalty chain read, k-contiguously).  planes = L.STORE_BF16: one bf16 per value (rounded); L.STORE_BF16X3: plane storage, three
        bf16 per value whose sum is the fp32 value exactly (include/addhip.h, "plane storage")."""
        self.shadow_planes = int(planes)
        self.params16 = torch.zeros(self.shadow_planes * self.count, dtype=torch.bfloat16, device=self.device)
        self.params16t = torch.zeros(self.shadow_planes * self.count, dtype=torch.bfloat16, device=self.device)
        self._transposed = [(net.name, f"W{i}") for net in self.nets for i in range(len(net.hidden)) if i > 0 or net is self.disc]
        import ctypes as C
        n = len(self._transposed)
        offs, dims = zip(*(self.offsets[k] for k in self._transposed))
        self._shadow_table = ((C.c_int64 * n)(*offs), (C.c_int32 * n)(*(d[0] for d in dims)), (C.c_int32 * n)(*(d[1] for d in dims)), n)
        self._net_tables = {}
        for net in self.nets:  # the same table net by net: each net's transposed copies are refreshed on its own stream (transposed_refresh_args)
            keys = [k for k in self._transposed if k[0] == net.name]
            o, dm = zip(*(self.offsets[k] for k in keys))
            self._net_tables[net.name] = ((C.c_int64 * len(keys))(*o), (C.c_int32 * len(keys))(*(d[0] for d in dm)), (C.c_int32 * len(keys))(*(d[1] for d in dm)), len(keys))
        self.refresh_shadow()

    def enable_w_amax(self):
        """ADDHIP_PREC_F16X2: ADDHIP_AMAX_SLOTS slots bounding max |parameter| over the flat buffer (one bound for every weight: fp16 spans 2^18
        above the point where the split starts to lose relative precision, far more than the layers' maxima differ by)."""
        self.w_amax = torch.zeros(L.AMAX_SLOTS, dtype=torch.int32, device=self.device)
        self.refresh_shadow()

    def refresh_w_amax(self, stream):
        if self.w_amax is not None:
            L.call("addhip_fill_zero", L.ptr(self.w_amax), L.AMAX_SLOTS, stream)
            L.call("addhip_amax_f32", L.ptr(self.params), self.count, L.ptr(self.w_amax), stream)

    def refresh_dist(self, stream=None):
        """actor_std_type CONSTANT: std / log-probability constant / entropy from the current log-std (one launch, in stream order)."""
        if self.dist is not None:
            L.call("addhip_dist_refresh", self.p("actor", "logstd"), L.ptr(self.dist), L.current_stream() if stream is None else stream)

    def dist_ptr(self):
        return None if self.dist is None else L.ptr(self.dist)

    def refresh_shadow(self, stream=None):
        """Derived parameter state after the parameters changed (optimiser step, load, broadcast): params16 = bf16(params), round to nearest
        even, + the transposed copies in one launch (storage modes); the tracked maximum (f16x2)."""
        st = L.current_stream() if stream is None else stream
        self.refresh_w_amax(st)
        self.refresh_dist(st)
        if self.params16 is not None:
            offs, rows, cols, n = self._shadow_table
            L.call("addhip_shadow_refresh", L.ptr(self.params), L.ptr(self.params16), L.ptr(self.params16t), self.count, offs, rows, cols, n, self.shadow_planes, st)

    def transposed_refresh_args(self, net):
        """Arguments (without the stream) of the addhip_shadow_refresh call that rewrites only `net`'s transposed weight copies from the
        fp32 parameters (the flat shadow is written by the optimiser step itself)."""
        offs, rows, cols, n = self._net_tables[net]
        return (L.ptr(self.params), None, L.ptr(self.params16t), self.count, offs, rows, cols, n, self.shadow_planes)

    def p16(self, net, key):
        off, _ = self.offsets[(net, key)]
        return self.params16.data_ptr() + 2 * self.shadow_planes * off

    def p16t(self, net, key):
        """W^T [in, out] (leading dimension = out)."""
        assert (net, key) in self._transposed
        off, _ = self.offsets[(net, key)]
        return self.params16t.data_ptr() + 2 * self.shadow_planes * off

    def n_elem(self, net, key):
        return math.prod(self.offsets[(net, key)][1])

    # ---- init: nn.Linear default weights, zero biases, special heads
    def _init_params(self, seed):
        gen = torch.Generator().manual_seed(seed)

        def uniform(shape, bound):
            return (torch.rand(shape, generator=gen) * 2 - 1) * bound

        for net in self.nets:
            for i, h in enumerate(net.hidden):
                fan_in = net.in_dim if i == 0 else net.hidden[i - 1]
                w = self.view(net.name, f"W{i}")
                w.zero_()
                w[:, :fan_in] = uniform((h, fan_in), 1.0 / math.sqrt(fan_in)).to(self.device)
            fan_in = net.hidden[-1]
            bound = {"actor": self.init_output_scale, "critic": 1.0 / math.sqrt(fan_in), "disc": 1.0}[net.name]
            wh = self.view(net.name, "Wh")
            wh.zero_()
            wh[:net.head_dim] = uniform((net.head_dim, fan_in), bound).to(self.device)
            if net.head_rows == 64:  # distribution_gaussian_diag.py:38-43: the log-std head, same scale, bias log(action_std)
                wh[32:32 + net.head_dim] = uniform((net.head_dim, fan_in), bound).to(self.device)
                self.view(net.name, "bh")[32:32 + net.head_dim] = float(math.log(self.action_std))
        if self.std_type == "CONSTANT":  # distribution_gaussian_diag.py:25, 32-37
            ls = self.view("actor", "logstd")
            ls.zero_()
            ls[:L.NUM_DOF] = float(math.log(self.action_std))

    # ---- reference checkpoint keys <-> flat buffers
    def _key_map(self):
        m = []
        for net, prefix, head in ((self.actor, "_model._actor_layers", "_model._action_dist._mean_net"),
                                  (self.critic, "_model._critic_layers", "_model._critic_out"),
                                  (self.disc, "_model._disc_layers", "_model._disc_logits")):
            for i in range(len(net.hidden)):
                m.append((f"{prefix}.{2 * i}.weight", net, f"W{i}"))
                m.append((f"{prefix}.{2 * i}.bias", net, f"b{i}"))
            if net is self.actor and self.std_type == "CONSTANT":  # (registered before the mean head: distribution_gaussian_diag.py:19-37)
                m.append(("_model._action_dist._logstd_net", net, "logstd"))
            m.append((f"{head}.weight", net, "Wh"))
            m.append((f"{head}.bias", net, "bh"))
            if net is self.actor and self.std_type == "VARIABLE":  # (a sub-module created after the mean head: registered behind it)
                m.append(("_model._action_dist._logstd_net.weight", net, "Wh@32"))
                m.append(("_model._action_dist._logstd_net.bias", net, "bh@32"))
        return m

    def _block(self, net, key, buf=None):
        """The rows of a flat tensor a reference key maps to: `key@r` = rows r.. of `key` (the second head of a two-head matrix), and the first
        head's block of such a matrix ends where the second begins."""
        base, _, r0 = key.partition("@")
        v = self.view(net.name, base, buf)
        if r0:
            return v[int(r0):]
        return v[:32] if (net.head_rows == 64 and base in ("Wh", "bh")) else v

    def _ref_shape(self, net, key):
        key = key.partition("@")[0]
        if key.startswith("W") and key != "Wh":
            i = int(key[1:])
            return (net.hidden[i], net.in_dim if i == 0 else net.hidden[i - 1])
        if key.startswith("b") and key != "bh":
            return (net.hidden[int(key[1:])],)
        if key == "logstd":
            return (L.NUM_DOF,)
        return (net.head_dim, net.hidden[-1]) if key == "Wh" else (net.head_dim,)

    def export(self, buf=None):
        """{reference key: cpu tensor of the reference shape} from params (or another flat buffer)."""
        out = {}
        for name, net, key in self._key_map():
            v = self._block(net, key, buf)
            shape = self._ref_shape(net, key)
            if v.dim() == 2:
                v = v[:shape[0], :shape[1]]
            else:
                v = v[:shape[0]]
            if name == "_model._action_dist._mean_net.weight" and self.std_type == "FIXED":  # the reference's registration order: logstd precedes the mean head
                out["_model._action_dist._logstd_net"] = torch.full((L.NUM_DOF,), float(math.log(self.action_std)), dtype=torch.float32)
            out[name] = v.detach().clone().cpu()
        return out

    def load(self, state, buf=None):
        for name, net, key in self._key_map():
            if name not in state:
                raise KeyError(f"checkpoint is missing {name}")
            src = state[name].to(torch.float32)
            v = self._block(net, key, buf)
            v.zero_()
            if v.dim() == 2:
                v[:src.shape[0], :src.shape[1]] = src.to(self.device)
            else:
                v[:src.shape[0]] = src.to(self.device)
        if buf is None:
            self.refresh_shadow()

    def num_params(self):
        return sum(math.prod(self._ref_shape(net, key)) for _, net, key in self._key_map())


def split_k_for(out_dim, in_dim, rows):
    """K slices of a weight-gradient GEMM (the rule of csrc/learner.hip, restated to size the split-K scratch)."""
    tiles = ((out_dim + 127) // 128) * ((in_dim + 127) // 128)
    s = max(1, min(32, -(-512 // tiles)))
    while s > 1 and rows // s < 256:
        s //= 2
    return s


class Plan:
    """A recorded sequence of launches held by libaddhip (addhip_plan_t; include/addhip.h, "recorded plans").  add() calls an entry point
    in recording mode: its arguments are checked, its parameter blocks copied, and its launch -- or, for a composite entry point such as
    addhip_mlp_forward, each of its launches -- is appended to the plan.  len(plan) counts launches; run() replays a range with ONE C call."""

    def __init__(self):
        import ctypes as C

        self._lib = L.load()
        self.keep = []
        self._handle = C.c_void_p()
        self._check(self._lib.addhip_plan_create(C.byref(self._handle)), "addhip_plan_create")

    def _check(self, rc, what):
        if rc != 0:
            raise L.AddhipError(f"{what} failed ({rc}): {self._lib.addhip_last_error().decode()}")

    def add(self, name, *args):
        """Record one entry-point call (without its stream argument); returns the plan's length before it."""
        lib, at = self._lib, len(self)
        self._check(lib.addhip_plan_record_begin(self._handle), "addhip_plan_record_begin")
        try:
            rc = getattr(lib, name)(*args, None)  # recorded, not launched
            msg = lib.addhip_last_error().decode() if rc != 0 else ""
        finally:
            self._check(lib.addhip_plan_record_end(self._handle), "addhip_plan_record_end")
        if rc != 0:
            raise L.AddhipError(f"{name} failed ({rc}) while being recorded: {msg}")
        return at

    def hold(self, *objs):
        """Keep Python objects alive with the plan (host tables a recorded block points at are copied by the library; this is for the rest)."""
        self.keep.extend(objs)

    def __len__(self):
        return self._lib.addhip_plan_size(self._handle)

    @property
    def handle(self):
        return self._handle

    def launches(self):
        """[(entry point name, [addhip_gemm_t, ...])] per recorded launch: what tools and the bench's roofline accounting read."""
        lib, out = self._lib, []
        buf = (L.GemmT * L.GEMM_MAX_GROUP)()
        for i in range(len(self)):
            k = lib.addhip_plan_call_gemms(self._handle, i, buf, L.GEMM_MAX_GROUP)
            gemms = []
            for j in range(max(k, 0)):
                g = L.GemmT()
                import ctypes as C
                C.memmove(C.byref(g), C.byref(buf[j]), C.sizeof(L.GemmT))
                gemms.append(g)
            out.append((lib.addhip_plan_call_name(self._handle, i).decode(), gemms))
        return out

    def run(self, stream, first=0, last=None):
        rc = self._lib.addhip_plan_run(self._handle, first, -1 if last is None else last, stream)
        if rc != 0:
            raise L.AddhipError(f"addhip_plan_run failed ({rc}): {self._lib.addhip_last_error().decode()}")

    def __del__(self):
        if getattr(self, "_handle", None) is not None and self._handle:
            self._lib.addhip_plan_destroy(self._handle)
            self._handle = None


class Schedule:
    """addhip_schedule_t over a Plan: `sections` is an array of addhip_section_t (e.g. from addhip_update_schedule), issued in list order;
    bucket k of a section is reported to run()'s call-back as buckets[k]."""

    def __init__(self, plan, sections, num_streams, buckets=()):
        import ctypes as C

        self._lib, self._plan = L.load(), plan
        self.sections = [(s.stream, s.first, s.last, s.wait_before, s.wait_after, s.bucket) for s in sections]
        self.buckets = list(buckets)
        self._bucket_stream = {s.bucket: s.stream for s in sections if s.bucket >= 0}
        assert all(0 <= k < len(self.buckets) for k in self._bucket_stream)
        h = C.c_void_p()
        rc = self._lib.addhip_schedule_create(plan.handle, sections, len(sections), num_streams, C.byref(h))
        if rc != 0:
            raise L.AddhipError(f"addhip_schedule_create failed ({rc}): {self._lib.addhip_last_error().decode()}")
        self._handle, self._num_streams = h, num_streams
        self._null_cb = C.cast(None, L.BUCKET_FN)

    def run(self, streams, on_bucket=None):
        """streams: raw hipStream_t handles (ints); on_bucket(bucket, stream index) is called, inside this call, at the point of the
        issue order where that section's bucket is final on that stream."""
        import ctypes as C

        arr = (C.c_void_p * self._num_streams)(*streams)
        cb = self._null_cb
        err = []
        if on_bucket is not None:
            def _cb(user, k, stream):
                try:
                    on_bucket(self.buckets[k], self._bucket_stream[k])
                except BaseException as e:  # must not propagate through the C frame
                    err.append(e)
            cb = L.BUCKET_FN(_cb)
        rc = self._lib.addhip_schedule_run(self._handle, arr, cb, None)
        if err:
            raise err[0]
        if rc != 0:
            raise L.AddhipError(f"addhip_schedule_run failed ({rc}): {self._lib.addhip_last_error().decode()}")

    def __del__(self):
        if getattr(self, "_handle", None) is not None and self._handle:
            self._lib.addhip_schedule_destroy(self._handle)
            self._handle = None


class NetRunner:
    """One Mlp's activation / gradient workspace for up to `rows` rows, as the addhip_mlp_t the library's composite entry points take
    (addhip_mlp_forward / _backward, addhip_ppo_loss_fwd_bwd, addhip_disc_loss_fwd_bwd: csrc/learner.hip assembles the launches)."""

    def __init__(self, model, net, rows, device, slabs, precision=L.PREC_F32, storage16=0, deterministic=False):
        self.m, self.net, self.rows, self.slabs = model, net, rows, slabs
        self.precision = precision  # ADDHIP_PREC_* of every GEMM this runner records (agent.matmul_precision)
        # storage16 (agent.matmul_precision = bf16): hidden activations and pre-activation gradients are kept as bf16 in HBM and
        # the GEMMs read the model's bf16 weight shadow; the last hidden layer and the top gradient also exist in fp32 for the
        # loss-head kernels
        # storage16 = L.STORE_BF16X3 (agent.matmul_precision = bf16x3): the same buffers in plane storage, three bf16 per value (exact)
        self.storage16 = int(storage16)  # 0 | L.STORE_BF16 | L.STORE_BF16X3
        if self.storage16:
            self.h16 = [torch.zeros(rows, self.storage16 * h, dtype=torch.bfloat16, device=device) for h in net.hidden]
            self.dz16 = [torch.zeros(rows, self.storage16 * h, dtype=torch.bfloat16, device=device) for h in net.hidden]
        self.aux_slabs = None   # (layer, buffer): the top layer's weight-gradient group uses its own split-K scratch, so that it may run on
                                # another stream than the rest of the backward pass (addhip_mlp_t.slabs_top)
        # ReLU sign bits of the hidden activations (1 bit per element): what the backward GEMMs read as their mask instead of
        # the fp32 activations themselves (64 MB -> 2 MB per 16384 x 1024 layer)
        self.hb = [torch.zeros(rows, (h + 31) // 32, dtype=torch.int32, device=device) for h in net.hidden]
        self.h = [torch.zeros(rows, h, device=device) for h in net.hidden]
        self.dz = [torch.zeros(rows, h, device=device) for h in net.hidden]
        # bias-gradient column sums of the dX GEMMs, spread over 16 rows (same-line float atomics of all row tiles serialise otherwise);
        # summed into the gradient and cleared by the split-K combine that follows (addhip_slab_reduce_pair)
        # agent.deterministic: one replica row per 32-row block instead (every slot then receives ONE add, and the combine sums the rows in
        # order) and a scratch for the fixed-order column sums of the loss heads -- no float atomics behind any gradient
        self.deterministic = bool(deterministic)
        rep_rows = (rows + 31) // 32 if self.deterministic else 16
        self.bias_rep = torch.zeros(rep_rows, max(net.hidden), device=device) if slabs is not None else None
        self.ordered = torch.zeros(L.HEAD_BWD_BLOCKS * (2 * max(net.hidden) + 4), device=device) if self.deterministic else None
        # agent.matmul_precision = f16x2: the tracked maxima of this runner's activations / gradients (include/addhip.h: ADDHIP_MLP_AMAX_*)
        self.amax = torch.zeros(L.MLP_AMAX_TENSORS * L.AMAX_SLOTS, dtype=torch.int32, device=device) if precision == L.PREC_F16X2 else None

    def c_struct(self):
        """addhip_mlp_t over this runner's buffers (include/addhip.h, "composite entry points")."""
        net, m = self.net, self.m
        n = len(net.hidden)
        c = L.MlpT()
        c.num_hidden, c.in_dim, c.in_ld, c.head_rows, c.precision, c.rows_cap = n, net.in_dim, net.in_ld, net.head_rows, self.precision, self.rows
        c.storage = self.storage16
        for i, h in enumerate(net.hidden):
            c.hidden[i] = h
            c.W[i], c.b[i], c.gW[i], c.gb[i] = m.p(net.name, f"W{i}"), m.p(net.name, f"b{i}"), m.g(net.name, f"W{i}"), m.g(net.name, f"b{i}")
            c.h[i], c.dz[i], c.hbits[i] = L.ptr(self.h[i]), L.ptr(self.dz[i]), L.ptr(self.hb[i])
            if self.storage16:
                c.W16[i], c.h16[i], c.dz16[i] = m.p16(net.name, f"W{i}"), L.ptr(self.h16[i]), L.ptr(self.dz16[i])
                if (net.name, f"W{i}") in m._transposed:
                    c.W16t[i] = m.p16t(net.name, f"W{i}")
        c.Wh, c.bh, c.gWh, c.gbh = m.p(net.name, "Wh"), m.p(net.name, "bh"), m.g(net.name, "Wh"), m.g(net.name, "bh")
        if self.slabs is not None:
            c.slabs, c.slab_floats = L.ptr(self.slabs), self.slabs.numel()
        if self.aux_slabs is not None:
            assert self.aux_slabs[0] == n - 1
            c.slabs_top = L.ptr(self.aux_slabs[1])
        if self.bias_rep is not None:
            c.bias_replicas, c.bias_replica_rows = L.ptr(self.bias_rep), self.bias_rep.shape[0]
        if self.amax is not None:
            c.amax, c.w_amax = L.ptr(self.amax), L.ptr(m.w_amax)
        if self.deterministic:
            c.deterministic, c.ordered_scratch = 1, L.ptr(self.ordered)
        if self.storage16:
            offs, rows, cols, cnt = m._net_tables[net.name]
            c.flat_params, c.flat_trans16, c.flat_count = L.ptr(m.params), L.ptr(m.params16t), m.count
            import ctypes as C
            c.t_offset, c.t_rows, c.t_cols, c.t_count = C.cast(offs, C.POINTER(C.c_int64)), C.cast(rows, C.POINTER(C.c_int32)), C.cast(cols, C.POINTER(C.c_int32)), cnt
        self._c = c
        return c

    def forward(self, plan, x_ptr, rows, a_mean=None, a_std=None, sign_bits=False, x16_ptr=None, x_amax=None):
        """Record this net's forward pass over `rows` rows into `plan` (addhip_mlp_forward: one GEMM with fused bias + ReLU per layer).
        sign_bits: also write the ReLU sign bits (a backward pass follows).  x16_ptr: bf16 copy of the input rows (storage16 runners)."""
        if not hasattr(self, "_c"):
            self.c_struct()
        return plan.add("addhip_mlp_forward", self._c, x_ptr, x16_ptr, rows, a_mean, a_std, int(bool(sign_bits)), x_amax)
