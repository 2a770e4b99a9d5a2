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

    def __init__(self, name, in_dim, in_ld, hidden, head_dim):
        self.name, self.in_dim, self.in_ld, self.hidden, self.head_dim = name, in_dim, in_ld, list(hidden), head_dim
        self.head_rows = 32 if head_dim > 1 else 1  # 29-wide head padded to 32 rows; 1-wide head = vector
        self.specs = []  # (key, shape)
        prev = in_ld
        for i, h in enumerate(hidden):
            self.specs += [(f"W{i}", (h, prev)), (f"b{i}", (h,))]
            prev = h
        self.specs += [("Wh", (self.head_rows, prev)), ("bh", (max(4, self.head_rows),))]


class Model:
    def __init__(self, model_cfg, obs_dim, obs_ld, disc_dim, disc_ld, device, seed=0):
        self.device = device
        self.actor = Mlp("actor", obs_dim, obs_ld, build_net_sizes(model_cfg["actor_net"]), L.NUM_DOF)
        self.critic = Mlp("critic", obs_dim, obs_ld, build_net_sizes(model_cfg["critic_net"]), 1)
        self.disc = Mlp("disc", disc_dim, disc_ld, build_net_sizes(model_cfg["disc_net"]), 1)
        self.nets = [self.actor, self.critic, self.disc]
        self.action_std = float(model_cfg["action_std"])
        if model_cfg.get("actor_std_type", "FIXED") != "FIXED":
            raise NotImplementedError("only actor_std_type FIXED (configs/agent/add_g1.yaml) is implemented")
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
        self._init_params(seed)
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

    def enable_shadow(self):
        """bf16 shadows of the parameters: params16 in the flat layout (W[out,in]: what the forward and weight-gradient GEMMs read) and
        params16t holding W^T[in,out] of every hidden weight past a net's first layer (what the backward dX GEMMs and the gradient-
        penalty chain read, k-contiguously)."""
        self.params16 = torch.zeros(self.count, dtype=torch.bfloat16, device=self.device)
        self.params16t = torch.zeros(self.count, dtype=torch.bfloat16, device=self.device)
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

    def refresh_shadow(self, stream=None):
        """params16 = bf16(params), round to nearest even, + the transposed copies: one launch."""
        if self.params16 is not None:
            st = L.current_stream() if stream is None else stream
            offs, rows, cols, n = self._shadow_table
            L.call("addhip_shadow_refresh", L.ptr(self.params), L.ptr(self.params16), L.ptr(self.params16t), self.count, offs, rows, cols, n, st)

    def transposed_refresh_args(self, net):
        """Arguments (without the stream) of the addhip_shadow_refresh call that rewrites only `net`'s transposed weight copies from the
        fp32 parameters (the flat shadow is written by the optimiser step itself)."""
        offs, rows, cols, n = self._net_tables[net]
        return (L.ptr(self.params), None, L.ptr(self.params16t), self.count, offs, rows, cols, n)

    def p16(self, net, key):
        off, _ = self.offsets[(net, key)]
        return self.params16.data_ptr() + 2 * off

    def p16t(self, net, key):
        """W^T [in, out] (leading dimension = out)."""
        assert (net, key) in self._transposed
        off, _ = self.offsets[(net, key)]
        return self.params16t.data_ptr() + 2 * off

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

    # ---- reference checkpoint keys <-> flat buffers
    def _key_map(self):
        m = []
        for net, prefix, head in ((self.actor, "_model._actor_layers", "_model._action_dist._mean_net"),
                                  (self.critic, "_model._critic_layers", "_model._critic_out"),
                                  (self.disc, "_model._disc_layers", "_model._disc_logits")):
            for i in range(len(net.hidden)):
                m.append((f"{prefix}.{2 * i}.weight", net, f"W{i}"))
                m.append((f"{prefix}.{2 * i}.bias", net, f"b{i}"))
            m.append((f"{head}.weight", net, "Wh"))
            m.append((f"{head}.bias", net, "bh"))
        return m

    def _ref_shape(self, net, key):
        if key.startswith("W") and key != "Wh":
            i = int(key[1:])
            return (net.hidden[i], net.in_dim if i == 0 else net.hidden[i - 1])
        if key.startswith("b") and key != "bh":
            return (net.hidden[int(key[1:])],)
        return (net.head_dim, net.hidden[-1]) if key == "Wh" else (net.head_dim,)

    def export(self, buf=None):
        """{reference key: cpu tensor of the reference shape} from params (or another flat buffer)."""
        out = {}
        for name, net, key in self._key_map():
            v = self.view(net.name, key, buf)
            shape = self._ref_shape(net, key)
            if v.dim() == 2:
                v = v[:shape[0], :shape[1]]
            else:
                v = v[:shape[0]]
            if name == "_model._action_dist._mean_net.weight":  # the reference's registration order: logstd precedes the mean head
                out["_model._action_dist._logstd_net"] = torch.full((L.NUM_DOF,), float(math.log(self.action_std)), dtype=torch.float32)
            out[name] = v.detach().clone().cpu()
        return out

    def load(self, state, buf=None):
        for name, net, key in self._key_map():
            if name not in state:
                raise KeyError(f"checkpoint is missing {name}")
            src = state[name].to(torch.float32)
            v = self.view(net.name, key, buf)
            v.zero_()
            if v.dim() == 2:
                v[:src.shape[0], :src.shape[1]] = src.to(self.device)
            else:
                v[:src.shape[0]] = src.to(self.device)

    def num_params(self):
        return sum(math.prod(self._ref_shape(net, key)) for _, net, key in self._key_map())


def split_k_for(out_dim, in_dim, rows):
    tiles = ((out_dim + 127) // 128) * ((in_dim + 127) // 128)
    s = max(1, min(32, -(-512 // tiles)))
    while s > 1 and rows // s < 256:
        s //= 2
    return s


class Plan:
    """A recorded sequence of C-ABI calls (argument tuples are prebuilt once; replay costs one
    ctypes call per kernel)."""

    def __init__(self):
        self.calls = []
        self.keep = []
        self._lib = L.load()

    def add(self, name, *args):
        self.calls.append((name, getattr(self._lib, name), args))

    def hold(self, *objs):
        self.keep.extend(objs)

    def run(self, stream, first=0, last=None):
        for name, fn, args in self.calls[first:last]:
            rc = fn(*args, stream)
            if rc != 0:
                raise L.AddhipError(f"{name} failed ({rc}): {self._lib.addhip_last_error().decode()}")


def _group_key(call):
    """Signature under which two recorded GEMM calls may share a grouped launch (None: never)."""
    name, _, args = call
    if name != "addhip_gemm_f32":
        return None
    g = args[0]
    if g.split_k > 1 or g.M <= 8 or g.a_mean:  # split-K weight gradients gain nothing from grouping (measured), few-row launches take another kernel
        return None
    return (g.M, g.N, g.K, g.a_kcontig, g.b_kcontig, g.epilogue, g.precision, g.operands_bf16, g.accumulate, bool(g.mask_bits), bool(g.relu_bits),
            bool(g.C16), bool(g.C), bool(g.colsum))


def merge_sections(plan, sec_a, sec_b, group=True):
    """Append the calls of two independent recorded sections to `plan`, pairing GEMMs of equal signature -- in the order they occur in
    either section -- into addhip_gemm_grouped launches; every other call keeps its order within its own section.  Returns
    (pos_a, pos_b): the index in `plan.calls` at which each call of either section ended up."""
    lib = plan._lib
    base = len(plan.calls)
    out, pos_a, pos_b, ib = [], [], [None] * len(sec_b.calls), 0
    for ca in sec_a.calls:
        key = _group_key(ca) if group else None
        j = None
        if key is not None:
            j = next((j for j in range(ib, len(sec_b.calls)) if _group_key(sec_b.calls[j]) == key), None)
        if j is not None:
            for i in range(ib, j):
                pos_b[i] = base + len(out)
                out.append(sec_b.calls[i])
            arr = (L.GemmT * 2)(ca[2][0], sec_b.calls[j][2][0])
            plan.hold(arr)
            pos_a.append(base + len(out))
            pos_b[j] = base + len(out)
            out.append(("addhip_gemm_grouped", lib.addhip_gemm_grouped, (arr, 2)))
            ib = j + 1
        else:
            pos_a.append(base + len(out))
            out.append(ca)
    for i in range(ib, len(sec_b.calls)):
        pos_b[i] = base + len(out)
        out.append(sec_b.calls[i])
    plan.calls.extend(out)
    plan.keep.extend(sec_a.keep + sec_b.keep)
    return pos_a, pos_b


class NetRunner:
    """Forward / backward call recording for one Mlp over `rows` rows with its own activation buffers."""

    def __init__(self, model, net, rows, device, slabs, precision=L.PREC_F32, storage16=False):
        self.m, self.net, self.rows, self.slabs = model, net, rows, slabs
        self.precision = precision  # ADDHIP_PREC_* of every GEMM this runner records (agent.matmul_precision)
        # storage16 (agent.matmul_precision = bf16): hidden activations and pre-activation gradients are kept as bf16 in HBM and
        # the GEMMs read the model's bf16 weight shadow; the last hidden layer and the top gradient also exist in fp32 for the
        # loss-head kernels
        self.storage16 = bool(storage16)
        if self.storage16:
            self.h16 = [torch.zeros(rows, h, dtype=torch.bfloat16, device=device) for h in net.hidden]
            self.dz16 = [torch.zeros(rows, h, dtype=torch.bfloat16, device=device) for h in net.hidden]
        self.early_mark = None
        self.dw_marks = {}      # layer -> (first, last) plan call index of its weight-gradient group (GEMMs + split-K combine)
        self.aux_slabs = None   # (layer, buffer): that layer's weight-gradient group uses its own split-K scratch, so that it
        self._bits_valid = False  # may run on another stream than the rest of the backward pass
        # ReLU sign bits of the hidden activations (1 bit per element): what the backward GEMMs read as their mask instead of
        # the fp32 activations themselves (64 MB -> 2 MB per 16384 x 1024 layer)
        self.hb = [torch.zeros(rows, (h + 31) // 32, dtype=torch.int32, device=device) for h in net.hidden]
        self.h = [torch.zeros(rows, h, device=device) for h in net.hidden]
        self.dz = [torch.zeros(rows, h, device=device) for h in net.hidden]

    def _row_chunks(self, rows):
        """A few rows past a multiple of 128 (the discriminator's extra zero-difference sample: Mb + 1 rows) would cost a
        whole extra wave of 128-row tiles; they go into a second, tiny launch instead.  -> [(first_row, count), ...]
        (not in bf16-storage mode: its GEMM keeps 4 workgroups per CU in flight, so eight more tiles cost less than the
        launch, which sits on the discriminator's serial chain)"""
        rem = rows % 128
        if rows > 128 and 0 < rem <= 8 and not self.storage16:
            return [(0, rows - rem), (rows - rem, rem)]
        return [(0, rows)]

    def mask_args(self, layer, r0, cnt):
        """Mask of the backward pass through the ReLU of `layer` for rows [r0, r0+cnt): the sign bits where the forward pass
        of this plan wrote them (row chunks of more than 8 rows), the fp32 activations otherwise."""
        h = self.net.hidden[layer]
        if self._bits_valid and (cnt > 8 or self.storage16):
            ldb = (h + 31) // 32
            return dict(mask_bits=L.ptr(self.hb[layer]) + 4 * r0 * ldb, ldbits=ldb)
        return dict(mask=L.ptr(self.h[layer]) + 4 * r0 * h, ldmask=h)

    def forward(self, plan, x_ptr, rows, a_mean=None, a_std=None, sign_bits=False, x16_ptr=None):
        """sign_bits: also write the ReLU sign bits (forward passes that are followed by a backward pass in the same plan).
        x16_ptr: bf16 copy of the input rows (storage16 runners)."""
        net, m = self.net, self.m
        self._bits_valid = bool(sign_bits)
        if self.storage16:
            assert x16_ptr is not None and a_mean is None, "bf16-storage runners take a bf16 copy of their (already normalised) input"
            prev, ld, k = x16_ptr, net.in_ld, net.in_ld
            n = len(net.hidden)
            for i, h in enumerate(net.hidden):
                for r0, cnt in self._row_chunks(rows):
                    bits = dict(relu_bits=L.ptr(self.hb[i]) + 4 * r0 * ((h + 31) // 32), ldbits=(h + 31) // 32) if sign_bits else {}
                    g = gemm(cnt, h, k, prev + 2 * r0 * ld, ld, 1, m.p16(net.name, f"W{i}"), k, 1,
                             L.ptr(self.h[i]) + 4 * r0 * h if i == n - 1 else None, h, L.EPI_BIAS_RELU, m.p(net.name, f"b{i}"),
                             precision=L.PREC_BF16, operands_bf16=1, C16=L.ptr(self.h16[i]) + 2 * r0 * h, ldc16=h, **bits)
                    plan.hold(g)
                    plan.add("addhip_gemm_f32", g)
                prev, ld, k = L.ptr(self.h16[i]), h, h
            return
        prev, ld, k = x_ptr, net.in_ld, net.in_ld
        for i, h in enumerate(net.hidden):
            for r0, cnt in self._row_chunks(rows):
                bits = dict(relu_bits=L.ptr(self.hb[i]) + 4 * r0 * ((h + 31) // 32), ldbits=(h + 31) // 32) if sign_bits and cnt > 8 else {}
                g = gemm(cnt, h, k, prev + 4 * r0 * ld, ld, 1, m.p(net.name, f"W{i}"), k, 1, L.ptr(self.h[i]) + 4 * r0 * h, h, L.EPI_BIAS_RELU,
                         m.p(net.name, f"b{i}"), a_mean=a_mean if i == 0 else None, a_std=a_std if i == 0 else None, precision=self.precision, **bits)
                plan.hold(g)
                plan.add("addhip_gemm_f32", g)
            prev, ld, k = L.ptr(self.h[i]), h, h

    def backward(self, plan, x_ptr, rows, extra_dw=None, grads_zeroed=False, top_bias_done=False, x16_ptr=None, accumulate_dw=False, top_cast_done=False):
        """dz[-1] must hold d loss / d (pre-activation of the last hidden layer).  extra_dw: {layer: (A_ptr, lda, B_ptr, ldb)}
        second product accumulated into dW of that layer (the gradient-penalty terms).  grads_zeroed: the caller cleared the
        whole gradient buffer at the start of the step (no per-bias memsets here); top_bias_done: the kernel that produced
        dz[-1] also accumulated the top layer's bias gradient.  storage16 runners: x16_ptr = bf16 copy of the input rows, the
        extra_dw operands are bf16 too, and the fp32 top gradient is rounded to bf16 once at the start (top_cast_done: its producer
        already wrote dz16[-1]).  accumulate_dw: the
        weight gradients are ADDED to what the gradient buffer holds (L2 terms written there earlier in the step)."""
        net, m = self.net, self.m
        n = len(net.hidden)
        s16 = self.storage16
        esz = 2 if s16 else 4
        kw = dict(precision=L.PREC_BF16, operands_bf16=1) if s16 else dict(precision=self.precision)
        dz = self.dz16 if s16 else self.dz
        acts = self.h16 if s16 else self.h
        if s16:
            assert x16_ptr is not None and grads_zeroed
            if not top_cast_done:
                plan.add("addhip_to_bf16", L.ptr(self.dz[n - 1]), L.ptr(self.dz16[n - 1]), rows, net.hidden[n - 1], net.hidden[n - 1], net.hidden[n - 1])
        for i in reversed(range(n)):
            out_d = net.hidden[i]
            in_ld = net.in_ld if i == 0 else net.hidden[i - 1]
            inp = (x16_ptr if s16 else x_ptr) if i == 0 else L.ptr(acts[i - 1])
            s = split_k_for(out_d, in_ld, rows)
            slab = out_d * in_ld
            slabs = self.aux_slabs[1] if self.aux_slabs is not None and self.aux_slabs[0] == i else self.slabs
            dw_first = len(plan.calls)
            g = gemm(out_d, in_ld, rows, L.ptr(dz[i]), out_d, 0, inp, in_ld, 0, L.ptr(slabs), in_ld, split_k=s, **kw)
            plan.hold(g)
            plan.add("addhip_gemm_f32", g)
            total = s
            if extra_dw and i in extra_dw:
                a_ptr, lda, b_ptr, ldb, erows = extra_dw[i]
                g2 = gemm(out_d, in_ld, erows, a_ptr, lda, 0, b_ptr, ldb, 0, L.ptr(slabs) + 4 * s * slab, in_ld, split_k=s, **kw)
                plan.hold(g2)
                plan.add("addhip_gemm_f32", g2)
                total = 2 * s
            plan.add("addhip_slab_reduce", L.ptr(slabs), total, slab, m.g(net.name, f"W{i}"), slab, 1.0, int(accumulate_dw))
            self.dw_marks[i] = (dw_first, len(plan.calls))
            if i == 1:
                # every gradient of this net except W0 / b0 is final here (b1 came with the dX GEMM of layer 2, the head's
                # with the loss kernels): an early bucket for the data-parallel exchange
                self.early_mark = len(plan.calls)
            if i == n - 1 and not top_bias_done:  # the top layer's dz comes from the loss kernels; the others get their bias
                plan.add("addhip_col_sum", L.ptr(self.dz[i]), rows, out_d, out_d, m.g(net.name, f"b{i}"), 1.0, int(grads_zeroed))  # gradient from the dX GEMM below
            if i > 0:
                prev_d = net.hidden[i - 1]
                if not grads_zeroed:
                    plan.add("addhip_fill_zero", m.g(net.name, f"b{i - 1}"), prev_d)
                for r0, cnt in self._row_chunks(rows):
                    out = dict(C16=L.ptr(self.dz16[i - 1]) + 2 * r0 * prev_d, ldc16=prev_d) if s16 else {}
                    # dX = dz W: fp32 path reads W[out,in] n-contiguously; the bf16-storage path reads the transposed shadow W^T[in,out]
                    wb = (m.p16t(net.name, f"W{i}"), out_d, 1) if s16 else (m.p(net.name, f"W{i}"), prev_d, 0)
                    g3 = gemm(cnt, prev_d, out_d, L.ptr(dz[i]) + esz * r0 * out_d, out_d, 1, wb[0], wb[1], wb[2],
                              None if s16 else L.ptr(self.dz[i - 1]) + 4 * r0 * prev_d, prev_d, L.EPI_MASK, colsum=m.g(net.name, f"b{i - 1}"),
                              **kw, **out, **self.mask_args(i - 1, r0, cnt))
                    plan.hold(g3)
                    plan.add("addhip_gemm_f32", g3)
