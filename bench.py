#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of rollout + update (BaseAgent._train_iter, base_agent.py:353-374; test rollouts
excluded), G1 imitation, 4096 envs/GPU, fp32 PPO + ADD discriminator (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one training iteration: T=32 env steps of all envs (actor MLP, engine step, fused obs/reward/done kernel,
masked reset) + build-train-data (disc reward, critic values, TD(lambda), advantage) + 5 epochs x 8 minibatches of
forward/backward/AdamW.  Synthetic G1 clips, random-init weights, the KinematicEngine stand-in simulator (physics is
out of scope: DESIGN.md).  Environments are sharded across ranks (weak scaling); the only data-path collective is the
all-reduce (RCCL) of the flat fp32 gradient each optimiser step plus the once-per-iteration normaliser sums.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the fp32 MFMA GEMM, timed live with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle's loop on a bounded sample, N=1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (matrix), v_mfma_f32_32x32x2_f32
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16, 32 cycles per instruction per SIMD)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy ceiling)
ENV_STEP_BYTES = 4573         # algorithmic bytes per env-step of the fused obs/reward/done kernel
VALU_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
RIGID_FLOP_PER_SUBSTEP = 44000  # fp32 operations of one articulated-body sweep of the 30-body G1 (29 x ~1400 per body + root solve + contacts)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=0, help="envs of the CPU-oracle sample; 0 = the headline's own env count (BASELINE.md section 3: same N). "
                    "One warm-up + one timed iteration (about 25 s each at 4096 envs on 16 cores)")
    ap.add_argument("--no-configs2", action="store_true", help="skip the BASELINE configs[2] entry (16 384 envs, 5-clip set, bf16-storage MLP path)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra bf16x3 measurement after the headline run")
    ap.add_argument("--with-alt", action="store_true", help="N > 1: also run the alternative modes / engine / configs[2] entries (default there: headline only, "
                                                             "so that a scaling run is the headline workload and nothing else)")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="N=1 only: join a 1-rank RCCL group so that the multi-rank exchange path (async gradient buckets) runs; rehearsal, not a headline")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16x2", "bf16"],
                    help="agent.matmul_precision: how addhip_gemm_f32 forms its fp32 products (include/addhip.h ADDHIP_PREC_*)")
    return ap.parse_args()


def traffic(key):
    """HBM bytes per launch from the committed PMC profiles (profiles/traffic.json, written from rocprofv3 --pmc passes): (bytes, source)
    or (None, None).  Never a constant in this file: the JSON names the profile each figure came from."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]
        return float(e["traffic_bytes_per_launch"]), e["source"]
    except Exception:
        return None, None


def time_gemms(agent, reps=10, warm=5):
    """Duration of the GEMM launches of one optimiser step: every launch of the recorded step (addhip_plan_t) replayed alone through the C
    ABI, in plan order, bracketed by HIP events recorded on the stream the kernel is launched on (torch's current stream == the stream
    handed to the C ABI).  `warm` untimed replays first (an idle chip's clocks ramp over the first milliseconds of load), then `reps` timed
    ones; a launch's duration is its median over the replays, the step's figure their sum."""
    import statistics
    import torch

    st = torch.cuda.current_stream()
    plan = agent._update_plan
    # (the actor's head section is one launch since round 4 -- addhip_actor_head: three 32-wide products on the fp32 MFMA + the loss; it stays in
    #  this account with its 3 x 2 x Mb x 32 x hidden FLOP, as the three GEMM launches it replaced were)
    listed = plan.launches()
    calls = [(i, gemms) for i, (name, gemms) in enumerate(listed) if gemms or name == "addhip_actor_head"]
    head_flops = 3 * 2.0 * agent.Mb * 32 * agent._model.actor.hidden[-1]
    flops = sum(2.0 * g.M * g.N * g.K for _, gemms in calls for g in gemms) + head_flops * sum(1 for i, gemms in calls if not gemms)
    for _ in range(warm):
        for i, _ in calls:
            plan.run(st.cuda_stream, i, i + 1)
    torch.cuda.synchronize()
    per_launch = [[] for _ in calls]
    for _ in range(reps):
        for n, (i, _) in enumerate(calls):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            plan.run(st.cuda_stream, i, i + 1)
            e1.record(st)
            e1.synchronize()
            per_launch[n].append(e0.elapsed_time(e1))
    ms = sum(statistics.median(t) for t in per_launch)
    return dict(launches=len(calls), flops=flops, ms=ms)


def gemm_roofline(agent, precision, with_traffic=True):
    """HIP-event time of all GEMM launches of one optimiser step against the MFMA peak of the instruction that forms the
    products: fp32 MFMA 157.3 TFLOP/s; bf16x3 = six bf16 MFMAs per fp32 product -> 2500 / 6 TFLOP/s of fp32 work."""
    g = time_gemms(agent)
    tf = g["flops"] / (g["ms"] * 1e-3) / 1e12
    if precision == "fp32":
        peak, kern = MFMA_F32_PEAK_TFLOPS, "gemm_kernel (fp32 v_mfma_f32_32x32x2_f32; all GEMM launches of one optimiser step)"
        tr, tr_src = traffic("gemm_fp32_step") if with_traffic else (None, None)
    else:
        products = {"bf16x3": 6, "bf16x3_planes": 6, "f16x2": 4, "bf16x2": 3, "bf16": 1}[precision]
        peak = MFMA_BF16_PEAK_TFLOPS / products  # (fp16 MFMAs run at the bf16 rate)
        kern = {"bf16": "gemm_dma_kernel<bf16> (bf16 operands in HBM, ", "bf16x3_planes": "gemm_x3_kernel (operands stored as three exact bf16 planes, LDS-DMA ring, ",
                "f16x2": "gemm_split_kernel<f16> (fp32 operands split two ways into fp16 on tracked per-tensor scales, "}.get(precision, "gemm_split_kernel (") + \
               "v_mfma_f32_32x32x16_%s x %d per k-step; small shapes stay on gemm_kernel); all GEMM launches of one optimiser step" % ("f16" if precision == "f16x2" else "bf16", products)
        # (counters: profiles/traffic.json -- taken at 4096 envs, and for bf16 storage at the 16 384 envs of BASELINE configs[2] as well)
        key = {"bf16": "gemm_bf16_step_16384" if agent.N == 16384 else "gemm_bf16_step", "f16x2": "gemm_f16x2_step"}.get(precision)
        tr, tr_src = traffic(key) if key and (with_traffic or agent.N == 16384) else (None, None)
    return {"bound": "mfma", "kernel": kern, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": tr, "traffic_source": tr_src,
            "launches_per_step": g["launches"], "gflop_per_step": g["flops"] / 1e9, "ms_per_step": g["ms"],
            "frac_of_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS}


def env_step_at_scale(num_envs=65536, motion="synthetic:1x3600"):
    """The fused env step alone at a size where it is not launch-latency bound (same kernels, same task config)."""
    import add_gym_amd  # noqa: F401
    from add_gym_amd.config import load_config
    from add_gym_amd.learning.add_agent import ADDAgent

    cfg = load_config("train", [f"engine.num_envs={num_envs}", "agent.steps_per_iter=2", "agent.batch_size=1", f"task.motion_file={motion}"])
    ag = ADDAgent(cfg)
    ag.reset_all_envs()
    ag._init_train()
    gc.collect()
    gc.freeze()
    ms = time_env_step(ag, reps=50)
    table_mb = 2 * ag._motion_lib.total_steps * 36 * 4 / 1e6
    del ag
    return ms, table_mb


def rigid_step_roofline(num_envs):
    """The rigid-body engine's control step alone (addhip_rigid_step: `substeps` articulated-body sweeps per launch) on standing
    robots.  Neither HBM- nor MFMA-bound: four lanes per env (one per chain of the tree), dependent fp32 VALU chains, a wave or two per
    CU; reported against the fp32 vector peak with the kernel's arithmetic counted from its own operation list."""
    import torch
    import add_gym_amd  # noqa: F401
    from add_gym_amd.config import load_config
    from add_gym_amd.envs.env import ImitationEnvironment

    cfg = load_config("train", ["engine=rigid", f"engine.num_envs={num_envs}"])
    env = ImitationEnvironment(cfg, torch.device("cuda", torch.cuda.current_device()))
    ent = env.robot.entity
    pose0 = ent.pose.clone()
    pose0[:, 2] = 0.79
    tgt = (torch.randn(num_envs, 32, device="cuda") * 0.1).contiguous()
    ent.control_dofs_position(tgt)
    st = torch.cuda.current_stream()
    tot, cnt = 0.0, 0
    for chunk in range(4):
        ent.pose.copy_(pose0)
        ent.vel.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            env.scene.step()
        e1.record(st)
        e1.synchronize()
        if chunk:
            tot += e0.elapsed_time(e1)
            cnt += 20
    ms = tot / cnt
    sub = int(ent._opts["substeps"])
    flop = RIGID_FLOP_PER_SUBSTEP * sub * num_envs
    tf = flop / (ms * 1e-3) / 1e12
    lanes = 4 if getattr(ent, "_d_chains", None) is not None else 1
    return {"bound": "valu-latency", "kernel": "rigid_step%s_kernel (addhip_rigid_step, %d lane(s) per env, %d substeps per launch)" % ("4" if lanes == 4 else "", lanes, sub), "achieved": tf, "peak": VALU_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / VALU_F32_PEAK_TFLOPS, "traffic": None, "us_per_launch": ms * 1e3, "envs": num_envs,
            "env_steps_per_s": num_envs / (ms * 1e-3), "flop_per_env_substep": RIGID_FLOP_PER_SUBSTEP,
            "algorithmic_bytes_per_env_step": 2 * 2 * 36 * 4 + 29 * 4 + 5,
            "note": "issue-bound by construction: a 30-body tree walked three times per substep, cut into four chains on four lanes of a quad"}


def time_env_step(agent, reps=50, chunk=10):
    """Average duration of addhip_env_step (one launch: env_step_kernel), HIP events on the launch stream.  Chunks of 10
    steps with an untimed reset of every env in between: the simulator is not stepped here, so a longer run would drift
    into "every env fails every step", which is not the steady state of a rollout."""
    import torch
    import add_gym_amd._lib as L

    st = torch.cuda.current_stream()
    out = agent._step_out[0]
    total = 0.0
    for c in range(reps // chunk + 1):
        agent.reset_all_envs()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(chunk):
            L.call("addhip_env_step", agent._motion_lib.c_struct, agent._task, agent._env_c, out, 0, st.cuda_stream)
        e1.record(st)
        e1.synchronize()
        if c > 0:  # first chunk = warm-up
            total += e0.elapsed_time(e1)
    return total / (reps // chunk * chunk)


def host_cores():
    """CPU share of this job: cgroup quota if set, else the affinity mask, capped at 16 (the GPU box gives one GPU's job
    16 cores; os.cpu_count() reports the whole host and oversubscribing OpenMP by 10x makes torch-CPU crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(num_envs, steps_per_iter, timed_iters=1):
    """The CPU oracle's iteration (oracle/loop.py) on a bounded sample of the same workload, all host cores."""
    import numpy as np
    import torch
    from oracle import learn as OL
    from oracle import loop as LP
    from oracle import task as OT
    from oracle.kin import KinTree
    from oracle.motion import MotionLib
    import add_gym_amd  # noqa: F401
    from add_gym_amd.anim.kin_char_model import KinCharModel
    from add_gym_amd.anim.synth import synth_clip
    from add_gym_amd.config import load_config

    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = load_config("train", [])
    xml = cfg["robot"]["urdf_path"]
    order = list(cfg["task"]["motion_joint_order"])
    kin_p = KinCharModel()
    kin_p.load_char_file(xml)
    frames = synth_clip(kin_p, order, 0, 900)
    lib = MotionLib([frames], [1.0], order, KinTree(xml), 0.01)
    rng = np.random.RandomState(0)
    n, Tn = num_envs, steps_per_iter
    ag = LP.Agent(LP.AgentCfg(steps_per_iter=Tn), OT.TaskCfg(), lib, n, OL.synth_params(1, bias_scale=0.0))
    draw = lambda k: dict(ids=np.zeros(k, np.int64), segments=rng.randint(0, 20, k), jitter=rng.rand(k).astype(np.float32))
    ag.init(draw(n))
    total = Tn * n
    perms = [rng.permutation(total) for _ in range(32)]
    draws = LP.Draws(rng.standard_normal((Tn, n, 29)).astype(np.float32), lambda t, ids: draw(len(ids)), perms)
    times = []
    for it in range(1 + timed_iters):  # bounded sample: 1 warm-up + `timed_iters` timed iterations of the headline's own size, median
        t0 = time.perf_counter()
        ag.train_iter(draws)
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu baseline iteration {it}: {times[-1]:.1f} s", file=sys.stderr, flush=True)
    dt = sorted(times[1:])[(timed_iters - 1) // 2]
    num_mb = int(np.ceil(total / float(ag.cfg.batch_size * n)))
    return dict(value=total / dt, unit="env-steps/s", cores=cores, kind="port", envs=n,
                sample=f"CPU oracle (numpy + torch-CPU fp32, {cores} threads): {n} envs x {Tn} steps rollout + {ag.cfg.update_epochs} epochs x "
                       f"{num_mb} minibatches update per iteration; 1 warm-up + {timed_iters} timed iteration(s), median {dt:.1f} s")


def cpu_baseline_reference():
    """The imported reference's own _train_iter, timed in the BUILD container (tools/time_reference.py; the reference cannot travel to
    the GPU box): a committed record, copied here with its source so that both CPU figures of BASELINE.md section 3 are on the line."""
    src = os.path.join("profiles", "r02_time_reference_4096.log")
    try:
        rec = json.loads(open(os.path.join(ROOT, src)).read().strip().splitlines()[-1])
        return {"value": rec["env_steps_per_s"], "unit": "env-steps/s", "cores": rec["threads"], "kind": "reference", "envs": rec["num_envs"],
                "where": "build container (%s), not the GPU box" % rec["cpu"], "source": src,
                "sample": "add_gym's own _train_iter on a kinematic fake engine, %d envs x %d steps, %d epochs; 1 warm-up + %d timed iterations, median %.1f s; %s"
                          % (rec["num_envs"], rec["steps_per_iter"], rec["update_epochs"], rec["timed_iters"], rec["median_s"], rec["physics"])}
    except Exception:
        return None


def main():
    a = parse()
    from add_gym_amd import launch

    if not launch.launched_by_a_launcher():
        if a.gpus > 1:
            # `python bench.py --gpus N` with no launcher around it: this process becomes the launcher (it never touches the GPU;
            # counting devices does not initialise HIP) and every rank is a fresh child of this same file.
            import torch

            launch.check_world_fits(a.gpus, torch.cuda.device_count())
            raise SystemExit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%s ranks" % (a.gpus, os.environ["WORLD_SIZE"]))
    # stdout carries exactly one thing: the JSON line.  Libraries that print there (RCCL's version banner at communicator
    # creation, for one) are sent to stderr for the whole run; the line itself goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    launch.check_world_fits(world, torch.cuda.device_count())
    launch.bind_device(local_rank)
    distributed = world > 1 or a.exercise_exchange
    if world > 1 and not a.with_alt:
        a.no_alt = True
    if distributed and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if distributed:
        # "nccl" is RCCL on ROCm.  ADDHIP_DIST_BACKEND=gloo exists only to rehearse the multi-rank plumbing on a box with
        # fewer GPUs than ranks (several ranks then share a device); it is never the measured configuration.
        dist.init_process_group(backend=launch.backend())
    import add_gym_amd  # noqa: F401
    from add_gym_amd.config import load_config
    from add_gym_amd.learning.add_agent import ADDAgent

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def make_agent(precision, engine="kinematic", envs=None, motion="synthetic:1x3600", extra=(), dist_on=None):
        cfg = load_config("train", [f"engine={engine}", f"engine.num_envs={envs or a.envs}", f"task.motion_file={motion}", f"seed={1 + rank}",
                                    f"agent.matmul_precision={precision}"] + list(extra))
        ag = ADDAgent(cfg, distributed=distributed if dist_on is None else dist_on)
        ag.reset_all_envs()
        ag._init_train()
        gc.collect()
        gc.freeze()  # as ADDAgent.train_model does: keep full collections (tens of ms) out of the launch loop
        return ag

    def timed(ag):
        """W untimed + exactly K timed iterations, barrier + synchronize on both sides, max over ranks."""
        for _ in range(a.warmup):
            ag._train_iter()
            ag._iter += 1
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ag._train_iter()
            ag._iter += 1
        sync()
        dt = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    agent = make_agent(a.precision)
    dt = timed(agent)
    env_steps = agent.T * agent.N * world * a.steps
    out = {
        "metric": "env-steps/sec (rollout+update), G1 imitation, 4096 envs/GPU",
        "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (generated G1 clip, random-init weights, kinematic stand-in simulator)"
                + ("" if not distributed or dist.get_backend() == "nccl" else " [REHEARSAL: %s backend, ranks share GPUs]" % dist.get_backend()),
        "config": {"workload": "G1 walk-like synthetic motion, num_envs=%d/GPU, fp32 PPO + ADD discriminator (BASELINE configs[1])" % a.envs,
                   "envs_per_gpu": a.envs, "steps_per_iter": agent.T, "update_epochs": agent._update_epochs, "minibatch_rows": agent.Mb,
                   "params": agent.calc_num_params(), "parallelism": "dp%d (envs sharded, gradient all-reduce)" % world},
        "dist": {"backend": dist.get_backend() if distributed else None, "group_size": dist.get_world_size() if distributed else 1,
                 "launcher": "torchrun/env" if "TORCHELASTIC_RUN_ID" in os.environ else ("bench.py --gpus" if launch.launched_by_a_launcher() else None)},
    }
    if rank == 0:
        print(f"[bench] timed region done: {env_steps / dt:.0f} env-steps/s; measuring kernels + cpu baseline", file=sys.stderr, flush=True)
        out["roofline"] = gemm_roofline(agent, a.precision)
        ms = time_env_step(agent)
        gbs = ENV_STEP_BYTES * agent.N / (ms * 1e-3) / 1e9
        out["roofline_env_step"] = {"bound": "hbm", "kernel": "env_step_kernel (addhip_env_step)", "achieved": gbs,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "us_per_launch": ms * 1e3,
                                    "envs": agent.N, "note": "launch-latency scale at this env count; see roofline_env_step_65536"}
        if world == 1:
            big = 65536
            for key, motion in (("roofline_env_step_65536", "synthetic:1x3600"), ("roofline_env_step_65536_fulltable", "synthetic:43x3600")):
                ms2, table_mb = env_step_at_scale(big, motion)
                gbs2 = ENV_STEP_BYTES * big / (ms2 * 1e-3) / 1e9
                tr, tr_src = traffic(key[len("roofline_"):])
                out[key] = {"bound": "hbm", "kernel": "env_step_kernel (addhip_env_step)", "achieved": gbs2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": gbs2 / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src, "us_per_launch": ms2 * 1e3, "envs": big,
                            "motion_library": motion, "step_tables_mb": table_mb}
            out["roofline_rigid_step"] = rigid_step_roofline(a.envs)
            # (one wave per CU at the headline's env count; the same kernel with the chip full -- the register form, four waves per CU)
            out["roofline_rigid_step_65536"] = rigid_step_roofline(65536)
    steps_per_iter = agent.T
    if a.precision == "fp32" and not a.no_alt:
        # Same workload with the bf16-MFMA product modes of addhip_gemm_f32 (operands, results and every other kernel stay fp32;
        # include/addhip.h ADDHIP_PREC_*), reported beside the headline, which stays on the fp32 MFMA instruction:
        #   bf16x3 = exact 3-way bf16 split of every operand, six MFMAs per k-step, fp32-level error bound;
        #   bf16x2 = the two leading chunks, three MFMAs, error 2^-15 |a||b| (the reference's GPU path runs TF32: 2^-11).
        # Every rank runs them (the gradient all-reduce needs all ranks).
        del agent
        alts = []
        for prec, what in (("bf16x3", "6 bf16 MFMAs on an exact 3-way split (fp32-level error)"),
                           ("f16x2", "4 fp16 MFMAs on a two-way fp16 split (22 bits + sign per operand) scaled by exact per-tensor powers of two from "
                                     "device-tracked maxima: fp32-class error, 2^-21 |a||b| per product"),
                           ("bf16x3_planes", "6 bf16 MFMAs on operands STORED as three exact bf16 planes (update step: csrc/gemm_x3.hip, LDS-DMA ring; "
                                             "the error of bf16x3)"),
                           ("bf16x2", "3 bf16 MFMAs on the two leading chunks (16 significant bits per operand; TF32-class, 64x less error than TF32)"),
                           ("bf16", "bf16 STORAGE: activations, gradients and a weight shadow kept as bf16 in HBM (update step), 1 bf16 MFMA per k-step, "
                                    "fp32 accumulate and master weights; rollout / evaluation passes on bf16x2")):
            agent2 = make_agent(prec)
            dt2 = timed(agent2)
            if rank == 0:
                alts.append({"matmul_precision": prec, "value": env_steps / dt2, "unit": "env-steps/s", "ms_per_step": 1000.0 * dt2 / a.steps,
                             "dtype": ("bf16 operands; " if prec == "bf16" else "f32 operands and results; products = ") + what + ", fp32 accumulate",
                             "roofline": gemm_roofline(agent2, prec)})
            del agent2
        if rank == 0:
            out["alt_precision"] = alts
    if not a.no_alt:
        # Same workload with the rigid-body engine (engine=rigid: articulated-body dynamics + PD + ground contact, csrc/rigid.hip)
        # in place of the kinematic stand-in the headline uses (the headline keeps BASELINE.md's "physics excluded" protocol).
        agent3 = make_agent(a.precision, engine="rigid")
        dt3 = timed(agent3)
        if rank == 0:
            out["alt_engine"] = {"engine": "rigid (add_gym_amd.engine.rigid_engine.RigidBodyEngine, 4 substeps per control step)",
                                 "value": env_steps / dt3, "unit": "env-steps/s", "ms_per_step": 1000.0 * dt3 / a.steps,
                                 "matmul_precision": a.precision}
        del agent3
    if not a.no_alt and not a.no_configs2:
        # BASELINE configs[2]: "G1 mixed locomotion motion set, num_envs=16384, 1xMI355X, bf16 MLP MFMA path" -- five synthetic clips,
        # 16 384 envs per GPU, agent.matmul_precision=bf16 (bf16 storage) in the update step AND, with agent.rollout_precision=bf16_storage, in the
        # rollout / value / discriminator-reward passes; parity of this composition with its stated tolerances: tests/test_hip_fullsize.py
        # (test_16384_envs_five_clips_subset_matches_oracle[bf16+bf16_storage])
        agent4 = make_agent("bf16", envs=16384, motion="synthetic:5x1200", extra=("agent.rollout_precision=bf16_storage",))
        dt4 = timed(agent4)
        if rank == 0:
            out["alt_config"] = {"config": "BASELINE configs[2]: 5-clip synthetic locomotion set, num_envs=16384/GPU, bf16-storage MLP MFMA path "
                                           "(update step, rollout and evaluation passes: agent.matmul_precision=bf16, agent.rollout_precision=bf16_storage)",
                                 "envs": 16384, "precision": "bf16", "matmul_precision": "bf16", "rollout_precision": "bf16_storage", "motion_library": "synthetic:5x1200",
                                 "value": agent4.T * agent4.N * world * a.steps / dt4, "unit": "env-steps/s", "ms_per_step": 1000.0 * dt4 / a.steps,
                                 "minibatch_rows": agent4.Mb, "roofline": gemm_roofline(agent4, "bf16", with_traffic=False)}
        del agent4
    if world == 1 and not distributed and not a.no_alt:
        # N > 1 pre-flight that one GPU can give: the SAME workload with the exchange machinery on -- a 1-rank RCCL group, so every optimiser
        # step issues its four asynchronous gradient buckets from the schedule's call-backs on their streams, the normaliser statistics and
        # the logged scalars are all-reduced -- next to the plain run above.  The collectives are identities; what is measured is their issue
        # and synchronisation cost on the path an N-rank job takes.
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group(backend=launch.backend())
            agent5 = make_agent(a.precision, dist_on=True)
            for _ in range(a.warmup):
                agent5._train_iter()
                agent5._iter += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                agent5._train_iter()
                agent5._iter += 1
            torch.cuda.synchronize()
            dt5 = time.perf_counter() - t0
            br = agent5._model.bucket_ranges
            out["dist_overhead_1rank"] = {"backend": dist.get_backend(), "ms_per_step": 1000.0 * dt5 / a.steps, "plain_ms_per_step": out["ms_per_step"],
                                          "vs_plain": (dt5 / a.steps) / (dt / a.steps), "optimiser_steps_per_iter": agent5._update_epochs * ((agent5.T * agent5.N + agent5.Mb - 1) // agent5.Mb),
                                          "buckets_per_optimiser_step": [{"bucket": k, "bytes": 4 * (v[1] - v[0])} for k, v in br.items()],
                                          "note": "1-rank RCCL group: collectives are identities; the figure is the cost of issuing / joining them on the N-rank code path"}
            del agent5
            dist.destroy_process_group()
        except Exception as e:  # (never let the pre-flight take the headline line down)
            out["dist_overhead_1rank"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # last: the oracle's 16 torch-CPU threads keep spinning after their parallel regions, and the launch-bound GPU runs above
        # (the bf16 modes enqueue ~90 kernels per 1.1 ms optimiser step) slow down when they share the host cores with them
        out["cpu_baseline"] = cpu_baseline(a.cpu_envs or a.envs, steps_per_iter)
        out["cpu_baseline_reference"] = cpu_baseline_reference()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
