"""Timing of addhip_env_step / addhip_env_reset alone at several env counts (HIP events on the launch stream)."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
args = [a for a in sys.argv[1:] if not a.startswith("--")]
motion = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--motion=")), "synthetic:1x3600")
print(f"motion library: {motion}", flush=True)
for N in [int(x) for x in (args or ["4096", "16384", "65536"])]:
    cfg = load_config("train", [f"engine.num_envs={N}", "agent.steps_per_iter=2", "agent.batch_size=1", f"task.motion_file={motion}"])
    ag = ADDAgent(cfg); ag.reset_all_envs(); ag._init_train()
    lib = ag._motion_lib
    print(f"step tables: {lib.get_num_motions()} clips, {lib.total_steps} rows, {2 * lib.total_steps * 36 * 4 / 1e6:.1f} MB", flush=True)
    st = torch.cuda.current_stream()
    gc.collect(); gc.freeze()  # a full collection inside the timed loop costs tens of ms
    out = ag._step_out[0]
    def run(reps, chunk=10):
        # chunks of 10 steps with an (untimed) reset of every env in between: the engine is not stepped here, so a longer
        # run would drift into "every env fails every step", which is not the steady state of a rollout
        import time
        total, wall = 0.0, 0.0
        for _ in range(reps // chunk):
            ag.reset_all_envs()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record(st)
            for _ in range(chunk):
                L.call("addhip_env_step", ag._motion_lib.c_struct, ag._task, ag._env_c, out, 0, st.cuda_stream)
            e1.record(st); e1.synchronize()
            wall += (time.perf_counter() - t0) * 1e3
            total += e0.elapsed_time(e1)
        n = reps // chunk * chunk
        return min(total, wall) / n  # HIP events occasionally report a bogus long interval on this stack; wall clock bounds it
    run(10)
    ms = run(50)
    gbs = 4573 * N / (ms * 1e-3) / 1e9
    print(f"env_step N={N:6d}: {ms*1e3:8.1f} us  {gbs:7.1f} GB/s algorithmic ({gbs/8000*100:.1f}% of 8 TB/s, {gbs/6290*100:.1f}% of 6.29 TB/s copy ceiling)", flush=True)
    del ag
