"""rigid.hip under a list of compiler flag sets: each variant linked with the other objects into a private libaddhip (built in the build
container), then on a GPU box: the float64-oracle case (tools/slp_repro/gpu_case.py) and the time of a control step at 4096 / 65 536 envs.

    python tools/rigid_flag_sweep.py build        (here)
    python tools/rigid_flag_sweep.py run          (gpurun)  -> gpurun_out/rigid_flag_sweep.log
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "add-gym_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "ubench", "_rigid_variants")
HIPCC = "/opt/rocm/bin/hipcc"
BASE = ["-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-function"]
OTHERS = [os.path.join(CSRC, f) for f in ("capi.o", "plan.o", "learner.o", "env_step.o", "gemm.o", "gemm_split.o", "gemm_bf16.o", "gemm_x3.o", "learn.o", "actor_head.o")]
VARIANTS = {
    "product": [],
    "approx-div-sqrt": ["-fno-hip-fp32-correctly-rounded-divide-sqrt"],
    "no-unroll": ["-fno-unroll-loops"],
    "unroll-more": ["-mllvm", "-unroll-threshold=2000"],
    "ilp-sched": ["-mllvm", "-amdgpu-schedule-metric-bias=90"],
    "max-ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "max-memory-clause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "early-inline-all": ["-mllvm", "-amdgpu-early-inline-all=true"],
    "divergent-reg-indexing": ["-mllvm", "-amdgpu-use-divergent-register-indexing"],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    for name, flags in VARIANTS.items():
        obj = os.path.join(OUT, name + ".o")
        r = subprocess.run([HIPCC] + BASE + flags + ["-c", os.path.join(CSRC, "rigid.hip"), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{name:24s} NOT BUILT: {r.stderr.strip().splitlines()[-1][:150] if r.stderr.strip() else '?'}", flush=True)
            continue
        subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(OUT, f"libaddhip_{name}.so"), obj] + OTHERS, check=True)
        os.remove(obj)
        print(f"{name:24s} built  {' '.join(flags)}", flush=True)


BENCH = r"""
import os, sys
sys.path.insert(0, %r)
import add_gym_amd
from add_gym_amd import _lib as L
L.LIB_PATH = sys.argv[1]
import torch
from tests.test_hip_rigid import make_entity
out = []
for n in (4096, 65536):
    eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=4)
    pose0 = ent.pose.clone(); pose0[:, 2] = 0.79
    ent.control_dofs_position((torch.randn(n, 32, device="cuda") * 0.1).contiguous())
    st = torch.cuda.current_stream()
    ts = []
    for chunk in range(6):
        ent.pose.copy_(pose0); ent.vel.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            scene.step()
        e1.record(st); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    out.append("%%d envs %%.1f us" %% (n, sorted(ts[1:])[2]))
print("BENCH " + ", ".join(out))
""" % ROOT


def run():
    log = os.path.join(ROOT, "gpurun_out", "rigid_flag_sweep.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as f:
        for name, flags in VARIANTS.items():
            lib = os.path.join(OUT, f"libaddhip_{name}.so")
            if not os.path.exists(lib):
                continue
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slp_repro", "gpu_case.py"), lib], capture_output=True, text=True, timeout=300)
            cases = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
            ok = "WRONG" if any("WRONG" in c for c in cases) else ("ok" if len(cases) == 2 else "FAILED")
            b = subprocess.run([sys.executable, "-c", BENCH, lib], capture_output=True, text=True, timeout=300)
            bl = [l for l in b.stdout.splitlines() if l.startswith("BENCH")]
            line = f"{name:24s} oracle {ok:6s} {(bl[0][6:] if bl else 'bench failed: ' + b.stderr[-200:])}   [{' '.join(flags)}]"
            print(line, flush=True)
            f.write(line + "\n")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
