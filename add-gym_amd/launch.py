"""One process per GPU, started from a single command (the role `torchrun --nproc_per_node=$GPUS_PER_NODE` plays for the reference:
sagemaker-entrypoint.sh:139-147; each rank then finds RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment exactly as
add_gym/main.py:128-176 expects).

Two rules of this pool shape the code: the LAUNCHER never touches the GPU (it only counts devices, which does not initialise HIP),
and a rank is always a fresh child process — a process that has initialised the GPU is never replaced or forked.
"""
import os
import socket
import subprocess
import sys
import threading

REHEARSAL_ENV = "ADDHIP_DIST_BACKEND"  # "gloo": ranks may share a GPU (plumbing rehearsal on a box with fewer GPUs than ranks); never a measurement


def launched_by_a_launcher(env=None):
    """True inside a rank started by torchrun / spawn_ranks (main.py:127: both variables present)."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def backend(env=None):
    env = os.environ if env is None else env
    return env.get(REHEARSAL_ENV, "nccl")  # "nccl" is RCCL on ROCm


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_world_fits(world, device_count, env=None):
    """One rank per GPU or nothing: more ranks than devices is an error unless the gloo rehearsal switch is set."""
    if world > device_count and backend(env) != "gloo":
        raise SystemExit(f"{world} ranks asked for but {device_count} GPU(s) visible: one process per GPU "
                         f"(set {REHEARSAL_ENV}=gloo only to rehearse the rank plumbing on shared devices)")


def bind_device(local_rank, env=None):
    """Bind this rank to its own GPU (main.py:141-155 does it by masking).  A LOCAL_RANK beyond the visible devices fails loudly
    instead of wrapping onto somebody else's GPU; only the gloo rehearsal may share devices."""
    import torch

    n = torch.cuda.device_count()
    if n == 0:
        raise SystemExit("no GPU visible: the hot path is HIP-only")
    if local_rank >= n:
        if backend(env) != "gloo":
            raise SystemExit(f"LOCAL_RANK={local_rank} but only {n} GPU(s) visible: refusing to put two ranks on one device "
                             f"(set {REHEARSAL_ENV}=gloo to rehearse on shared devices)")
        local_rank %= n
    torch.cuda.set_device(local_rank)
    return local_rank


def _pump(src, dst):
    for line in iter(src.readline, b""):
        dst.write(line)
        dst.flush()
    src.close()


def spawn_ranks(script_argv, world, timeout=None, extra_env=None):
    """Start `world` fresh children of `python script_argv...`, one per rank, on 127.0.0.1; rank 0's stdout is forwarded to ours
    (it carries the result line), every other stream goes to stderr.  Returns 0 when all ranks exit 0; on the first failure the
    remaining ranks (exactly the PIDs started here) are terminated and that rank's exit code is returned.  The same clean-up runs on
    every other way out: timeout (124), SIGTERM / SIGINT delivered to the launcher (128 + signal), an exception in the wait loop."""
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver
    env0.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world)})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # where the `add_gym_amd` import shim lives
    env0["PYTHONPATH"] = root + (os.pathsep + env0["PYTHONPATH"] if env0.get("PYTHONPATH") else "")
    env0.update(extra_env or {})
    procs, pumps = [], []
    out = getattr(sys.stdout, "buffer", sys.stdout)
    err = getattr(sys.stderr, "buffer", sys.stderr)
    import signal
    import time

    class _Stop(Exception):
        pass

    def _on_signal(signum, frame):  # a launcher told to stop (driver timeout, SIGTERM, ^C) takes its ranks with it: the finally below
        raise _Stop(signum)

    # (signal handlers can only be installed from the main thread; elsewhere the finally still covers exceptions and normal exits)
    in_main = threading.current_thread() is threading.main_thread()
    old_handlers = {sig: signal.signal(sig, _on_signal) for sig in (signal.SIGTERM, signal.SIGINT)} if in_main else {}
    rc, alive = 0, set()
    try:
        for rank in range(world):
            env = dict(env0, RANK=str(rank), LOCAL_RANK=str(rank))
            p = subprocess.Popen([sys.executable] + list(script_argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            procs.append(p)
            alive.add(rank)
            for src, dst in ((p.stdout, out if rank == 0 else err), (p.stderr, err)):
                t = threading.Thread(target=_pump, args=(src, dst), daemon=True)
                t.start()
                pumps.append(t)
        t0 = time.monotonic()
        while alive and rc == 0:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is not None:
                    alive.discard(r)
                    if code != 0:
                        rc = code if code > 0 else 1
                        print(f"[launch] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr, flush=True)
                        break
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc = 124
                print(f"[launch] ranks still running after {timeout} s: stopping them", file=sys.stderr, flush=True)
            time.sleep(0.05)
    except _Stop as stop:
        rc = 128 + int(stop.args[0])
        print(f"[launch] signal {int(stop.args[0])}: stopping the ranks", file=sys.stderr, flush=True)
    finally:
        # every exit path -- failure, timeout, signal, an exception in the loop above -- stops exactly the PIDs started here: a rank left
        # behind would sit in its rendezvous or a collective and keep its GPU
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
        left = [p for p in procs if p.poll() is None]
        for p in left:
            p.terminate()
        for p in left:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for t in pumps:
            t.join(timeout=5)
    return rc
