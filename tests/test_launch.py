"""The one-command launcher behind `bench.py --gpus N` / `python -m add_gym_amd.main --gpus=N` (the role torchrun plays for the
reference: sagemaker-entrypoint.sh:139-147, add_gym/main.py:128-176), on CPU ranks over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "launch_child.py")
DRIVER = "import sys, add_gym_amd; from add_gym_amd import launch; sys.exit(launch.spawn_ranks([%r] + sys.argv[1:], %d, timeout=120))"


def run(world, *args):
    return subprocess.run([sys.executable, "-c", DRIVER % (CHILD, world), *args], cwd=ROOT, capture_output=True, text=True, timeout=300)


@pytest.mark.timeout(300)
def test_spawn_ranks_starts_world_ranks_and_forwards_rank0_stdout_only():
    r = run(3)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # stdout carries rank 0's line and nothing else
    res = json.loads(lines[0])
    assert res == {"group_size": 3, "sum": 6.0, "local_rank": "0", "master": "127.0.0.1"}
    assert "must not reach" in r.stderr and "noise from rank 2" in r.stderr


@pytest.mark.timeout(300)
def test_a_failing_rank_stops_the_others_and_its_code_is_returned():
    r = run(2, "fail")
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with 7" in r.stderr


@pytest.mark.timeout(120)
def test_a_signalled_launcher_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to the launcher (a driver timeout, a killed job) must not leave ranks behind holding their GPUs."""
    import signal
    import time

    p = subprocess.Popen([sys.executable, "-c", DRIVER % (CHILD, 2), "hang", str(tmp_path)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.monotonic() + 60
        while time.monotonic() < deadline and not all((tmp_path / f"pid.{r}").exists() and (tmp_path / f"pid.{r}").read_text() for r in range(2)):
            time.sleep(0.1)
        pids = [int((tmp_path / f"pid.{r}").read_text()) for r in range(2)]
        p.send_signal(signal.SIGTERM)
        _, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-2000:])
    assert "stopping the ranks" in err
    for pid in pids:  # both children are gone (reaped by the launcher: the PID no longer exists, or at least is not our child's program)
        alive = True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            alive = False
        assert not alive, pid


def test_more_ranks_than_devices_is_refused_unless_rehearsing():
    import add_gym_amd  # noqa: F401
    from add_gym_amd import launch

    with pytest.raises(SystemExit):
        launch.check_world_fits(8, 1, env={})
    launch.check_world_fits(8, 8, env={})
    launch.check_world_fits(2, 1, env={launch.REHEARSAL_ENV: "gloo"})
    assert launch.backend({}) == "nccl" and launch.backend({launch.REHEARSAL_ENV: "gloo"}) == "gloo"
    assert launch.launched_by_a_launcher({"RANK": "0", "WORLD_SIZE": "2"}) and not launch.launched_by_a_launcher({"RANK": "0"})


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_launcher_refuses_more_ranks_than_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ADDHIP_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "one process per GPU" in r.stderr
