"""CPU-only: include/addhip.h is valid C99 and valid C++17, and a host written in plain C records the optimiser step through the composite
entry points and lays the schedule over it (tests/c_host/plan_host.c = INTEGRATION.md section 2b) -- the figures it prints are the ones the
Python agent's plan has on the GPU (55 launches, 33 of them GEMMs, 471.9 GFLOP at a 16 384-row minibatch)."""
import os
import re
import subprocess

from tests.util import ROOT

INC = os.path.join(ROOT, "include")


def test_header_is_valid_c_and_cpp(tmp_path):
    for compiler, std, src in (("gcc", "-std=c99", "t.c"), ("g++", "-std=c++17", "t.cpp")):
        f = tmp_path / src
        f.write_text('#include "addhip.h"\nint main(void) { return sizeof(addhip_gemm_t) > 0 ? 0 : 1; }\n')
        subprocess.run([compiler, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I" + INC, str(f)], check=True)


def test_c_host_records_the_optimiser_step(tmp_path):
    import add_gym_amd._lib as L

    L.load()  # (fails loudly if the library is not built)
    exe = str(tmp_path / "plan_host")
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + INC, os.path.join(ROOT, "tests", "c_host", "plan_host.c"), "-o", exe,
                    "-L" + libdir, "-laddhip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    m = re.search(r"version=(\d+) launches=(\d+) gemm_launches=(\d+) gflop=([\d.]+) sections=(\d+) actor=\[0,(\d+)\) early=(\d+) critic_end=(\d+) disc_end=(\d+) "
                  r"first=(\w+) last=(\w+)", out)
    assert m, out
    version, launches, gemms, gflop, sections, actor_end, early, critic_end, disc_end = (float(x) if "." in x else int(x) for x in m.groups()[:9])
    assert version >= 5 and sections == 10
    # (round 4: the actor's three 32-wide head GEMMs + loss + column sum are one launch, addhip_actor_head: 55 -> 52 launches, 33 -> 30 GEMMs;
    #  its 1.6 GFLOP are no longer GEMM descriptors of the plan)
    assert launches == 52 and gemms == 30 and abs(gflop - 470.3) < 0.1, out
    assert 0 < early < actor_end < critic_end < disc_end == launches
    assert m.group(10) == "addhip_gemm_f32" and m.group(11) in ("addhip_slab_reduce", "addhip_slab_reduce_pair")
    assert "refused=1" in out and "workspace holds" in out


import pytest


@pytest.mark.gpu
def test_cpp_host_without_pytorch_runs_the_step_on_the_gpu(tmp_path):
    """tests/c_host/step_host.cpp: hipMalloc'ed buffers, the loss sections run directly and from the recorded plan under the four-stream
    schedule with a bucket call-back; the two runs' gradients agree (float atomics aside) and the buckets arrive in issue order."""
    import add_gym_amd._lib as L

    libdir = os.path.dirname(L.LIB_PATH)
    assert os.path.exists(L.LIB_PATH)
    exe = str(tmp_path / "step_host")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I" + INC, os.path.join(ROOT, "tests", "c_host", "step_host.cpp"), "-o", exe, "-L" + libdir, "-laddhip",
                    "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    m = re.search(r"launches=(\d+) buckets=(\d+) order=(\d+),(\d+),(\d+),(\d+) grad_scale=([\d.e+-]+) worst_rel_diff=([\d.e+-]+)", r.stdout)
    assert m, r.stdout
    # four exchange buckets, reported where they become final: actor tail, critic tail, the two first layers (3), the discriminator (2)
    assert int(m.group(1)) >= 50 and int(m.group(2)) == 4 and [int(m.group(i)) for i in (3, 4, 5, 6)] == [0, 1, 3, 2]
    assert float(m.group(7)) > 0 and float(m.group(8)) < 1e-4
