"""GPU side of the SLP-miscompile investigation (runs ON the GPU box: hipcc is there).  Builds csrc/rigid.hip under a list of flag sets,
links each with the already-built objects of the other sources into a private libaddhip variant, and runs gpu_case.py on it in a fresh
process.  One line per variant: flags, packed-fp32 instruction count of the four-lane kernel, verdict against the float64 oracle.

    python tools/slp_repro/gpu_bisect.py [stage]     -> gpurun_out/slp_bisect_<stage>.log
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "add-gym_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
BASE = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-function"]
OTHERS = [os.path.join(CSRC, f) for f in ("capi.o", "plan.o", "learner.o", "env_step.o", "gemm.o", "gemm_split.o", "gemm_bf16.o", "gemm_x3.o", "learn.o", "actor_head.o")]

# Stage 2: the default -O3 build (SLP on: wrong) with ONE region of rigid_step4_kernel fenced off from the vectorizer -- an empty
# `asm volatile` on the values that leave the region makes them opaque, so no SLP tree can span it.  (text anchor in rigid.hip,
# replacement): the patched copy is compiled from a temporary file; the product source is not touched.
F = 'asm volatile("" : "+v"(%s));'
FENCES = {
    "cholesky factorisation": ("          A[a][j] = (j == a) ? sqrtf(fmaxf(sum, 1e-20f)) : sum / A[a][a];",
                               "          " + F % "sum" + "\n          A[a][j] = (j == a) ? sqrtf(fmaxf(sum, 1e-20f)) : sum / A[a][a];"),
    "forward / back substitution": ("        b[a] = sum / A[a][a];", "        " + F % "sum" + "\n        b[a] = sum / A[a][a];"),
    "root: sum of the chains' inertias": ("      art_add_array(I, p, tot);", "      for (int x = 0; x < 27; ++x) { " + F % "tot[x]" + " }\n      art_add_array(I, p, tot);"),
    "pass 2: joint terms (Dinv, u)": ("        const float u = tau - comp(p.a, ax);", "        float u = tau - comp(p.a, ax);\n        " + F % "u"),
    "pass 2: hand-over to the parent (carry)": ("        art_to_array(P, pp, carry);", "        art_to_array(P, pp, carry);\n        for (int x = 0; x < 27; ++x) { " + F % "carry[x]" + " }"),
    "pass 3: joint acceleration": ("        caa = aa; cal = al;\n        const float qdn",
                                   "        caa = aa; cal = al;\n        " + " ".join(F % v for v in ("caa.x", "caa.y", "caa.z", "cal.x", "cal.y", "cal.z")) + "\n        const float qdn"),
    "root: a0 out of the solve": ("      for (int a = 0; a < 6; ++a) a0[a] = b[a];", "      for (int a = 0; a < 6; ++a) { a0[a] = b[a]; " + F % "a0[a]" + " }"),
    "pass 1: body velocity": ("        cw = w; cv = vl; cnz = nz;\n      }\n    }\n    // ---------------- pass 2 (inward)",
                              "        cw = w; cv = vl; cnz = nz;\n        " + " ".join(F % v for v in ("cw.x", "cw.y", "cw.z", "cv.x", "cv.y", "cv.z")) + "\n      }\n    }\n    // ---------------- pass 2 (inward)"),
}

STAGES = {
    "1": [("noslp", ["-fno-slp-vectorize"]), ("slp", []),
          ("slp, no packed fp32", ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]),
          ("slp, no horizontal reductions", ["-mllvm", "-slp-vectorize-hor=false"]),
          ("slp, O2", ["-O2"]),
          ] + [(f"slp-threshold={t}", ["-mllvm", f"-slp-threshold={t}"]) for t in (1, 2, 3, 5, 8, 16, 24, 40, 60)],
    "2": [("slp (control)", [])] + [("slp, fence: " + k, ["@fence", k]) for k in FENCES],
    # Stage 3 (round 4): is the VECTORIZED IR wrong, or its lowering to packed instructions?  With the packed-fp32 target feature off the
    # cost model rejects the trees, so the vectorizer is forced (negative threshold) and the <2 x float> operations are scalarised by
    # instruction selection: same IR-level decisions, no v_pk_*_f32.  `vec2` = <2 x float> arithmetic operations in the device IR of rigid.hip.
    "3": [("slp (control)", [])] +
         [(f"slp forced (threshold {t}), packed fp32 off", ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-mllvm", f"-slp-threshold={t}"]) for t in (-4, -20)] +
         [(f"slp forced (threshold {t}), packed fp32 on", ["-mllvm", f"-slp-threshold={t}"]) for t in (-4, -20)],
}


def ir_vec2(flags, tmp, tag):
    """<2 x float> fadd / fmul / fma / fsub operations in the device IR (how much the SLP vectorizer formed)."""
    ll = os.path.join(tmp, f"rigid_{tag}.ll")
    r = subprocess.run([HIPCC] + BASE + flags + ["--cuda-device-only", "-emit-llvm", "-S", os.path.join(CSRC, "rigid.hip"), "-o", ll], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    txt = open(ll).read()
    return len(re.findall(r"= (?:f(?:add|mul|sub)[^\n]*<2 x float>|[^\n]*@llvm\.fma\.v2f32|[^\n]*@llvm\.fmuladd\.v2f32)", txt))


def patched_source(key, tmp):
    old, new = FENCES[key]
    txt = open(os.path.join(CSRC, "rigid.hip")).read()
    assert txt.count(old) >= 1, key
    # the anchors of the one-lane kernel come first in the file where they exist twice: patch the LAST occurrence (four-lane kernel)
    i = txt.rindex(old)
    path = os.path.join(tmp, "rigid_" + re.sub(r"\W+", "_", key) + ".hip")
    open(path, "w").write(txt[:i] + new + txt[i + len(old):])
    return path


def kernel_stats(src_flags, tmp, tag):
    s = os.path.join(tmp, f"rigid_{tag}.s")
    r = subprocess.run([HIPCC] + BASE + src_flags + ["--cuda-device-only", "-S", os.path.join(CSRC, "rigid.hip"), "-o", s], capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr[-400:]
    txt = open(s).read()
    # the LDS form of the four-lane kernel: rigid_step4_kernelILb0EE
    m = re.search(r"^(_ZN\S*rigid_step4_kernelILb0EE\S*):\n(.*?)\n\s*s_endpgm", txt, re.S | re.M)
    body = m.group(2) if m else ""
    return dict(pk=len(re.findall(r"\bv_pk_\w+_f32", body)), instr=len([l for l in body.splitlines() if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")])), None


def main():
    stage = sys.argv[1] if len(sys.argv) > 1 else "1"
    variants = STAGES[stage]
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    log = open(os.path.join(out_dir, f"slp_bisect_{stage}.log"), "w")

    def say(*a):
        line = " ".join(str(x) for x in a)
        print(line, flush=True)
        log.write(line + "\n")
        log.flush()

    say(subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0])
    tmp = tempfile.mkdtemp(prefix="slp_gpu_")
    for i, (name, flags) in enumerate(variants):
        obj, so = os.path.join(tmp, f"rigid_{i}.o"), os.path.join(tmp, f"libaddhip_{i}.so")
        src = os.path.join(CSRC, "rigid.hip")
        if flags and flags[0] == "@fence":
            src, flags = patched_source(flags[1], tmp), []
        r = subprocess.run([HIPCC] + BASE + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            say(f"[{name}] flags {flags}: DOES NOT BUILD: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''}")
            continue
        subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950"] + OTHERS + [obj, "-o", so], check=True)
        st = None
        if stage == "3":
            ks, _ = kernel_stats(flags, tmp, str(i))
            st = dict(vec2_ir_ops=ir_vec2(flags, tmp, str(i)), **(ks or {}))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slp_repro", "gpu_case.py"), so], capture_output=True, text=True, timeout=600)
        cases = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
        say(f"[{name}] flags {flags}: four-lane kernel {st}")
        for c in cases:
            say("    " + c)
        if r.returncode != 0 or not cases:
            say("    run failed:", r.stderr[-600:])
    log.close()


if __name__ == "__main__":
    main()
