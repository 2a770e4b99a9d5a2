"""Is the SLP-VECTORIZED IR of rigid.hip wrong, or its lowering to packed fp32 instructions?  (round 4, third attempt)

hipcc's device pipeline taken apart by hand (the steps `hipcc -###` prints), so that the IR the vectorizer produced can be lowered under
a DIFFERENT subtarget than the one it was vectorized for:

    rigid.hip --(hipcc -O3 --cuda-device-only -emit-llvm)--> optimised device IR   [SLP decides here, packed fp32 ON]
              --(edit)--> e.g. "target-features" of every function: +packed-fp32-ops -> -packed-fp32-ops
              --(llc -O3)--> device object  --(lld)--> code object  --(clang-offload-bundler)--> fat binary
    rigid.hip --(hipcc --cuda-host-only -fcuda-include-gpubinary)--> rigid.o  --(link with the other objects)--> libaddhip_<variant>.so

The variants are BUILT in the build container (cross-compiling needs no GPU) under tools/slp_repro/_relower/ (git-ignored, travels to the GPU
box) and RUN on the box by gpu_case.py, one fresh process each:

    python tools/slp_repro/relower.py build          (here)
    python tools/slp_repro/relower.py run            (gpurun)   -> gpurun_out/slp_relower.log

Variants:
  noslp            -fno-slp-vectorize IR, lowered as is                              (control: ok)
  slp              vectorized IR, lowered as is                                      (control: WRONG, as the direct hipcc build)
  slp-nopk         vectorized IR, packed-fp32 feature removed before llc: every <2 x float> operation is split by instruction selection
  slp-scalarized   vectorized IR run through `opt -passes=scalarizer` (vector arithmetic rewritten to scalar IR, packed feature still on)
  slp-<llc flag>   vectorized IR, packed on, one code-generation option changed
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "add-gym_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "slp_repro", "_relower")
LLVM = "/opt/rocm/lib/llvm/bin"
HIPCC = "/opt/rocm/bin/hipcc"
BASE = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-function"]
OTHERS = [os.path.join(CSRC, f) for f in ("capi.o", "plan.o", "learner.o", "env_step.o", "gemm.o", "gemm_split.o", "gemm_bf16.o", "gemm_x3.o", "learn.o", "actor_head.o")]
SRC = os.path.join(CSRC, "rigid.hip")

# name -> (front-end flags, IR edit, llc flags)
VARIANTS = {
    "noslp": (["-fno-slp-vectorize"], None, []),
    "slp": ([], None, []),
    "slp-nopk": ([], "nopk", []),
    "slp-scalarized": ([], "scalarize", []),
    "slp-no-misched": ([], None, ["-enable-misched=false"]),
    "slp-no-postra-sched": ([], None, ["-enable-post-misched=false"]),
    "slp-O1-codegen": ([], None, ["-O1"]),
    "slp-global-isel": ([], None, ["-global-isel", "-global-isel-abort=2"]),
    "slp-no-sdwa-dpp": ([], None, ["-amdgpu-sdwa-peephole=false", "-amdgpu-dpp-combine=false"]),
    "slp-no-machine-licm": ([], None, ["-disable-machine-licm"]),
}
# Stage 2: stage 1 says the vectorized IR is right (scalarized by `opt`, it runs correctly) and SelectionDAG's handling of it is wrong
# (GlobalISel lowers the same IR, packed instructions and all, correctly; with the packed feature off SelectionDAG is still wrong).  Which
# vector construct?  ONE kind of vector instruction of rigid_step4_kernel<false> rewritten to scalar IR at a time (ir_unvector.py).
STAGE2 = {
    "slp": ([], None, []),
    "slp-no-dagcombine": ([], None, ["--combiner-disabled"]),
    "slp-unvec-select-icmp": ([], "unvec:select,icmp", []),
    "slp-unvec-fneg-fdiv": ([], "unvec:fneg,fbin", []),
    "slp-unvec-int-ops": ([], "unvec:ibin,zext", []),
    "slp-unvec-bitcast": ([], "unvec:bitcast", []),
    "slp-unvec-shuffle": ([], "unvec:shuffle", []),
    "slp-unvec-fadd-fsub-fmul": ([], "unvec:farith", []),
    "slp-unvec-all-but-arith": ([], "unvec:select,icmp,fneg,fbin,ibin,zext,bitcast,shuffle", []),
}


# Stage 4: the faulty lowering needs the whole kernel around it (the 8-instruction dataflow alone, pk_repro/, compiles correctly), which
# points behind instruction selection: which machine pass?  One switched off at a time.
STAGE4 = {
    "slp-no-subreg-liveness": ([], None, ["-enable-subreg-liveness=false"]),
    "slp-no-coalescer": ([], None, ["-join-liveintervals=false"]),
    "slp-regalloc-basic": ([], None, ["-vgpr-regalloc=basic"]),
    "slp-no-rewrite-partial-reg-uses": ([], None, ["-amdgpu-enable-rewrite-partial-reg-uses=false"]),
    "slp-no-machine-cse": ([], None, ["-disable-machine-cse"]),
    "slp-no-machine-sink": ([], None, ["-disable-machine-sink"]),
    "slp-no-peephole": ([], None, ["-disable-peephole"]),
    "slp-no-dce-in-ra": ([], None, ["-amdgpu-dce-in-ra=false"]),
    "slp-no-vgpr-liverange-opt": ([], None, ["-amdgpu-opt-vgpr-liverange=false"]),
    "slp-no-early-ifcvt": ([], None, ["-amdgpu-early-ifcvt=false", "-disable-early-ifcvt"]),
    "slp-no-pre-ra-opts": ([], None, ["-amdgpu-enable-pre-ra-optimizations=false"]),
    "slp-no-copyprop": ([], None, ["-disable-copyprop"]),
    "slp-no-scalar-ir-passes": ([], None, ["-amdgpu-scalar-ir-passes=false"]),
    "slp-no-cgp": ([], None, ["-disable-cgp"]),
    "slp-no-load-store-vectorizer": ([], None, ["-amdgpu-load-store-vectorizer=false"]),
}

# Stage 3: stage 2 says it is the <2 x float> fadd / fsub / fmul (963 in the kernel): scalarized alone they cure it, nothing else does, and
# disabling the DAG combiner does not.  Which of them?  All of them scalarized EXCEPT one window of the file order: a variant is wrong
# exactly when its window holds an instruction whose vector lowering is wrong.
N_ARITH = 963


def windows(first, last, parts):
    step = -(-(last - first) // parts)
    return [(a, min(a + step, last)) for a in range(first, last, step)]


def stage3(first=0, last=N_ARITH, parts=16):
    return {f"slp-vector-arith-only-{a}-{b}": ([], f"unvec:farith@{a}:{b}", []) for a, b in windows(first, last, parts)}


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-2000:])
    return r


def build(variants=VARIANTS, log="build.log"):
    os.makedirs(OUT, exist_ok=True)
    irs = {}
    report = []
    for name, (fe, edit, llc) in variants.items():
      try:
        key = " ".join(fe)
        if key not in irs:
            ll = os.path.join(OUT, "ir_" + ("noslp" if fe else "slp") + ".ll")
            run([HIPCC] + BASE + fe + ["--cuda-device-only", "-emit-llvm", "-S", SRC, "-o", ll])
            irs[key] = ll
        ll = irs[key]
        if edit == "nopk":
            txt = open(ll).read()
            # (the attribute lists the features clang names; packed fp32 is implied by "target-cpu"="gfx950": switch it off explicitly)
            txt, n = re.subn(r'("target-features"="[^"]*)"', r'\1,-packed-fp32-ops"', txt)
            assert n > 0
            ll = os.path.join(OUT, "ir_slp_nopk.ll")
            open(ll, "w").write(txt)
        elif edit and edit.startswith("unvec:"):
            dst = os.path.join(OUT, "ir_" + name + ".ll")
            kinds, _, keep = edit[6:].partition("@")
            r = run([sys.executable, os.path.join(ROOT, "tools", "slp_repro", "ir_unvector.py"), ll, dst, kinds, "rigid_step4_kernelILb0"] + ([keep] if keep else []))
            print("   rewritten:", r.stdout.strip(), flush=True)
            ll = dst
        elif edit == "scalarize":
            dst = os.path.join(OUT, "ir_slp_scalarized.ll")
            run([os.path.join(LLVM, "opt"), "-S", "-passes=scalarizer", ll, "-o", dst])
            ll = dst
        txt = open(ll).read()
        vec2 = len(re.findall(r"= (?:f(?:add|mul|sub)[^\n]*<2 x float>|[^\n]*@llvm\.fma\.v2f32|[^\n]*@llvm\.fmuladd\.v2f32)", txt))
        obj = os.path.join(OUT, name + ".dev.o")
        run([os.path.join(LLVM, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-filetype=obj", "-relocation-model=pic"] + (llc if any(f.startswith("-O") for f in llc) else ["-O3"] + llc) + [ll, "-o", obj])
        asm = os.path.join(OUT, name + ".s")
        if log.startswith("build3"):  # (many variants: no second code-generation run for the instruction counts)
            open(asm, "w").write("")
        else:
            run([os.path.join(LLVM, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-filetype=asm", "-relocation-model=pic"] + (llc if any(f.startswith("-O") for f in llc) else ["-O3"] + llc) + [ll, "-o", asm])
        body = open(asm).read()
        m = re.search(r"^_ZN\S*rigid_step4_kernelILb0EE\S*:[^\n]*\n(.*?)\n\.Lfunc_end", body, re.S | re.M)
        pk = len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32", m.group(1))) if m else -1
        insts = len([l for l in m.group(1).splitlines() if re.match(r"\s+[vs]_|\s+ds_|\s+global_|\s+buffer_", l)]) if m else -1
        co = os.path.join(OUT, name + ".hsaco")
        run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", co, obj])
        fb = os.path.join(OUT, name + ".hipfb")
        run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", "-input=" + co, "-output=" + fb])
        host = os.path.join(OUT, name + ".o")
        run([HIPCC] + BASE + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", SRC, "-o", host])
        lib = os.path.join(OUT, f"libaddhip_{name}.so")
        run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, host] + OTHERS)
        for f in (obj, co, fb, host):
            os.remove(f)
        report.append(f"{name:22s} <2 x float> arithmetic in IR {vec2:5d}   v_pk_* in rigid_step4_kernel<false> {pk:4d} of {insts} instructions")
        print(report[-1], flush=True)
      except RuntimeError as e:  # (some switches crash this llc: reported, not run)
        msg = "the compiler crashed" if "PrintStackTrace" in str(e) else str(e).splitlines()[-1][:120]
        report.append(f"{name:22s} NOT BUILT: {msg}")
        print(report[-1], flush=True)
    open(os.path.join(OUT, log), "w").write("\n".join(report) + "\n")


def run_all(variants=VARIANTS, log="build.log", out_name="slp_relower.log"):
    out = os.path.join(ROOT, "gpurun_out", out_name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    build_log = dict(l.split(None, 1) for l in open(os.path.join(OUT, log)).read().splitlines())
    with open(out, "w") as f:
        for name in variants:
            lib = os.path.join(OUT, f"libaddhip_{name}.so")
            if not os.path.exists(lib):
                line = f"{name:22s} not built | {build_log.get(name, '').strip()}"
                print(line, flush=True)
                f.write(line + "\n")
                continue
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slp_repro", "gpu_case.py"), lib], capture_output=True, text=True, timeout=300)
            cases = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
            verdict = "WRONG" if any("WRONG" in c for c in cases) else ("ok" if len(cases) == 2 else "FAILED " + r.stderr[-300:])
            line = f"{name:22s} {verdict:6s} | {build_log.get(name, '').strip()} | " + " ; ".join(re.sub(r"^CASE (\w+) \w+ worst (\S+).*", r"\1 \2", c) for c in cases)
            print(line, flush=True)
            f.write(line + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 2 and ":" in sys.argv[2]:  # explicit windows  a:b,c:d,...
        s3 = {f"slp-vector-arith-only-{w.replace(':', '-')}": ([], f"unvec:farith@{w}", []) for w in sys.argv[2].split(",")}
        tag = "w" + str(abs(hash(sys.argv[2])) % 100000) if len(sys.argv) < 4 else sys.argv[3]
    else:
        s3 = stage3(*(int(x) for x in sys.argv[2:5])) if len(sys.argv) > 2 else stage3()
        tag = "_".join(sys.argv[2:5])
    {"build": build, "run": run_all, "build2": lambda: build(STAGE2, "build2.log"),
     "run2": lambda: run_all(STAGE2, "build2.log", "slp_relower2.log"),
     "build4": lambda: build(STAGE4, "build3_stage4.log"), "run4": lambda: run_all(STAGE4, "build3_stage4.log", "slp_relower4.log"),
     "build3": lambda: build(s3, f"build3_{tag}.log"), "run3": lambda: run_all(s3, f"build3_{tag}.log", f"slp_relower3_{tag}.log")}[sys.argv[1]]()
