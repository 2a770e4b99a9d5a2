"""Load a code object built from hand-written LLVM IR (tools/slp_repro/pk_repro.ll -> llc -> lld, see pk_repro.sh) and launch its kernel
through the HIP module API: kernel(const float* in, float* out), one workgroup of 64 lanes per 64 rows, IN floats in / OUT floats out
per lane.  Prints the output rows next to what numpy says they should be.

    python tools/slp_repro/run_hsaco.py <file.hsaco> <kernel name> <floats in per lane> <floats out per lane>
"""
import ctypes as C
import sys

import numpy as np
import torch


def launch(hsaco, name, x, n_out):
    hip = C.CDLL("libamdhip64.so")
    mod, fn = C.c_void_p(), C.c_void_p()
    assert hip.hipModuleLoad(C.byref(mod), hsaco.encode()) == 0, "hipModuleLoad"
    assert hip.hipModuleGetFunction(C.byref(fn), mod, name.encode()) == 0, "hipModuleGetFunction"
    rows = x.shape[0]
    assert rows % 64 == 0
    d_in = torch.tensor(x, device="cuda")
    d_out = torch.zeros(rows, n_out, device="cuda")
    a0, a1 = C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr())
    params = (C.c_void_p * 2)(C.cast(C.byref(a0), C.c_void_p), C.cast(C.byref(a1), C.c_void_p))
    rc = hip.hipModuleLaunchKernel(fn, rows // 64, 1, 1, 64, 1, 1, 0, None, params, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return d_out.cpu().numpy()


if __name__ == "__main__":
    hsaco, name, n_in, n_out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    x = np.random.RandomState(0).uniform(-1, 1, (64, n_in)).astype(np.float32)
    np.set_printoptions(precision=5, suppress=True, linewidth=200)
    print(launch(hsaco, name, x, n_out)[:4])
