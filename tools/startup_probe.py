"""GPU-box helper: where the start-up second of a command-line run goes (interpreter, numpy, torch import, device bring-up
through torch, the same through the HIP runtime alone)."""
import subprocess
import sys
import time

SNIPPETS = {
    "python -c pass": "pass",
    "import numpy": "import numpy",
    "import torch": "import torch",
    "import torch + cuda init + 1 MiB alloc": "import torch; torch.cuda.set_device(0); torch.empty(1<<20, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize()",
    "+ load libmprg_hip.so": "import torch, ctypes; torch.cuda.set_device(0); torch.empty(1<<20, dtype=torch.uint8, device='cuda'); ctypes.CDLL('make_prg_amd/_lib/libmprg_hip.so').mprg_device_cus()",
    "HIP runtime alone: hipInit + hipMalloc + libmprg_hip.so": (
        "import ctypes; h=ctypes.CDLL('libamdhip64.so'); h.hipInit(0); p=ctypes.c_void_p(); h.hipMalloc(ctypes.byref(p), 1<<20); "
        "h.hipDeviceSynchronize(); ctypes.CDLL('make_prg_amd/_lib/libmprg_hip.so').mprg_device_cus()"),
    "HIP runtime alone + 2 GiB pinned": (
        "import ctypes; h=ctypes.CDLL('libamdhip64.so'); h.hipInit(0); p=ctypes.c_void_p(); h.hipMalloc(ctypes.byref(p), 1<<20); "
        "q=ctypes.c_void_p(); h.hipHostMalloc(ctypes.byref(q), ctypes.c_size_t(2<<30), 0); h.hipDeviceSynchronize()"),
}
for name, code in SNIPPETS.items():
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        rc = subprocess.run([sys.executable, "-c", code]).returncode
        best = min(best, time.time() - t0)
    print(f"{best:6.2f} s  rc {rc}  {name}", flush=True)
