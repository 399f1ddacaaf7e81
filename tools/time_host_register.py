#!/usr/bin/env python3
"""How fast is hipHostRegister on this box (would registering the caller's pageable arrays beat staging through a pinned
ring)?  Registers / unregisters numpy buffers of several sizes, then times a registered H2D copy.  GPU box only."""
import ctypes as C
import json
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
torch.zeros(1, device="cuda")
out = {}
for mb in (64, 1024, 8192):
    a = np.ones(mb << 20, dtype=np.uint8)
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0)
    t1 = time.perf_counter()
    d = torch.empty(a.nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rc2 = hip.hipMemcpy(d.data_ptr(), a.ctypes.data, a.nbytes, 1)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rc3 = hip.hipHostUnregister(a.ctypes.data)
    t4 = time.perf_counter()
    out[f"{mb} MiB"] = {"register_rc": rc, "register_GBps": round(a.nbytes / (t1 - t0) / 1e9, 2), "h2d_registered_GBps": round(a.nbytes / (t3 - t2) / 1e9, 2),
                        "unregister_GBps": round(a.nbytes / (t4 - t3) / 1e9, 2), "rcs": [rc2, rc3]}
    del d, a
# pageable copy for comparison
a = np.ones(1 << 30, dtype=np.uint8)
d = torch.empty(a.nbytes, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
hip.hipMemcpy(d.data_ptr(), a.ctypes.data, a.nbytes, 1)
torch.cuda.synchronize()
out["pageable_h2d_GBps"] = round(a.nbytes / (time.perf_counter() - t0) / 1e9, 2)
print(json.dumps(out))
