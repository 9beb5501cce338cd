#!/usr/bin/env python3
"""Can a timing event be recorded BETWEEN the kernel nodes of a captured hipGraph on this ROCm build?  torch refuses
`Event(external=True)` on ROCm, so this goes to the HIP runtime directly (hipEventRecordWithFlags, hipEventRecordExternal)."""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
x = torch.randn(4096, 4096, device="cuda")
evs = []
def mark(stream):
    e = ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(e)) == 0
    rc = hip.hipEventRecordWithFlags(e, ctypes.c_void_p(stream.cuda_stream), 1)
    evs.append(e)
    return rc
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
y = x @ x
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    rcs = [mark(s)]
    y = x @ x
    rcs.append(mark(s))
    z = y.relu()
    rcs.append(mark(s))
print("record rc inside capture:", rcs)
for rep in range(3):
    g.replay()
    torch.cuda.synchronize()
    out = []
    for a, b in zip(evs[:-1], evs[1:]):
        ms = ctypes.c_float()
        rc = hip.hipEventElapsedTime(ctypes.byref(ms), a, b)
        out.append((rc, round(ms.value * 1e3, 2)))
    print("replay", rep, "intervals (rc, us):", out)
