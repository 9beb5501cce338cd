"""One shape of the depthwise convolution, a few eager launches: the program to put under rocprofv3 --pmc."""
import sys
import os

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from v2a_amd import _lib as L

B, N, d = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 782, 1024)))
tn = int(sys.argv[4]) if len(sys.argv) > 4 else 0
x = torch.randn(B, N, d, device="cuda")
out = torch.empty_like(x)
wt = torch.randn(31, d, device="cuda")
bias = torch.randn(d, device="cuda")
L.set_tuning(dwconv_rows_per_wave=tn)
for _ in range(6):
    L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31)
torch.cuda.synchronize()
print("done", B, N, d, tn)
