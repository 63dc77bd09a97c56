#!/usr/bin/env python
"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, HBM section: the counters are exact for some access widths, off by 2x for
others).  Each launch moves a buffer far larger than the 256 MiB Infinity Cache exactly once:
  - tfc_gather_rows with an identity index: one dword per lane, coalesced (the access shape of the
    conv kernels' weight stream and dw store)
  - torch copy_: 16 bytes per lane
Run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from matten_amd import _lib

dev = torch.device("cuda")
rows, d = 1 << 20, 256          # 1 GiB of fp32
src = torch.randn(rows, d, device=dev)
dst = torch.empty_like(src)
idx = torch.arange(rows, dtype=torch.int64, device=dev)
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    _lib.check(L.tfc_gather_rows(P(src), P(idx), rows, rows, d, P(dst), None))
    dst.copy_(src)
torch.cuda.synchronize()
print("bytes_per_launch", rows * d * 4)
