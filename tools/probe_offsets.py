#!/usr/bin/env python3
"""Does the relative placement of two same-shape fields matter?  cdfpsi's vertical integral reads
u(i,j,k) and e3u(i,j,k) at the same offsets of two arrays at the same time; carve both out of one
buffer with a varying gap and time the kernel (eORCA12 size).  One JSON line per gap."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdftools_amd import hip  # noqa: E402

npi, npj, npk = 4322, 3606, 75
n3 = npi * npj * npk
dev = torch.device("cuda", 0)
e2 = torch.rand(npj, npi, device=dev) + 1
p = hip.PsiPlan(npi, npj, npk)
gaps = [0, 64, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 4096 + 256, 1 << 20, (1 << 20) + 65536 + 4096 + 256, (1 << 21) + 333 * 256]
big = torch.empty(2 * n3 + max(gaps) // 4 + 1024, device=dev)
big.uniform_(0.5, 1.5)
for gap in gaps:
    a0 = (-big.data_ptr() // 4) % 4                      # 16-byte aligned start
    u = big[a0:a0 + n3].view(npk, npj, npi)
    b0 = a0 + n3 + gap // 4
    b0 += (-(big.data_ptr() + 4 * b0) // 4) % 4
    e3 = big[b0:b0 + n3].view(npk, npj, npi)
    delta = e3.data_ptr() - u.data_ptr()
    p.integrate(zu=u, e2u=e2, e3u=e3)
    torch.cuda.synchronize()
    tm = hip.Timer()
    tm.start()
    for _ in range(10):
        p.integrate(zu=u, e2u=e2, e3u=e3)
    tm.stop()
    ms = tm.elapsed_ms() / 10
    print(json.dumps({"gap": gap, "delta_bytes": delta, "delta_mod_2MiB": delta % (1 << 21), "ms": round(ms, 4),
                      "GBps": round(8.0 * n3 / ms / 1e6, 1)}), flush=True)
