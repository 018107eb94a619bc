"""Does the 256 MiB Infinity Cache help the streaming GroupNorm passes?  Times mi_gn_apply (1 read + 1 write per element) and mi_gn_bwd
(partial: 2 reads; apply: 2-3 reads + 1 write) back to back on ONE tensor at sizes whose working set is below / above 256 MiB, and the
apply pass right after a kernel that wrote its input (the situation in the step) against the same pass after the cache was flushed by
a 1 GiB fill.  usage: python tools/diag/mall_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_image_generation_amd import hipops as ops

dev = torch.device("cuda")
C, G = 32, 32
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB


def timed(fn, iters=10, flush=False):
    st = torch.cuda.current_stream()
    tot = 0.0
    for _ in range(iters):
        if flush:
            trash.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3  # us


for d in (64, 96, 112, 128, 144, 160):
    x = torch.randn((1, d, d, d, C), device=dev).to(torch.bfloat16)
    g = torch.randn_like(x)
    st = ops.gn_stats(x, G, 1e-6, gamma, beta)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    mb = x.numel() * 2 / 1e6
    for _ in range(3):
        ops.gn_apply(x, st, True), ops.gn_bwd(g, x, st, gamma, True, dg, db)
    a_hot, a_cold = timed(lambda: ops.gn_apply(x, st, True)), timed(lambda: ops.gn_apply(x, st, True), flush=True)
    b_hot, b_cold = timed(lambda: ops.gn_bwd(g, x, st, gamma, True, dg, db)), timed(lambda: ops.gn_bwd(g, x, st, gamma, True, dg, db), flush=True)
    s_hot, s_cold = timed(lambda: ops.gn_stats(x, G, 1e-6, gamma, beta)), timed(lambda: ops.gn_stats(x, G, 1e-6, gamma, beta), flush=True)
    print(f"{d}^3 x {C}ch = {mb:6.1f} MB/tensor | apply (2 passes) hot {a_hot:7.1f} us = {2 * mb / a_hot:5.2f} TB/s, cold {a_cold:7.1f} us = "
          f"{2 * mb / a_cold:5.2f} | stats (1 pass) hot {s_hot:6.1f} us = {mb / s_hot:5.2f} TB/s, cold {s_cold:6.1f} = {mb / s_cold:5.2f} | "
          f"bwd (5 passes) hot {b_hot:7.1f} us = {5 * mb / b_hot:5.2f} TB/s, cold {b_cold:7.1f} us = {5 * mb / b_cold:5.2f}", flush=True)
