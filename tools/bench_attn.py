import math, sys, torch
sys.path.insert(0, "/root/repo")
from medical_image_generation_amd._lib import call, call_raw, ptr
dev = torch.device("cuda")
# usage: python tools/bench_attn.py [B H S d]     (default: the C4 attention level; "1 1 8000 512" = the latent UNet's 20^3 level)
B, H, S, d = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (1, 4, 4096, 64)
C = H * d
sc = 1 / math.sqrt(d)
qkv = (torch.randn(B * S, 3 * C) * 1.0).bfloat16().to(dev)
x = torch.randn(B, S, C).bfloat16().to(dev); dy = torch.randn(B, S, C).bfloat16().to(dev)
y = torch.empty_like(x); lse = torch.empty(B * H, S, device=dev); dsum = torch.empty(B * H, S, device=dev); dqkv = torch.empty_like(qkv)
nws = call_raw("mi_attn_workspace_bytes", C, H, B, S)
ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
for use in (False, True):
    w, n = (ptr(ws), nws) if use else (None, 0)
    def f(): call("mi_attn_fwd", ptr(qkv), 3 * C, C, H, B, S, sc, ptr(x), ptr(y), ptr(lse), w, n)
    def g(): call("mi_attn_bwd", ptr(qkv), 3 * C, C, H, B, S, sc, ptr(y), ptr(x), ptr(dy), ptr(lse), ptr(dsum), ptr(dqkv), w, n)
    for fn, name in ((f, "fwd"), (g, "bwd")):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        fl = 4 * B * H * S * S * d * (1 if name == "fwd" else 2.5)  # algorithmic: 2 products forward, 5 backward
        print(f"B={B} H={H} S={S} d={d} split={use} nws={n} {name}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s (algorithmic)", flush=True)
