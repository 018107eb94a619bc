import torch
a, b, c, d = [torch.load(f"/tmp/{k}.pt") for k in "abcd"]
r = lambda u, v: float((u - v).norm() / v.norm())
print("compute-side vs helper-side residual, batch 1 (fused stats):", r(b["y1"], a["y1"]), " (separate stats):", r(c["y1"], d["y1"]))
print("fused vs separate statistics, batch 1 (helper-side):", r(a["y1"], d["y1"]), " (compute-side):", r(b["y1"], c["y1"]))
