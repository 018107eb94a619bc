"""Probe: the k3 s1 32->32 forward with bias rows + residual must give every sample of a batch of identical samples the batch-1 result
bit for bit, with and without the GroupNorm-sum epilogue, dense tensors and channel-slice views (voxel pitch 64)."""
import sys, math, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
d = 64
g = torch.Generator().manual_seed(1)
x1 = torch.randn(1, d, d, d, 32, generator=g).to(dev, torch.bfloat16)
r1 = torch.randn(1, d, d, d, 32, generator=g).to(dev, torch.bfloat16)
w = (torch.randn(32, 32, 3, 3, 3, generator=g) / math.sqrt(27 * 32)).to(dev)
av1 = torch.randn(1, 32, generator=g).to(dev)
for sums in (False, True):
    for views in (False, True):
        outs = {}
        for n in (1, 2):
            plan = ops.ConvPlan(n, (d, d, d), 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)); plan.pack(w)
            x, r, av = x1.repeat(n, 1, 1, 1, 1), r1.repeat(n, 1, 1, 1, 1), av1.repeat(n, 1)
            out = None
            if views:
                rb = torch.zeros(n, d, d, d, 64, device=dev, dtype=torch.bfloat16); rb[..., 32:] = r; r = rb[..., 32:]
                ob = torch.zeros(n, d, d, d, 96, device=dev, dtype=torch.bfloat16); out = ob[..., :32]
            y = plan.fwd(x, addvec=av, res=r, out=out, want_sums=sums)
            y = y[0] if sums else y
            torch.cuda.synchronize()
            outs[n] = y.contiguous()
            print(f"sums={sums} views={views} n={n}: samples equal {[bool(torch.equal(outs[n][i], outs[n][0])) for i in range(n)]}  == batch 1: "
                  f"{bool(torch.equal(outs[n][0:1], outs[1]))} ({int((outs[n][0:1] != outs[1]).sum())} differ)", flush=True)
# statistics from the conv's sums vs the statistics pass over the stored tensor, at a size where a workgroup's run has several tiles
for n in (1, 2):
    for use_res in (False, True):
        plan = ops.ConvPlan(n, (d, d, d), 32, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)); plan.pack(w)
        x, r, av = x1.repeat(n, 1, 1, 1, 1), r1.repeat(n, 1, 1, 1, 1), av1.repeat(n, 1)
        y, sums = plan.fwd(x, addvec=av, res=r if use_res else None, want_sums=True)
        gamma, beta = torch.ones(32, device=dev), torch.zeros(32, device=dev)
        ref = ops.gn_stats(y, 32, 1e-6, gamma, beta)
        got = ops.gn_stats_from_sums(sums, None, n, d ** 3, 32, 1e-6, gamma, beta)
        e1 = float((got.scale_shift - ref.scale_shift).abs().max() / ref.scale_shift.abs().max())
        print(f"stats n={n} res={use_res}: max rel err of scale/shift {e1:.3e}", flush=True)
