"""k_conv_wgrad2 hands its tiles from the loader waves to the compute waves through counters in LDS (MI_WGRAD_FLAGS=1, the default)
instead of a workgroup barrier per tile (MI_WGRAD_FLAGS=0).  The counters rest on an exact per-wave request count and on LDS-DMA data
being visible once vmcnt has retired; a miscount would be masked by the barrier form and show as a wrong result (or a hang) in the
counter form.  The library reads the switch once, so the two forms run in two child processes on the shapes where a count could go
wrong -- tile counts that are not a multiple of 8, fewer splits than 8, ragged border tiles, 1x1 pairs with four tiles in flight,
the in-place class gather of a k3 s2 conv -- and must agree bit for bit (same kernel arithmetic, same summation order).
k_conv_wgrad3 (the rolling x halo of the k3 s1 layers, MI_WGRAD_ROLL=1, the default) deals the tiles to the splits in another order:
it is switched off for the bit-wise comparison and compared with k_conv_wgrad2 by value (fp32 partial sums added in another order;
its parity against the fp32 reference is tests/test_kernels_gpu.py's, which run it by default)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import torch
from medical_image_generation_amd import hipops as ops
dev = torch.device("cuda")
out = {}
cases = [(1, 32, 32, (12, 24, 24), 3, 1), (1, 64, 96, (5, 9, 11), 3, 1), (2, 96, 32, (4, 8, 16), 3, 1), (1, 256, 256, (8, 8, 8), 3, 1),
         (1, 64, 64, (6, 10, 12), 3, 2), (2, 32, 32, (24, 40, 36), 3, 2), (1, 192, 64, (4, 8, 8), 1, 1), (1, 32, 96, (4, 8, 9), 1, 1)]
for n, cin, cout, dims, k, s in cases:
    g = torch.Generator(device="cpu").manual_seed(sum(dims) + cin)
    x = torch.randn((n,) + dims + (cin,), generator=g).to(dev, torch.bfloat16)
    w = (torch.randn((cout, cin, k, k, k), generator=g) / (cin * k ** 3) ** 0.5).to(dev)
    plan = ops.ConvPlan(n, dims, cin, cout, (k,) * 3, (s,) * 3, (k // 2,) * 3)
    plan.pack(w)
    dy = torch.randn((n,) + plan.out_dims + (cout,), generator=g).to(dev, torch.bfloat16)
    dw, cs = torch.zeros_like(w), torch.zeros(cout, device=dev)
    plan.wgrad(x, dy, dw, colsum=cs)
    torch.cuda.synchronize()
    out[f"{cin}->{cout} {dims} k{k}s{s}"] = [hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()[:16], float(dw.abs().sum()), float(cs.abs().sum()),
                                            dw.flatten()[::53][:256].cpu().tolist(), float(dw.abs().max())]
print("RESULT " + json.dumps(out))
"""


def _run(flags, roll=0):
    env = dict(os.environ, MI_WGRAD_FLAGS=str(flags), MI_WGRAD_ROLL=str(roll))
    r = subprocess.run([sys.executable, "-c", _CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_wgrad_counter_handoff_equals_barrier_handoff():
    a, b = _run(1), _run(0)
    assert a.keys() == b.keys()
    for k in a:
        # weight gradients: bit-identical (the hash); the column sums of the 1x1 / strided pairs end in fp32 atomics: compared by value
        assert a[k][0] == b[k][0] and a[k][1] > 0, f"{k}: counters {a[k]} vs barrier {b[k]}"
        assert abs(a[k][2] - b[k][2]) <= 1e-5 * max(abs(b[k][2]), 1e-6), f"{k}: column sums differ"


def test_rolling_halo_wgrad_equals_plain_wgrad_by_value():
    a, b = _run(1, roll=1), _run(1, roll=0)
    assert a.keys() == b.keys()
    for k in a:
        tol = 2e-5 * b[k][4] + 1e-7  # fp32 sums of the same bf16 products in another order
        worst = max(abs(u - v) for u, v in zip(a[k][3], b[k][3]))
        assert worst <= tol, f"{k}: rolling {worst} > {tol}"
        assert abs(a[k][1] - b[k][1]) <= 1e-5 * b[k][1] and abs(a[k][2] - b[k][2]) <= 1e-5 * max(abs(b[k][2]), 1e-6), k
