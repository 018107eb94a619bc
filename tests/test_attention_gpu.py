"""Fused attention kernels (csrc/attention.hip) against a plain PyTorch fp32 reference of softmax(QK^T*scale)V + residual,
forward and backward, through the C ABI.  bf16 operands, fp32 softmax: |err| <= 2e-2 of the reference's max."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda")


# One wide head (csrc/attention_wide.hip): the reference's latent UNets (configuration.py:894, num_head_channels = [0, 512, 768]) at
# the token counts of 8^3 .. 20^3 latents, a ragged count (1000, 125, 27), two images, and 8 tokens (fewer than one tile)
WIDE = [(1, 1, 512, 512), (2, 1, 1000, 512), (1, 1, 4096, 512), (1, 1, 8000, 512), (1, 1, 125, 512), (1, 1, 1000, 768), (2, 1, 27, 768),
        (1, 1, 8, 768)]


@pytest.mark.parametrize("B,H,S,d", [(1, 1, 64, 32), (2, 2, 64, 32), (1, 4, 512, 64), (2, 1, 1000, 64), (1, 2, 4096, 64), (1, 3, 200, 32)] + WIDE)
@pytest.mark.parametrize("split", [True, False])
def test_flash_attention_fwd_bwd(B, H, S, d, split):
    """split: hand the kernels their scratch, so the reduction axis is cut into up to 8 partial passes + a merge kernel
    (2 / 4 / 8 ways for the S = 512 / 1000 / 4096 cases); without it the single-pass kernels run."""
    from medical_image_generation_amd._lib import call, call_raw, ptr
    C = H * d
    g = torch.Generator().manual_seed(S + d)
    qkv = (torch.randn(B, S, 3 * C, generator=g) * 1.5).bfloat16()
    x = torch.randn(B, S, C, generator=g).bfloat16()
    dy = torch.randn(B, S, C, generator=g).bfloat16()
    scale = 1 / math.sqrt(d)
    # reference (fp32 on the bf16-rounded operands)
    qr = qkv.float().clone().requires_grad_(True)
    q, k, v = (qr[..., i * C:(i + 1) * C].reshape(B, S, H, d).permute(0, 2, 1, 3) for i in range(3))
    att = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    o = (att @ v).permute(0, 2, 1, 3).reshape(B, S, C)
    y_ref = o + x.float()
    o.backward(dy.float())
    # HIP
    qd, xd, dyd = qkv.to(dev).reshape(B * S, 3 * C), x.to(dev), dy.to(dev)
    y = torch.empty_like(xd)
    lse = torch.empty(B * H, S, device=dev)
    nws = call_raw("mi_attn_workspace_bytes", C, H, B, S) if split else 0
    assert call_raw("mi_attn_supported", C, H) == 1
    assert nws > 0 or not split or S < 512
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    call("mi_attn_fwd", ptr(qd), 3 * C, C, H, B, S, scale, ptr(xd), ptr(y), ptr(lse), ptr(ws) if split else None, nws)
    err = float((y.float().cpu() - y_ref.detach()).abs().max())
    assert err <= 2e-2 * float(y_ref.abs().max()), f"fwd err {err}"
    lse_ref = torch.logsumexp((q @ k.transpose(-1, -2) * scale).detach(), dim=-1).reshape(B * H, S) / math.log(2)
    assert float((lse.cpu() - lse_ref).abs().max()) <= 2e-2
    dqkv = torch.zeros_like(qd)
    dsum = torch.empty(B * H, S, device=dev)
    call("mi_attn_bwd", ptr(qd), 3 * C, C, H, B, S, scale, ptr(y), ptr(xd), ptr(dyd), ptr(lse), ptr(dsum), ptr(dqkv),
         ptr(ws) if split else None, nws)
    ref = qr.grad.reshape(B * S, 3 * C)
    got = dqkv.float().cpu()
    for name, sl in (("dQ", slice(0, C)), ("dK", slice(C, 2 * C)), ("dV", slice(2 * C, 3 * C))):
        e = float((got[:, sl] - ref[:, sl]).abs().max())
        assert e <= 2.5e-2 * float(ref[:, sl].abs().max()), f"{name} err {e} vs {float(ref[:, sl].abs().max())}"
