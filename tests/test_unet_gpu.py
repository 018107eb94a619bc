"""End-to-end parity of the HIP DiffusionModelUNet against (a) the golden vectors produced by the reference's own
model file and (b) the CPU oracle, on identical weights and inputs.

Tolerances (bf16 activations vs an fp32 reference; SURVEY 8c measured torch's own bf16-autocast drift of the
reference at 2.1e-2 rel-L2 on the prediction and 1.5e-2 on the global gradient):
  prediction / input gradient : rel-L2 <= 3e-2
  global parameter gradient   : rel-L2 <= 4e-2
  per-tensor gradients        : rel-L2 <= 0.15 for tensors that carry signal (norm > 1e-3 of the largest)"""
import math

import pytest
import torch

from oracle import cases, nets, synth

pytestmark = pytest.mark.gpu
S = cases.SEED
# the last four are the BASELINE configs on their exact kwargs (C2/C4 net at 32^3 x 2 and at 24x40x48; C3b and C5 latent nets)
# `unet2d_updown`: resblock_updown=True (avg-pool / nearest resnet resamplers)
# `unet2d_xattn`: with_conditioning=True (SpatialTransformer: self- + cross-attention on a context, GEGLU feed-forward)
HIP_CASES = ["unet3d", "unet_ldm", "unet_c1", "unet_c4", "unet_c4_np2", "unet_c3b", "unet_c5", "unet2d_updown", "unet2d_xattn"]


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def build(name):
    from medical_image_generation_amd.unet import DiffusionModelUNet
    c = cases.UNET_CASES[name]
    ref = nets.DiffusionModelUNet(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd)
    net = DiffusionModelUNet(**c["kwargs"])
    assert sorted(net.state_dict()) == sorted(sd)
    net.load_state_dict(sd)  # strict
    return c, ref, net.cuda()


@pytest.mark.parametrize("name", HIP_CASES)
def test_unet_forward_backward_matches_reference(golden, name):
    g, meta = golden(name)
    c, ref, net = build(name)
    x = synth.tensor(S, "x", c["shape"])
    t = torch.tensor(c["timesteps"])
    gy = synth.tensor(S, "grad_out", g["pred"].shape)
    xd = x.cuda().requires_grad_(True)
    extra = {"context": synth.tensor(S, "context", c["context"])} if "context" in c else {}
    pred = net(xd, t.cuda(), **{k: v.cuda() for k, v in extra.items()})
    assert pred.dtype == torch.float32 and pred.shape == g["pred"].shape
    e_pred = rel_l2(pred.detach().cpu(), g["pred"])
    pred.backward(gy.cuda())
    e_dx = rel_l2(xd.grad.cpu(), g["dx"])
    # oracle gradients (full tensors) for the per-parameter comparison
    xr = x.clone().requires_grad_(True)
    ref(xr, t, **extra).backward(gy)
    rg = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    hg = {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}
    assert sorted(hg) == sorted(rg) == meta["grad_names"].split("\n")  # proj_attn.* grad-less on both sides
    flat_r = torch.cat([rg[n].flatten() for n in sorted(rg)])
    flat_h = torch.cat([hg[n].flatten() for n in sorted(rg)])
    e_glob = rel_l2(flat_h, flat_r)
    big = max(float(v.norm()) for v in rg.values())
    worst = max(((rel_l2(hg[n], rg[n]), n) for n in rg if float(rg[n].norm()) > 1e-3 * big), default=(0, ""))
    print(f"\n[{name}] rel-L2: pred {e_pred:.3e}  dx {e_dx:.3e}  grads(global) {e_glob:.3e}  worst tensor {worst[0]:.3e} ({worst[1]})")
    assert e_pred <= 3e-2 and e_dx <= 3e-2 and e_glob <= 4e-2 and worst[0] <= 0.15


def test_unet_pristine_output_is_zero_and_state_dict_roundtrips():
    from medical_image_generation_amd.unet import DiffusionModelUNet
    c = cases.UNET_CASES["unet3d"]
    net = DiffusionModelUNet(**c["kwargs"]).cuda()
    y = net(synth.tensor(S, "x", c["shape"]).cuda(), torch.tensor(c["timesteps"]).cuda())
    assert float(y.detach().abs().max()) == 0.0  # zero_module'd output conv (UNet:1934)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net2 = DiffusionModelUNet(**c["kwargs"])
    net2.load_state_dict(sd)
    assert all(torch.equal(net2.state_dict()[k].cpu(), sd[k].cpu()) for k in sd)
    ref = nets.DiffusionModelUNet(**c["kwargs"])
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}


def test_unet_errors():
    from medical_image_generation_amd.unet import DiffusionModelUNet
    with pytest.raises(ValueError):
        DiffusionModelUNet(3, 1, 1, num_channels=(30, 64), attention_levels=(False, False))
    with pytest.raises(ValueError):
        DiffusionModelUNet(3, 1, 1, num_channels=(32, 64), attention_levels=(False,))
    c = cases.UNET_CASES["unet3d"]
    net = DiffusionModelUNet(**c["kwargs"]).cuda()
    with pytest.raises(ValueError):
        net(torch.zeros(1, 2, 8, 8, 8, device="cuda"), torch.zeros(1, dtype=torch.long, device="cuda"))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 8, 8, 8, device="cuda"), torch.zeros(1, 1, dtype=torch.long, device="cuda"))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 8, 8, 8), torch.zeros(1, dtype=torch.long))


def test_batch_consistency_at_realistic_size():
    """Size-independent property at a size where HBM latency is real (the 8^3-32^3 parity cases cannot see a kernel that consumes an
    asynchronous load too early): the BASELINE C4 net at 64^3 on a batch of two IDENTICAL samples must give each sample the
    batch-1 prediction bit for bit, a finite loss equal to the batch-1 loss, and the same parameter gradients (mean-reduced loss)."""
    import bench
    from medical_image_generation_amd.trainer import DDPMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet
    torch.manual_seed(3)
    net = DiffusionModelUNet(**bench.C4)
    for p in net.parameters():
        if float(p.detach().abs().max()) == 0:
            torch.nn.init.normal_(p, std=0.02)
    net = net.cuda()
    d = 64
    x1 = bench.synthetic_volume((1, 1, d, d, d), 5, torch.device("cuda"))
    n1 = torch.randn((1, 1, d, d, d), device="cuda")
    t1 = torch.tensor([417], device="cuda")
    x2, n2, t2 = x1.repeat(2, 1, 1, 1, 1), n1.repeat(2, 1, 1, 1, 1), t1.repeat(2)
    with torch.no_grad():
        y1 = net(x1, t1)
        y2 = net(x2, t2)
    assert torch.isfinite(y2).all()
    # (Bit-identity across batch sizes is a canary, not a structural guarantee: the conv epilogue's fp32 GroupNorm partial sums cover a
    # workgroup's run of tiles, which depends on the batch size.  It holds for the shipped kernels; a variant that added the residual
    # before the conv's rounding moved the sums enough to flip roundings, and this random-weight net amplifies single flips to 1-2e-2
    # (tools/diag/t_net_sens.py: statistics from the epilogue vs from the separate pass alone move the prediction by 1.3e-2).  If this
    # fails after a kernel change: MI_FUSE_GN_STATS=0 must make it pass again, and y2[0] == y2[1] must still hold.)
    assert torch.equal(y2[0:1], y2[1:2])
    assert torch.equal(y2[0:1], y1)
    tr = DDPMTrainer(net, lr=1e-4)
    tr.forward_backward(x1, n1, t1)
    l1, g1 = float(tr.loss), tr.arena.grad[:tr.arena.n_trainable].clone()
    tr.forward_backward(x2, n2, t2)
    l2, g2 = float(tr.loss), tr.arena.grad[:tr.arena.n_trainable].clone()
    assert math.isfinite(l2) and abs(l1 - l2) <= 1e-4 * abs(l1)  # (fp32 block sums of the loss in a different order)
    assert bool(torch.isfinite(g2).all())
    err = float((g2 - g1).norm() / g1.norm())
    print(f"\n[C4 @64^3] loss {l1:.6f} / {l2:.6f}, batch-2 vs batch-1 gradient rel-L2 {err:.3e}")
    assert err <= 1e-2  # same math; bf16 rounding of the halved output gradient and split-reduction order differ (measured 2.3e-3)


def test_class_embedding_matches_oracle():
    """num_class_embeds (UNet:1837-1839, 1975-1980): emb += class_embedding(class_labels); forward, input gradient and the embedding
    table's gradient (repeated labels accumulate) against the CPU restatement."""
    from medical_image_generation_amd.unet import DiffusionModelUNet
    c = cases.UNET_CASES["unet3d"]
    kw = dict(c["kwargs"], num_class_embeds=5)
    ref = nets.DiffusionModelUNet(**kw)
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, cases.SEED)
    ref.load_state_dict(sd)
    net = DiffusionModelUNet(**kw)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    net.load_state_dict(sd)
    net = net.cuda()
    x = synth.ellipsoid_volume(cases.SEED, "x", c["shape"])
    t = torch.tensor(c["timesteps"])
    labels = torch.tensor([3] * c["shape"][0])  # the same row for every sample: its gradient rows must accumulate
    xr = x.clone().requires_grad_(True)
    yr = ref(xr, t, class_labels=labels)
    g = synth.tensor(cases.SEED, "gy", tuple(yr.shape))
    yr.backward(g)
    xd = x.cuda().requires_grad_(True)
    y = net(xd, t.cuda(), class_labels=labels.cuda())
    y.backward(g.cuda())
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    e_y, e_dx = rel(y.detach().cpu(), yr.detach()), rel(xd.grad.cpu(), xr.grad)
    gw_ref = dict(ref.named_parameters())["class_embedding.weight"].grad
    gw = dict(net.named_parameters())["class_embedding.weight"].grad.cpu()
    e_w = rel(gw, gw_ref)
    print(f"\n[class embedding] prediction {e_y:.3e}, dx {e_dx:.3e}, d(class_embedding) {e_w:.3e}")
    assert e_y <= 3e-2 and e_dx <= 3e-2 and e_w <= 6e-2
    assert float(gw[[0, 1, 2, 4]].abs().max()) == 0.0  # untouched rows
    with pytest.raises(ValueError):
        net(xd, t.cuda())


def test_controlnet_residuals_match_oracle():
    """down_block_additional_residuals / mid_block_additional_residual (UNet:1995-2010): one tensor added to every skip and one to
    the middle block's output; prediction, input gradient and parameter gradients against the CPU restatement."""
    from medical_image_generation_amd.unet import DiffusionModelUNet
    c = cases.UNET_CASES["unet3d"]
    ref = nets.DiffusionModelUNet(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd)
    net = DiffusionModelUNet(**c["kwargs"])
    net.load_state_dict(sd)
    net = net.cuda()
    x, t = synth.tensor(S, "x", c["shape"]), torch.tensor(c["timesteps"])
    # skip shapes of this net at 16^3: conv_in + (resnet, downsampler) x 2 levels + the last level's resnet
    n = c["shape"][0]
    skip_shapes = [(n, 32, 16, 16, 16), (n, 32, 16, 16, 16), (n, 32, 8, 8, 8), (n, 64, 8, 8, 8), (n, 64, 4, 4, 4), (n, 64, 4, 4, 4)]
    down = [synth.tensor(S, f"ctrl{i}", s, 0.5) for i, s in enumerate(skip_shapes)]
    mid = synth.tensor(S, "ctrl_mid", (n, 64, 4, 4, 4), 0.5)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr, t, down_block_additional_residuals=down, mid_block_additional_residual=mid)
    gy = synth.tensor(S, "gy", tuple(yr.shape))
    yr.backward(gy)
    xd = x.cuda().requires_grad_(True)
    y = net(xd, t.cuda(), down_block_additional_residuals=[d.cuda() for d in down], mid_block_additional_residual=mid.cuda())
    y.backward(gy.cuda())
    rg = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    hg = {k: p.grad.cpu() for k, p in net.named_parameters() if p.grad is not None}
    flat = lambda d: torch.cat([d[k].flatten() for k in sorted(rg)])
    e_y, e_dx, e_g = rel_l2(y.detach().cpu(), yr.detach()), rel_l2(xd.grad.cpu(), xr.grad), rel_l2(flat(hg), flat(rg))
    y0 = net(x.cuda(), t.cuda())
    print(f"\n[ControlNet residuals] prediction {e_y:.3e}, dx {e_dx:.3e}, grads(global) {e_g:.3e}")
    assert e_y <= 3e-2 and e_dx <= 3e-2 and e_g <= 4e-2
    assert rel_l2(y0.detach().cpu(), yr.detach()) > 0.1  # the residuals matter: without them the prediction is far off
