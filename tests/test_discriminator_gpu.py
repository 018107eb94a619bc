"""PatchDiscriminator / PatchAdversarialLoss and the GAN step of train_autoencoder.py (T-AE:371-435) on the HIP path against the torch
restatement in oracle/disc.py (third-party classes: PARITY UNPINNED against upstream itself; the call sites and the planner's arguments
are the reference's)."""
import pytest
import torch

from oracle import cases, disc as odisc, nets, step, synth

pytestmark = pytest.mark.gpu
S = cases.SEED
DKW = dict(spatial_dims=3, num_channels=16, in_channels=1, out_channels=1, num_layers_d=3)  # the planner's arguments with 16 base channels


def _pair(kw=DKW, seed=S):
    from medical_image_generation_amd.discriminator import PatchDiscriminator
    ref = odisc.PatchDiscriminator(**kw)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    for k, v in sd.items():  # the reference initialisation, redrawn so that biases / BatchNorm offsets are not zero
        if v.dtype.is_floating_point and "running" not in k:
            sd[k] = v + 0.05 * torch.randn(v.shape, generator=g)
    ref.load_state_dict(sd)
    net = PatchDiscriminator(**kw)
    net.load_state_dict(sd)
    return ref, net.cuda()


def test_discriminator_forward_backward_matches_oracle():
    ref, net = _pair()
    assert list(net.state_dict()) == list(ref.state_dict())
    x = synth.ellipsoid_volume(S, "dx", (2, 1, 32, 32, 32))
    xr = x.clone().requires_grad_(True)
    outs_r = ref(xr)
    adv = odisc.PatchAdversarialLoss()
    loss_r = adv(outs_r[-1], True) + 0.5 * adv(outs_r[2], False)
    loss_r.backward()
    xh = x.cuda().requires_grad_(True)
    outs_h = net(xh)
    assert [tuple(o.shape) for o in outs_h] == [tuple(o.shape) for o in outs_r] == [(2, 16, 16, 16, 16), (2, 32, 8, 8, 8), (2, 64, 4, 4, 4),
                                                                                      (2, 128, 3, 3, 3), (2, 1, 2, 2, 2)]
    for i, (a, b) in enumerate(zip(outs_h, outs_r)):
        e = float((a.detach().cpu() - b.detach()).norm() / b.detach().norm())
        assert e <= 3e-2, f"layer {i}: rel-L2 {e:.3e}"
    from medical_image_generation_amd.discriminator import PatchAdversarialLoss
    advh = PatchAdversarialLoss()
    loss_h = advh(outs_h[-1], True) + 0.5 * advh(outs_h[2], False)
    assert abs(float(loss_h) - float(loss_r)) <= 2e-2 * abs(float(loss_r))
    loss_h.backward()
    e = float((xh.grad.cpu() - xr.grad).norm() / xr.grad.norm())
    cosx = float(torch.dot(xh.grad.cpu().flatten(), xr.grad.flatten()) / (xh.grad.norm().cpu() * xr.grad.norm()))
    print(f"\n[PatchDiscriminator] loss {float(loss_h):.5f} vs {float(loss_r):.5f}; input-gradient rel-L2 {e:.3e}, cosine {cosx:.4f}")
    pr, ph = dict(ref.named_parameters()), dict(net.named_parameters())
    for n in pr:
        en = float((ph[n].grad.cpu() - pr[n].grad).norm() / (pr[n].grad.norm() + 1e-12))
        print(f"    {n:28s} grad rel-L2 {en:.3e}  |g| {float(pr[n].grad.norm()):.3e}")
    g_ref = torch.cat([pr[n].grad.flatten() for n in pr])
    g_hip = torch.cat([ph[n].grad.cpu().flatten() for n in pr])
    eg = float((g_hip - g_ref).norm() / g_ref.norm())
    print(f"  parameter-gradient rel-L2 {eg:.3e}")
    # Four bf16 layers with BatchNorm (whose backward subtracts means: cancellation) and LeakyReLU kinks between them.  Measured in the
    # build container: torch's own bf16 autocast of the restatement against its fp32 self on this very case -- input gradient rel-L2
    # 9.6e-2, parameter gradient 7.3e-2; ours 8.9e-2 / 7.1e-2.  The budget is that drift, plus direction (cosine).
    cosp = float(torch.dot(g_hip, g_ref) / (g_hip.norm() * g_ref.norm()))
    assert e <= 1.5e-1 and cosx >= 0.99
    assert eg <= 1.2e-1 and cosp >= 0.99
    # BatchNorm buffers after ONE training-mode forward
    for k, v in ref.state_dict().items():
        if "running" in k:
            assert torch.allclose(net.state_dict()[k].cpu(), v, rtol=3e-2, atol=3e-3), k
        if "num_batches_tracked" in k:
            assert int(net.state_dict()[k]) == int(v) == 1


@pytest.mark.parametrize("graph", [False, True])
def test_gan_step_matches_oracle_composition(graph):
    """AEGANTrainer: generator step with the adversarial term through the frozen discriminator, then the discriminator step on the same
    reconstruction -- losses and the gradients of BOTH networks against the same composition of the CPU restatements."""
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    from medical_image_generation_amd.trainer import AEGANTrainer
    c = cases.AEKL_CASES["aekl_c3a"]
    ae_ref = nets.AutoencoderKL(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ae_ref.state_dict().items()}, S)
    ae_ref.load_state_dict(sd)
    ae = AutoencoderKL(**c["kwargs"])
    ae.load_state_dict(sd)
    ae = ae.cuda()
    d_ref, d = _pair()
    x = synth.ellipsoid_volume(S, "x", (2, 1, 32, 32, 32))
    with torch.no_grad():
        zshape = tuple(ae_ref.encode(x)[0].shape)
    eps = synth.tensor(S, "eps0", zshape)
    klw, advw = 1e-3, 0.5  # (weights large enough for both terms to show in the gradients)
    adv = odisc.PatchAdversarialLoss()
    d_ref.train()
    for p in d_ref.parameters():
        p.requires_grad_(False)
    loss_g, recon, gen = odisc.generator_loss(ae_ref, d_ref, adv, x, eps, klw, advw)
    loss_g.backward()
    for p in d_ref.parameters():
        p.requires_grad_(True)
    loss_d = odisc.discriminator_loss(d_ref, adv, x, recon, advw)
    loss_d.backward()
    tr = AEGANTrainer(ae, d, adv_weight=advw, kl_weight=klw, lr=cases.STEP_LR, d_lr=cases.STEP_LR, max_grad_norm=1.0)
    xd, ed = x.cuda(), eps.cuda()
    if graph:
        tr.capture(xd, ed)
        tr._g_fb.replay()
        tr._g_d.replay()   # (its optimizer step runs behind the gradients: read the losses, compare updates below)
    else:
        tr.forward_backward(xd, ed)
        tr.d_forward_backward(xd)
    print(f"\n[GAN step graph={graph}] generator loss {float(tr.loss):.5f} vs {float(loss_g):.5f} (adv term {float(tr.gen_loss):.5f} vs {float(gen):.5f}); "
          f"discriminator loss {float(tr.disc_loss):.5f} vs {float(loss_d):.5f}")
    assert abs(float(tr.loss) - float(loss_g)) <= 2e-2 * abs(float(loss_g))
    assert abs(float(tr.gen_loss) - float(gen)) <= 5e-2 * abs(float(gen))
    assert abs(float(tr.disc_loss) - float(loss_d)) <= 5e-2 * abs(float(loss_d))
    names = [n for n, p in ae_ref.named_parameters() if p.grad is not None]
    g_ref = torch.cat([dict(ae_ref.named_parameters())[n].grad.flatten() for n in names])
    g_hip = torch.cat([tr.arena.gview(n).cpu().flatten() for n in names])
    cos = float(torch.dot(g_ref, g_hip) / (g_ref.norm() * g_hip.norm()))
    print(f"  autoencoder gradient: cosine {cos:.4f}, norm ratio {float(g_hip.norm() / g_ref.norm()):.3f}")
    assert cos >= 0.98 and abs(float(g_hip.norm() / g_ref.norm()) - 1) <= 0.05
    if not graph:
        dn = [n for n, p in d_ref.named_parameters()]
        gd_ref = torch.cat([dict(d_ref.named_parameters())[n].grad.flatten() for n in dn])
        gd_hip = torch.cat([tr.d_arena.gview(n).cpu().flatten() for n in dn])
        e = float((gd_hip - gd_ref).norm() / gd_ref.norm())
        cosd = float(torch.dot(gd_hip, gd_ref) / (gd_hip.norm() * gd_ref.norm()))
        print(f"  discriminator gradient rel-L2 {e:.3e}, cosine {cosd:.4f}")
        assert e <= 1.2e-1 and cosd >= 0.99  # (bf16 budget of this net: see test_discriminator_forward_backward_matches_oracle)
    # a full step of both networks moves both parameter sets and stays finite
    before_g, before_d = tr.arena.data.clone(), tr.d_arena.data.clone()
    loss = tr.step_graph(xd, ed) if graph else tr.step(xd, ed)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss).all()) and bool(torch.isfinite(tr.arena.data).all()) and bool(torch.isfinite(tr.d_arena.data).all())
    assert float((tr.arena.data - before_g).abs().max()) > 0 and float((tr.d_arena.data - before_d).abs().max()) > 0


def test_discriminator_constructor_surface():
    from medical_image_generation_amd.discriminator import PatchAdversarialLoss, PatchDiscriminator
    with pytest.raises(NotImplementedError):
        PatchDiscriminator(spatial_dims=2, num_channels=8, in_channels=1)
    with pytest.raises(NotImplementedError):
        PatchDiscriminator(spatial_dims=3, num_channels=8, in_channels=1, norm="INSTANCE")
    with pytest.raises(NotImplementedError):
        PatchAdversarialLoss(criterion="hinge")
    net = PatchDiscriminator(**dict(DKW, num_channels=64))  # the planner's discriminator_params (configuration.py:966-967)
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in odisc.PatchDiscriminator(**dict(DKW, num_channels=64)).parameters())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 16, 16, 16))
