"""End-to-end parity of the HIP AutoencoderKL against the golden vectors from the reference's model file and the CPU
oracle (same tolerances and rationale as tests/test_unet_gpu.py)."""
import pytest
import torch

from oracle import cases, nets, step, synth

pytestmark = pytest.mark.gpu
S = cases.SEED


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("name", list(cases.AEKL_CASES))
def test_aekl_matches_reference(golden, name):
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    g, meta = golden(name)
    c = cases.AEKL_CASES[name]
    ref = nets.AutoencoderKL(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd)
    net = AutoencoderKL(**c["kwargs"])
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    net.load_state_dict(sd)
    net = net.cuda()
    x = synth.ellipsoid_volume(S, "x", c["shape"])
    eps = synth.tensor(S, "eps", g["z_mu"].shape)
    xd = x.cuda()
    z_mu, z_sigma = net.encode(xd)
    recon = net.decode(z_mu + eps.cuda() * z_sigma)  # AutoencoderKL.forward with the sampling noise made explicit
    loss = torch.nn.functional.l1_loss(recon, xd) + step.kl_loss(z_mu, z_sigma) * cases.KL_WEIGHT
    loss.backward()
    e = {k: rel_l2(v.detach().cpu(), g[k]) for k, v in (("z_mu", z_mu), ("z_sigma", z_sigma), ("recon", recon))}
    lr, _, _, _ = step.ae_loss(ref, x, eps, cases.KL_WEIGHT)
    lr.backward()
    rg = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    hg = {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}
    assert sorted(hg) == sorted(rg) == meta["grad_names"].split("\n")
    flat_r = torch.cat([rg[n].flatten() for n in sorted(rg)])
    flat_h = torch.cat([hg[n].flatten() for n in sorted(rg)])
    e_glob = rel_l2(flat_h, flat_r)
    print(f"\n[{name}] rel-L2: {e}  loss {float(loss):.6f} vs {float(g['loss']):.6f}  grads(global) {e_glob:.3e}")
    assert all(v <= 3e-2 for v in e.values())
    assert abs(float(loss) - float(g["loss"])) <= 2e-2 * float(g["loss"])
    assert e_glob <= 6e-2  # L1 loss: sign(recon - x) flips on bf16-level differences add gradient noise
    assert net.encoder.spatial_dims == 3 and net.encoder.in_channels == 1 and net.latent_channels == c["kwargs"]["latent_channels"]


def test_aekl_convtranspose_matches_oracle():
    """use_convtranspose=True (AEKL:66-77; never set by the planner, CFG:843): the decoder's Upsample is a ConvTranspose3d with the
    level's (stride, kernel, padding) and output_padding = stride - 1.  On the HIP path it runs as the data gradient of the matching
    strided conv (phase kernels).  Against the CPU restatement (monai's wrapper is not under /root/reference: PARITY UNPINNED)."""
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    c = cases.AEKL_CASES["aekl_c3a"]
    kw = dict(c["kwargs"], use_convtranspose=True)
    ref = nets.AutoencoderKL(**kw)
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd)
    net = AutoencoderKL(**kw)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    net.load_state_dict(sd)
    net = net.cuda()
    x = synth.ellipsoid_volume(S, "x", c["shape"])
    xd = x.cuda()
    z_mu, z_sigma = net.encode(xd)
    eps = synth.tensor(S, "eps", tuple(z_mu.shape))
    recon = net.decode(z_mu + eps.cuda() * z_sigma)
    loss = torch.nn.functional.l1_loss(recon, xd) + step.kl_loss(z_mu, z_sigma) * cases.KL_WEIGHT
    loss.backward()
    lr, rrecon, rmu, rsig = step.ae_loss(ref, x, eps, cases.KL_WEIGHT)
    lr.backward()
    assert recon.shape == x.shape
    e_rec = rel_l2(recon.detach().cpu(), rrecon.detach())
    rg = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    hg = {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}
    assert sorted(hg) == sorted(rg)
    e_glob = rel_l2(torch.cat([hg[n].flatten() for n in sorted(rg)]), torch.cat([rg[n].flatten() for n in sorted(rg)]))
    ups = [n for n in rg if ".conv.conv.weight" in n and n.startswith("decoder")]
    e_up = max(rel_l2(hg[n], rg[n]) for n in ups)
    others = {n: rel_l2(hg[n], rg[n]) for n in rg if n.startswith("decoder") and n.endswith("conv.weight") and n not in ups}
    print("\n[aekl convtranspose] decoder conv weight gradients, rel-L2:", {n: round(rel_l2(hg[n], rg[n]), 4) for n in ups},
          "other decoder convs: max", round(max(others.values()), 4), "median", round(sorted(others.values())[len(others) // 2], 4))
    cos = min(float(torch.nn.functional.cosine_similarity(hg[n].flatten().double(), rg[n].flatten().double(), dim=0)) for n in ups)
    print("[aekl convtranspose] min cosine of the up weights' gradients:", round(cos, 5))
    print(f"\n[aekl convtranspose] recon rel-L2 {e_rec:.3e}  loss {float(loss):.6f} vs {float(lr):.6f}  grads(global) {e_glob:.3e}  up weights {e_up:.3e}")
    assert e_rec <= 3e-2 and abs(float(loss) - float(lr)) <= 2e-2 * float(lr) and e_glob <= 6e-2
    # per-layer weight gradients of this L1-loss net carry sign-flip noise (ordinary decoder convs: median 5e-2, max 1.1e-1 measured);
    # the transposed convs must sit inside that band and point the same way
    assert e_up <= 1.25 * max(others.values()) and cos >= 0.99


def test_aekl_api_surface():
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    c = cases.AEKL_CASES["aekl_c3a"]
    net = AutoencoderKL(**c["kwargs"]).cuda()
    x = synth.ellipsoid_volume(S, "x", (1, 1, 16, 16, 16)).cuda()
    with torch.no_grad():
        recon, z_mu, z_sigma = net(x)
        assert recon.shape == x.shape and z_mu.shape == (1, 8, 4, 4, 4) == z_sigma.shape
        assert net.reconstruct(x).shape == x.shape
        z = net.encode_stage_2_inputs(x)
        assert net.decode_stage_2_outputs(z).shape == x.shape
        assert float(z_sigma.min()) > 0
    with pytest.raises(ValueError):
        AutoencoderKL(3, num_channels=(30, 64), attention_levels=(False, False))
    with pytest.raises(RuntimeError):
        net.encode(x.cpu())
