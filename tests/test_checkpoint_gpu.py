"""Activation checkpointing (north_star: "mixed precision and activation checkpointing preserve the repo's memory story";
BASELINE config 5; reference: torch.utils.checkpoint around the AutoencoderKL encoder / decoder, autoencoderkl_with_strides.py:
761-762, 815-816).  engine.checkpoint drops a segment's intermediates after the forward and recomputes them inside the backward.

Required of it: (1) peak memory of a train step drops; (2) gradients equal the stored-activation path's -- every bf16 tensor bit for
bit (prediction, loss gradient path), the fp32 gradient arena up to the run-to-run noise of the fp32-atomic bias / GroupNorm-parameter
reductions (measured in the same test by running the stored path twice)."""
import pytest
import torch

from oracle import cases

pytestmark = pytest.mark.gpu
dev = torch.device("cuda")


def _peak_and_grads(make_trainer, inputs):
    tr = make_trainer()
    tr.forward_backward(*inputs)  # first pass creates plans / workspaces
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    tr.forward_backward(*inputs)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    n = tr.arena.n_trainable
    return peak, float(tr.loss), tr.arena.grad[:n].clone(), tr


def test_unet_checkpointing_saves_memory_and_keeps_gradients():
    import bench
    from medical_image_generation_amd.trainer import DDPMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet

    def make(ckpt):
        def f():
            torch.manual_seed(21)
            net = DiffusionModelUNet(**cases.UNET_CASES["unet_c4"]["kwargs"])
            for p in net.parameters():
                if float(p.detach().abs().max()) == 0:
                    torch.nn.init.normal_(p, std=0.02)
            net.use_checkpointing = ckpt
            return DDPMTrainer(net.to(dev), lr=1e-4)
        return f

    d = 64
    x0 = bench.synthetic_volume((1, 1, d, d, d), 3, dev)
    g = torch.Generator(device=dev).manual_seed(2)
    inputs = (x0, torch.randn((1, 1, d, d, d), device=dev, generator=g), torch.tensor([321], device=dev))
    p_a, l_a, g_a, tr_a = _peak_and_grads(make(False), inputs)
    p_b, l_b, g_b, _ = _peak_and_grads(make(False), inputs)
    p_c, l_c, g_c, tr_c = _peak_and_grads(make(True), inputs)
    noise = float((g_a - g_b).norm() / g_a.norm())
    err = float((g_a - g_c).norm() / g_a.norm())
    print(f"\n[UNet C4 @64^3] peak activation memory {p_a / 2**20:.0f} MiB stored -> {p_c / 2**20:.0f} MiB checkpointed; "
          f"gradient rel-L2 vs stored {err:.2e} (stored vs stored: {noise:.2e}); loss {l_a} / {l_c}")
    assert p_c < 0.75 * p_a, "checkpointing must lower the step's peak memory"
    assert abs(l_a - l_c) <= 2e-6 * abs(l_a) and err <= max(4 * noise, 1e-6)
    with torch.no_grad():  # bf16 path: bit-identical predictions
        t = inputs[2]
        assert torch.equal(tr_a.model(x0, t), tr_c.model(x0, t))


def test_aekl_use_checkpointing_saves_memory_and_keeps_gradients():
    import bench
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    from medical_image_generation_amd.trainer import AETrainer

    def make(ckpt):
        def f():
            torch.manual_seed(22)
            kw = dict(cases.AEKL_CASES["aekl_c3a"]["kwargs"], use_checkpointing=ckpt)
            return AETrainer(AutoencoderKL(**kw).to(dev), lr=1e-4, kl_weight=1e-4)
        return f

    d = 64
    x = bench.synthetic_volume((2, 1, d, d, d), 4, dev)
    g = torch.Generator(device=dev).manual_seed(6)
    inputs = (x, torch.randn((2, 8, d // 4, d // 4, d // 4), device=dev, generator=g))
    p_a, l_a, g_a, _ = _peak_and_grads(make(False), inputs)
    p_b, l_b, g_b, _ = _peak_and_grads(make(False), inputs)
    p_c, l_c, g_c, _ = _peak_and_grads(make(True), inputs)
    noise = float((g_a - g_b).norm() / g_a.norm())
    err = float((g_a - g_c).norm() / g_a.norm())
    print(f"\n[AEKL C3a @2x64^3] peak activation memory {p_a / 2**20:.0f} MiB stored -> {p_c / 2**20:.0f} MiB with use_checkpointing; "
          f"gradient rel-L2 vs stored {err:.2e} (stored vs stored: {noise:.2e})")
    assert p_c < 0.75 * p_a
    assert abs(l_a - l_c) <= 2e-6 * abs(l_a) and err <= max(4 * noise, 1e-6)


def test_checkpointing_through_the_autograd_edge():
    """The drop-in modules (loss.backward() through _NetFn) honour it too, and a checkpointed step can be captured in a hipGraph."""
    from medical_image_generation_amd.trainer import DDPMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet
    from oracle import synth
    c = cases.UNET_CASES["unet3d"]
    sd = None
    outs = []
    for ckpt in (False, True):
        net = DiffusionModelUNet(**c["kwargs"])
        if sd is None:
            sd = synth.state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, cases.SEED)
        net.load_state_dict(sd)
        net = net.to(dev)
        net.use_checkpointing = ckpt
        x = synth.tensor(cases.SEED, "x", c["shape"]).to(dev).requires_grad_(True)
        y = net(x, torch.tensor(c["timesteps"], device=dev))
        y.square().mean().backward()
        outs.append((y.detach().clone(), x.grad.clone(), torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float((outs[0][2] - outs[1][2]).norm() / outs[0][2].norm()) <= 1e-6
    net.use_checkpointing = True
    tr = DDPMTrainer(net, lr=1e-4)
    x0, nz, t = (synth.ellipsoid_volume(cases.SEED, "x0", c["shape"]).to(dev), synth.tensor(cases.SEED, "n", c["shape"]).to(dev),
                 torch.tensor(c["timesteps"], device=dev))
    l_eager = float(tr.step(x0, nz, t))
    tr2 = DDPMTrainer(net, lr=1e-4)
    tr2.capture(x0, nz, t)
    assert torch.isfinite(tr2.step_graph()).all() and l_eager > 0
