"""Fused HIP train step (q-sample -> UNet -> MSE -> backward -> clip -> Adam[W]) against the oracle's train step and
against the golden optimizer vectors produced with the reference's model file (oracle/tools/gen_golden.py)."""
import pytest
import torch

from oracle import cases, nets, step, synth

pytestmark = pytest.mark.gpu
S = cases.SEED


def _nets(name):
    from medical_image_generation_amd.unet import DiffusionModelUNet
    c = cases.UNET_CASES[name]
    ref = nets.DiffusionModelUNet(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd)
    net = DiffusionModelUNet(**c["kwargs"])
    net.load_state_dict(sd)
    return c, ref, net.cuda()


@pytest.mark.parametrize("name", list(cases.STEP_CASES))
@pytest.mark.parametrize("graph", [False, True])
def test_three_train_steps(golden, name, graph):
    from medical_image_generation_amd.trainer import DDPMTrainer
    g, meta = golden(name + "_steps")
    c, ref, net = _nets(name)
    opt_name = meta["optimizer"]
    tr = DDPMTrainer(net, lr=cases.STEP_LR, optimizer=opt_name, max_grad_norm=1.0)
    x0 = synth.ellipsoid_volume(S, "x0", c["shape"])
    t = torch.tensor(c["timesteps"])
    x0d = x0.cuda()
    losses = []
    for k in range(cases.STEP_COUNT):
        noise = synth.tensor(S, f"noise{k}", c["shape"]).cuda()
        tk = ((t + 37 * k) % 1000).cuda()
        if graph:
            if k == 0:
                tr.capture(x0d, noise, tk)
            loss = tr.step_graph(x0d, noise, tk)
        else:
            loss = tr.step(x0d, noise, tk)
        losses.append(float(loss))
    ref_losses = g["losses"].tolist()
    print(f"\n[{name} graph={graph}] losses hip {losses} ref {ref_losses}")
    # loss: mean of squares over >= 8k elements, bf16 forward -> 1% relative
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-2 * abs(b)
    # parameters after 3 clipped Adam steps: every element moved by <= 3*lr; compare the UPDATE direction globally
    sd0 = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    opt = getattr(torch.optim, opt_name)(ref.parameters(), lr=cases.STEP_LR)
    sched = step.DDPMSchedule()
    for k in range(cases.STEP_COUNT):
        step.ddpm_train_step(ref, opt, sched, x0, synth.tensor(S, f"noise{k}", c["shape"]), (t + 37 * k) % 1000, max_norm=1.0)
    names = [n for n, p in ref.named_parameters() if ".proj_attn." not in n]
    upd_ref = torch.cat([(ref.state_dict()[n] - sd0[n]).flatten() for n in names])
    upd_hip = torch.cat([(net.state_dict()[n].cpu() - sd0[n]).flatten() for n in names])
    cos = float(torch.dot(upd_ref, upd_hip) / (upd_ref.norm() * upd_hip.norm()))
    print(f"  update cosine {cos:.4f}  |upd| ref {float(upd_ref.norm()):.4f} hip {float(upd_hip.norm()):.4f}")
    # Adam turns every gradient into a +-lr step, so sign flips of noise-level gradients (bf16) cost cosine; 0.9 still
    # means > 95% of the elements moved the same way by the same amount
    assert cos >= 0.9 and abs(float(upd_hip.norm()) / float(upd_ref.norm()) - 1) <= 0.05
    # statically unused tensors are untouched, like torch.optim skipping grad-None parameters
    for n in sd0:
        if ".proj_attn." in n:
            assert torch.equal(net.state_dict()[n].cpu(), sd0[n])


@pytest.mark.parametrize("name", list(cases.AEKL_CASES))
@pytest.mark.parametrize("graph", [False, True])
def test_ae_three_train_steps(name, graph):
    """AETrainer (encode -> sample -> decode -> L1 + kl_weight*KL -> backward -> Adam) against the oracle's ae_loss driven
    by torch.optim.Adam on the CPU restatement; the forward/gradient parity of the same nets against the reference's
    golden vectors is in tests/test_aekl_gpu.py."""
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    from medical_image_generation_amd.trainer import AETrainer
    c = cases.AEKL_CASES[name]
    ref = nets.AutoencoderKL(**c["kwargs"])
    sd0 = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd0)
    net = AutoencoderKL(**c["kwargs"])
    net.load_state_dict(sd0)
    net = net.cuda()
    x = synth.ellipsoid_volume(S, "x", c["shape"])
    with torch.no_grad():
        zshape = tuple(ref.encode(x)[0].shape)
    klw = 1e-3  # large enough for the KL term to show in the loss and in the quant_conv gradients
    tr = AETrainer(net, lr=cases.STEP_LR, kl_weight=klw, max_grad_norm=1.0)
    opt = torch.optim.Adam(ref.parameters(), lr=cases.STEP_LR)
    xd = x.cuda()
    losses, ref_losses = [], []
    for k in range(cases.STEP_COUNT):
        eps = synth.tensor(S, f"eps{k}", zshape)
        opt.zero_grad(set_to_none=True)
        lr_, _, _, _ = step.ae_loss(ref, x, eps, klw)
        lr_.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
        ref_losses.append(float(lr_.detach()))
        if graph:
            if k == 0:
                tr.capture(xd, eps.cuda())
            loss = tr.step_graph(xd, eps.cuda())
        else:
            loss = tr.step(xd, eps.cuda())
        losses.append(float(loss))
    print(f"\n[{name} graph={graph}] losses hip {losses} ref {ref_losses}")
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-2 * abs(b)
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    upd_ref = torch.cat([(ref.state_dict()[n] - sd0[n]).flatten() for n in names])
    upd_hip = torch.cat([(net.state_dict()[n].cpu() - sd0[n]).flatten() for n in names])
    cos = float(torch.dot(upd_ref, upd_hip) / (upd_ref.norm() * upd_hip.norm()))
    print(f"  update cosine {cos:.4f}  |upd| ref {float(upd_ref.norm()):.4f} hip {float(upd_hip.norm()):.4f}")
    # L1's sign() gradient flips on bf16-level differences of recon - x, and Adam turns every flip into a full +-lr step
    assert cos >= 0.85 and abs(float(upd_hip.norm()) / float(upd_ref.norm()) - 1) <= 0.05


def test_ae_extra_loss_hook_matches_oracle():
    """The generator step with the reference's third-party terms (T-AE:411-421) supplied through AETrainer(extra_loss=...): a small
    torch patch discriminator with the least-squares generator loss (PatchAdversarialLoss(criterion="least_squares"): mean((D(x) - 1)^2))
    and a feature-space "perceptual" term.  The hook's gradient must reach the AE parameters exactly as autograd delivers it in the
    oracle composition: compared through the DIFFERENCE it makes to the parameter gradient (with the hook minus without)."""
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    from medical_image_generation_amd.trainer import AETrainer
    c = cases.AEKL_CASES["aekl_c3a"]
    ref = nets.AutoencoderKL(**c["kwargs"])
    sd0 = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd0)
    net = AutoencoderKL(**c["kwargs"])
    net.load_state_dict(sd0)
    net = net.cuda()
    torch.manual_seed(4)
    disc = torch.nn.Sequential(torch.nn.Conv3d(1, 8, 4, 2, 1), torch.nn.LeakyReLU(0.2), torch.nn.Conv3d(8, 1, 4, 1, 1))
    feat = torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, 1, 1), torch.nn.ReLU(), torch.nn.AvgPool3d(2))

    def extra(d, f):
        def fn(rec, images):
            return 2.0 * ((d(rec) - 1.0) ** 2).mean() + 20.0 * ((f(rec) - f(images)) ** 2).mean()  # (weights that put the hook gradient above the bf16 quantum of the L1 sign gradient)
        return fn

    x = synth.ellipsoid_volume(S, "x", c["shape"])
    with torch.no_grad():
        zshape = tuple(ref.encode(x)[0].shape)
    eps = synth.tensor(S, "eps0", zshape)
    names = [n for n, p in ref.named_parameters()]

    def ref_grads(with_hook):
        ref.zero_grad(set_to_none=True)
        loss, recon, _, _ = step.ae_loss(ref, x, eps, 1e-3)
        if with_hook:
            loss = loss + extra(disc, feat)(recon.float(), x)
        loss.backward()
        return float(loss.detach()), torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for _, p in ref.named_parameters()])

    import copy
    dg, fg = copy.deepcopy(disc).cuda(), copy.deepcopy(feat).cuda()
    for m in (dg, fg):
        for p in m.parameters():
            p.requires_grad_(False)  # T-AE:399-400: the discriminator is frozen in the generator step

    def hip_grads(with_hook):
        tr = AETrainer(net, lr=cases.STEP_LR, kl_weight=1e-3, max_grad_norm=1.0, extra_loss=extra(dg, fg) if with_hook else None)
        tr.forward_backward(x.cuda(), eps.cuda())
        g = torch.cat([tr.arena.gview(n).cpu().flatten() for n in names])
        if with_hook:
            assert tr.reconstruction is not None and tuple(tr.reconstruction.shape) == tuple(x.shape)
        return float(tr.loss), g

    l0r, g0r = ref_grads(False)
    l1r, g1r = ref_grads(True)
    l0h, g0h = hip_grads(False)
    l1h, g1h = hip_grads(True)
    dr, dh = g1r - g0r, g1h - g0h
    cos = float(torch.dot(dr, dh) / (dr.norm() * dh.norm()))
    ratio = float(dh.norm() / dr.norm())
    print(f"\n[AE extra_loss hook] loss {l1h:.5f} vs {l1r:.5f} (without: {l0h:.5f} vs {l0r:.5f}); hook gradient cosine {cos:.4f}, norm ratio {ratio:.3f}")
    assert abs(l1h - l1r) <= 2e-2 * abs(l1r) and abs((l1h - l0h) - (l1r - l0r)) <= 5e-2 * abs(l1r - l0r)
    assert float(dr.norm()) > 1e-3 * float(g0r.norm())  # the hook matters in this set-up
    # the difference of two bf16 backward passes, each carrying the sign-gradient noise of L1: cosine, not element-wise
    assert cos >= 0.9 and abs(ratio - 1) <= 0.1


@pytest.mark.parametrize("graph", [False, True])
def test_conditioned_step_matches_oracle(graph):
    """DDPMTrainer(..., context=...): the fused step of a with_conditioning=True net (cross-attention on a context, UNet:72-342,
    1936-1944) against the oracle's q-sample -> UNet(context) -> MSE -> backward; the context is a constant of the step."""
    from medical_image_generation_amd.trainer import DDPMTrainer
    c, ref, net = _nets("unet2d_xattn")
    tr = DDPMTrainer(net, lr=cases.STEP_LR, optimizer="AdamW", max_grad_norm=1.0)
    sched = step.DDPMSchedule()
    x0 = synth.ellipsoid_volume(S, "x0", c["shape"])
    noise = synth.tensor(S, "noise0", c["shape"])
    t = torch.tensor(c["timesteps"])
    context = synth.tensor(S, "context", c["context"])
    noisy = sched.add_noise(x0, noise, t)
    pred = ref(noisy, t, context=context)
    loss_ref = torch.nn.functional.mse_loss(pred.float(), noise.float())
    loss_ref.backward()
    args = (x0.cuda(), noise.cuda(), t.cuda(), None, context.cuda())
    if graph:
        tr.capture(*args)
        tr._g_fb.replay()
    else:
        tr.forward_backward(*args)
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    g_ref = torch.cat([dict(ref.named_parameters())[n].grad.flatten() for n in names])
    g_hip = torch.cat([tr.arena.gview(n).cpu().flatten() for n in names])
    e = float((g_hip - g_ref).norm() / g_ref.norm())
    print(f"\n[conditioned step graph={graph}] loss {float(tr.loss):.6f} vs {float(loss_ref):.6f}; gradient rel-L2 {e:.3e}")
    assert abs(float(tr.loss) - float(loss_ref)) <= 1e-2 * float(loss_ref) and e <= 4e-2
    with pytest.raises(ValueError):
        tr.forward_backward(x0.cuda(), noise.cuda(), t.cuda())  # a conditioned net needs its context


@pytest.mark.parametrize("graph", [False, True])
def test_concat_conditioned_step_matches_oracle(graph):
    """BASELINE configs[4]: label-channel (channel-concatenation) conditioning on the EXACT C5 net kwargs (in 9 / out 8, one head of
    512 / 768) at 12^3 with BATCH 2: the 8 latent channels are noised, the label channel is written un-noised behind them, the net
    predicts 8 channels and the MSE target is the latents' noise (the inferer's `condition=, mode="concat"` of train_ddpm.py:191,
    restated in oracle/step.py).  Batch 2 is the case a 9-channel target read with C = 8 got wrong."""
    from medical_image_generation_amd.trainer import DDPMTrainer
    c, ref, net = _nets("unet_c5")
    tr = DDPMTrainer(net, lr=cases.STEP_LR, optimizer="AdamW", max_grad_norm=1.0)
    sched = step.DDPMSchedule()
    shape = (2, 8, 12, 12, 12)
    x0 = synth.tensor(S, "latents", shape)
    noise = synth.tensor(S, "noise0", shape)
    label = (synth.ellipsoid_volume(S, "label", (2, 1, 12, 12, 12)) > 0).float()  # a binary mask, like a segmentation channel
    t = torch.tensor([42, 873])
    loss_ref, _ = step.ddpm_loss(ref, sched, x0, noise, t, condition=label)
    loss_ref.backward()
    args = (x0.cuda(), noise.cuda(), t.cuda(), None, None, label.cuda())
    if graph:
        tr.capture(*args)
        tr._g_fb.replay()
    else:
        tr.forward_backward(*args)
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    g_ref = torch.cat([dict(ref.named_parameters())[n].grad.flatten() for n in names])
    g_hip = torch.cat([tr.arena.gview(n).cpu().flatten() for n in names])
    e = float((g_hip - g_ref).norm() / g_ref.norm())
    print(f"\n[C5 concat-conditioned step b2 graph={graph}] loss {float(tr.loss):.6f} vs {float(loss_ref):.6f}; gradient rel-L2 {e:.3e}")
    assert abs(float(tr.loss) - float(loss_ref)) <= 1e-2 * float(loss_ref) and e <= 4e-2
    # channel bookkeeping is checked, not assumed: noising all 9 input channels cannot produce an 8-channel target ...
    x9 = torch.cat([x0, label], dim=1).cuda().contiguous()
    with pytest.raises(ValueError, match="predicts 8 channels"):
        tr.forward_backward(x9, torch.randn_like(x9), t.cuda())
    # ... and the input channel count must add up
    with pytest.raises(ValueError, match="expected number of channels"):
        tr.forward_backward(x0.cuda(), noise.cuda(), t.cuda())
    with pytest.raises(ValueError):
        tr.forward_backward(x0.cuda(), noise.cuda(), t.cuda(), None, None, label.cuda()[:, :, :6])  # spatial mismatch
    # the autograd-edge form: DiffusionInferer.__call__(condition=, mode="concat") is the same forward
    from medical_image_generation_amd.inferer import DDPMScheduler, DiffusionInferer
    inferer = DiffusionInferer(DDPMScheduler(1000, "scaled_linear_beta", beta_start=0.0015, beta_end=0.0205))
    with torch.no_grad():
        pred = inferer(inputs=x0.cuda(), diffusion_model=net, noise=noise.cuda(), timesteps=t.cuda(), condition=label.cuda(), mode="concat")
    l2 = float(torch.nn.functional.mse_loss(pred.float(), noise.cuda()))
    assert pred.shape == shape and abs(l2 - float(loss_ref)) <= 1e-2 * float(loss_ref)
    with pytest.raises(NotImplementedError):
        inferer(inputs=x0.cuda(), diffusion_model=net, noise=noise.cuda(), timesteps=t.cuda(), condition=label.cuda(), mode="film")


def test_v_prediction_step_matches_oracle():
    """prediction_type = "v_prediction" (train_ldm.py:163-165): the target is scheduler.get_velocity(x0, noise, t)."""
    from medical_image_generation_amd.trainer import DDPMSchedule, DDPMTrainer
    c, ref, net = _nets("unet3d")
    tr = DDPMTrainer(net, lr=cases.STEP_LR, optimizer="AdamW", max_grad_norm=1.0, schedule=DDPMSchedule(prediction_type="v_prediction"))
    sched = step.DDPMSchedule(prediction_type="v_prediction")
    x0 = synth.ellipsoid_volume(S, "x0", c["shape"])
    noise = synth.tensor(S, "noise0", c["shape"])
    t = torch.tensor(c["timesteps"])
    loss_ref, _ = step.ddpm_loss(ref, sched, x0, noise, t)
    loss_ref.backward()
    tr.forward_backward(x0.cuda(), noise.cuda(), t.cuda())
    assert abs(float(tr.loss) - float(loss_ref)) <= 1e-2 * abs(float(loss_ref))
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    g_ref = torch.cat([dict(ref.named_parameters())[n].grad.flatten() for n in names])
    g_hip = torch.cat([tr.arena.gview(n).cpu().flatten() for n in names])
    err = float((g_hip - g_ref).norm() / g_ref.norm())
    print(f"\n[v-prediction] loss {float(tr.loss):.6f} vs {float(loss_ref):.6f}, global gradient rel-L2 {err:.3e}")
    assert err <= 4e-2
    with pytest.raises(ValueError):
        DDPMSchedule(prediction_type="sample")


def test_gradient_accumulation_matches_oracle():
    """grad_accumulate_step = 2 (train_ldm.py:173-180): the gradients of two micro-batches are SUMMED (loss not divided), clipped and
    applied once; the optimizer does not move on the first micro-step."""
    from medical_image_generation_amd.trainer import DDPMTrainer
    c, ref, net = _nets("unet3d")
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    tr = DDPMTrainer(net, lr=cases.STEP_LR, optimizer="AdamW", max_grad_norm=1.0, grad_accumulate_step=2)
    opt = torch.optim.AdamW(ref.parameters(), lr=cases.STEP_LR)
    sched = step.DDPMSchedule()
    t = torch.tensor(c["timesteps"])
    x = [synth.ellipsoid_volume(S, f"x0_{k}", c["shape"]) for k in range(2)]
    nz = [synth.tensor(S, f"noise{k}", c["shape"]) for k in range(2)]
    losses_ref = []
    for k in range(2):  # reference loop body: backward every micro-step, clip + step + zero_grad on the boundary
        loss, _ = step.ddpm_loss(ref, sched, x[k], nz[k], (t + 11 * k) % 1000)
        loss.backward()
        losses_ref.append(float(loss))
    g_ref = torch.cat([p.grad.flatten() for n, p in ref.named_parameters() if p.grad is not None])
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    opt.step()
    before = tr.arena.data.clone()
    l0 = float(tr.step(x[0].cuda(), nz[0].cuda(), t.cuda()))
    assert torch.equal(tr.arena.data, before) and float(tr.step_count) == 0.0  # no optimizer step on a non-boundary micro-step
    l1 = float(tr.step(x[1].cuda(), nz[1].cuda(), ((t + 11) % 1000).cuda()))
    assert float(tr.step_count) == 1.0 and tr._micro == 0
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    g_hip = torch.cat([tr.arena.gview(n).cpu().flatten() for n in names])  # arena.grad holds the folded sum after the boundary
    e_g = float((g_hip - g_ref).norm() / g_ref.norm())
    upd_ref = torch.cat([(ref.state_dict()[n] - sd0[n]).flatten() for n in names])
    upd_hip = torch.cat([(net.state_dict()[n].cpu() - sd0[n]).flatten() for n in names])
    cos = float(torch.dot(upd_ref, upd_hip) / (upd_ref.norm() * upd_hip.norm()))
    print(f"\n[accumulate 2] losses {l0:.5f} {l1:.5f} vs {losses_ref}; summed-gradient rel-L2 {e_g:.3e}; update cosine {cos:.4f}")
    assert abs(l0 - losses_ref[0]) <= 1e-2 * losses_ref[0] and abs(l1 - losses_ref[1]) <= 1e-2 * losses_ref[1]
    assert e_g <= 4e-2 and cos >= 0.9
    # last_in_epoch forces the boundary (T-LDM:173: `(step + 1) == len(train_loader)`)
    tr.step(x[0].cuda(), nz[0].cuda(), t.cuda(), last_in_epoch=True)
    assert float(tr.step_count) == 2.0 and tr._micro == 0


def test_backward_cut_is_sound_and_split_graphs_agree():
    """Data-parallel overlap rests on one claim: when the tape reaches the model's cut mark, every gradient of the arena's early
    segment [n_late, n_trainable) is FINAL.  Check it on the real net: snapshot the early segment at the cut, finish the backward,
    compare bit for bit.  Then the two-graph capture (cut between the graphs, as world > 1 runs it) must reproduce the one-graph
    step: same loss, same parameters after two steps (up to the fp32-atomic summation-order noise of the bias reductions)."""
    import bench
    from medical_image_generation_amd.trainer import DDPMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet

    def make():
        torch.manual_seed(5)
        net = DiffusionModelUNet(**bench.C4)
        for p in net.parameters():
            if float(p.detach().abs().max()) == 0:
                torch.nn.init.normal_(p, std=0.02)
        return net.cuda()

    d = 32
    x0 = bench.synthetic_volume((2, 1, d, d, d), 3, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(1)
    noise = torch.randn((2, 1, d, d, d), device="cuda", generator=g)
    t = torch.tensor([100, 900], device="cuda")
    tr = DDPMTrainer(make(), lr=1e-4)
    a = tr.arena
    assert 0 < a.n_late < 0.1 * a.n_trainable  # the late prefix is a few percent of the bytes
    snap = {}
    tr.forward_backward(x0, noise, t, on_cut=lambda: snap.setdefault("early", a.grad[a.n_late:a.n_trainable].clone()))
    early_final = a.grad[a.n_late:a.n_trainable]
    assert float(snap["early"].abs().max()) > 0
    assert torch.equal(snap["early"], early_final), "a gradient of the early segment changed after the cut mark"
    late = a.grad[:a.n_late]
    assert float(late.abs().max()) > 0  # ... and the late prefix does receive gradients afterwards
    # one-graph vs two-graph capture
    out = []
    for split in (False, True):
        trn = DDPMTrainer(make(), lr=1e-4)
        trn._force_split = split
        trn.capture(x0, noise, t)
        assert (trn._g_fb2 is not None) == split
        l0 = float(trn.step_graph())
        g0 = trn.arena.grad[:trn.arena.n_trainable].clone()  # gradients of the first replay: before Adam has amplified anything
        out.append(([l0, float(trn.step_graph())], g0, trn.arena.data[:trn.arena.n_trainable].clone()))
    (l_a, g_a, p_a), (l_b, g_b, p_b) = out
    eg, e = float((g_a - g_b).norm() / g_a.norm()), float((p_a - p_b).norm() / p_a.norm())
    print(f"\n[cut] n_late {a.n_late} of {a.n_trainable}; one-graph vs two-graph: losses {l_a} / {l_b}, gradient rel-L2 {eg:.2e}, "
          f"parameters after 2 steps {e:.2e}")
    # gradients: fp32-atomic summation-order noise only.  Parameters / the second loss: Adam turns that noise into +-lr steps on the
    # tensors whose exact gradient is zero (to_k.bias, ...), so they agree to a few 1e-4, not to 1e-8
    assert abs(l_a[0] - l_b[0]) <= 2e-6 * abs(l_a[0]) and eg <= 1e-6
    assert abs(l_a[1] - l_b[1]) <= 1e-3 * abs(l_a[1]) and e <= 1e-3


_CUT_VARIANTS = {
    # the variants that change WHICH parameters complete late in the backward: resnet resamplers, cross-attention blocks,
    # the class-embedding table (written by the time-embedding backward, the very last tape entry), per-block recomputation
    "updown": dict(case="unet2d_updown", over=dict(num_channels=(16, 32, 32, 64), attention_levels=(False, False, False, True),
                                                   num_head_channels=(0, 0, 0, 16), num_res_blocks=1,
                                                   strides=[[1, 1]] + [[2, 2]] * 3, kernel_sizes=[[3, 3]] + [[2, 2]] * 3,
                                                   paddings=[[1, 1]] + [[0, 0]] * 3), shape=(2, 1, 32, 32)),
    "xattn": dict(case="unet2d_xattn", over=dict(num_channels=(32, 32, 64, 64), attention_levels=(False, False, True, True),
                                                 num_head_channels=(0, 0, 32, 32), strides=[[1, 1]] + [[2, 2]] * 3,
                                                 kernel_sizes=[[3, 3]] * 4, paddings=[[1, 1]] * 4), shape=(2, 1, 32, 32), context=(2, 5, 16)),
    "class": dict(case="unet2d_class", over=dict(num_channels=(16, 32, 32, 32), attention_levels=(False, False, False, True),
                                                 num_head_channels=(0, 0, 0, 16), strides=[[1, 1]] + [[2, 2]] * 3,
                                                 kernel_sizes=[[3, 3]] * 4, paddings=[[1, 1]] * 4), shape=(2, 1, 32, 32), labels=(1, 3)),
    "ckpt": dict(case="unet3d", over=dict(num_channels=(32, 32, 64, 64), attention_levels=(False, False, False, True),
                                          num_head_channels=(0, 0, 0, 32), strides=[[1] * 3] + [[2] * 3] * 3, kernel_sizes=[[3] * 3] * 4,
                                          paddings=[[1] * 3] * 4), shape=(2, 1, 16, 16, 16), ckpt=True),
}


@pytest.mark.parametrize("variant", list(_CUT_VARIANTS))
def test_backward_cut_is_sound_for_every_block_family(variant):
    """The claim of test_backward_cut_is_sound_and_split_graphs_agree -- at the cut mark every gradient of [n_late, n_trainable) is
    final -- on 4-level nets with resblock_updown, with_conditioning, num_class_embeds and use_checkpointing (each changes which
    parameters the tail of the backward still writes), and the two-graph capture against the one-graph capture on the same nets."""
    from medical_image_generation_amd.trainer import DDPMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet
    v = _CUT_VARIANTS[variant]
    kw = dict(cases.UNET_CASES[v["case"]]["kwargs"], **v["over"])

    def make():
        torch.manual_seed(5)
        net = DiffusionModelUNet(**kw)
        for p in net.parameters():
            if float(p.detach().abs().max()) == 0:
                torch.nn.init.normal_(p, std=0.02)
        net.use_checkpointing = bool(v.get("ckpt"))
        return net.cuda()

    shape = v["shape"]
    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.rand(shape, device="cuda", generator=g)
    noise = torch.randn(shape, device="cuda", generator=g)
    t = torch.tensor([100, 900], device="cuda")
    labels = torch.tensor(v["labels"], device="cuda") if "labels" in v else None
    context = torch.randn(v["context"], device="cuda", generator=g) if "context" in v else None
    args = (x0, noise, t, labels, context)
    tr = DDPMTrainer(make(), lr=1e-4)
    a = tr.arena
    assert 0 < a.n_late < a.n_trainable
    snap = {}
    tr.forward_backward(*args, on_cut=lambda: snap.setdefault("early", a.grad[a.n_late:a.n_trainable].clone()))
    assert "early" in snap and float(snap["early"].abs().max()) > 0
    assert torch.equal(snap["early"], a.grad[a.n_late:a.n_trainable]), f"{variant}: an early-segment gradient changed after the cut mark"
    assert float(a.grad[:a.n_late].abs().max()) > 0
    # every trainable tensor is in exactly one segment and received a gradient
    for name, _, trainable in tr.model._entries:
        if trainable:
            assert float(a.gview(name).abs().max()) > 0 or name.endswith(("to_k.bias",)), f"{variant}: no gradient reached {name}"
    out = []
    for split in (False, True):
        trn = DDPMTrainer(make(), lr=1e-4)
        trn._force_split = split
        trn.capture(*args)
        assert (trn._g_fb2 is not None) == split
        l0 = float(trn.step_graph())
        out.append((l0, trn.arena.grad[:trn.arena.n_trainable].clone()))
    (l_a, g_a), (l_b, g_b) = out
    eg = float((g_a - g_b).norm() / g_a.norm())
    print(f"\n[cut {variant}] n_late {a.n_late} of {a.n_trainable}; one-graph vs two-graph: loss {l_a} / {l_b}, gradient rel-L2 {eg:.2e}")
    assert abs(l_a - l_b) <= 1e-5 * abs(l_a) and eg <= 1e-5
    # overlap=False: the plain schedule (one backward graph, exchange afterwards) is what a capture gives even when a split is forced
    trp = DDPMTrainer(make(), lr=1e-4, overlap=False)
    trp._force_split = True
    trp.capture(*args)
    assert trp._g_fb2 is None


def test_rccl_collectives_between_graph_replays_one_rank():
    """The data-parallel step as world > 1 runs it, on the one GPU a test box has: a ONE-RANK process group on the real backend
    ("nccl" = RCCL), forward + backward captured as TWO hipGraphs, the early segment's all-reduce issued asynchronously between the
    replays, the late segment's after the second, then the optimizer graph.  A sum over one rank changes nothing, so gradients and
    parameters must equal the run without any exchange: this proves that RCCL calls between replays neither break the capture
    pools nor the ordering (what it cannot show is bandwidth or CU sharing with the persistent conv kernels: DESIGN.md section 6)."""
    import os
    import socket
    import bench
    import torch.distributed as dist
    from medical_image_generation_amd.trainer import DDPMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        def make():
            torch.manual_seed(5)
            net = DiffusionModelUNet(**bench.C4)
            for p in net.parameters():
                if float(p.detach().abs().max()) == 0:
                    torch.nn.init.normal_(p, std=0.02)
            return net.cuda()

        d = 32
        x0 = bench.synthetic_volume((2, 1, d, d, d), 3, torch.device("cuda"))
        g = torch.Generator(device="cuda").manual_seed(1)
        noise = torch.randn((2, 1, d, d, d), device="cuda", generator=g)
        t = torch.tensor([100, 900], device="cuda")
        out = []
        for exchange in (False, True):
            tr = DDPMTrainer(make(), lr=1e-4, bucket_mb=16)  # 16 MiB slices: the 165 MB early segment goes out as 10 collectives
            tr._force_split = True
            tr._force_exchange = exchange
            tr.capture(x0, noise, t)
            assert tr._g_fb2 is not None
            l0 = float(tr.step_graph())
            g0 = tr.arena.grad[:tr.arena.n_trainable].clone()
            l1 = float(tr.step_graph())
            out.append((l0, l1, g0, tr.arena.data[:tr.arena.n_trainable].clone()))
            # eager form of the same schedule (start_early from the tape's cut mark)
            tr.step(x0, noise, t)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(tr.loss).all())
        (la0, la1, ga, pa), (lb0, lb1, gb, pb) = out
        eg, ep = float((ga - gb).norm() / ga.norm()), float((pa - pb).norm() / pa.norm())
        print(f"\n[RCCL one-rank exchange between replays] losses {la0:.6f}/{la1:.6f} vs {lb0:.6f}/{lb1:.6f}; gradient rel-L2 {eg:.2e}, "
              f"parameters {ep:.2e}")
        # gradients: fp32-atomic summation-order noise only; parameters / second loss: Adam turns that noise into +-lr steps on tensors whose
        # exact gradient is zero, so they agree to ~1e-4 (the same floor as the one-graph vs two-graph comparison above)
        assert abs(la0 - lb0) <= 2e-6 * abs(la0) and eg <= 1e-6 and abs(la1 - lb1) <= 1e-3 * abs(la1) and ep <= 1e-3
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_child_process():
    """`python bench.py --gpus 2` as the driver's N > 1 run starts it, rehearsed on one GPU: bench.py launches its own ranks through
    torch.distributed.run as a CHILD process (never re-exec'ing a process that has touched the GPU), both ranks share this card with
    the gloo backend, capture the split graphs, exchange gradients and rank 0 prints the JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "32", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] > 0 and rec["scaling"] == "weak" and rec["loss"] == rec["loss"] and abs(rec["loss"]) < 1e3
    assert rec["config"]["hipgraph"] is True


def test_graph_survives_other_shape_forward():
    """a captured train step holds raw pointers into the PackBatch / conv plans / GroupNorm workspace; a forward at
    another shape between train steps (validation, sampling) used to replace and free them.  capture -> sample at another batch
    size -> replay must equal the eager trainer that never saw the detour."""
    from medical_image_generation_amd.inferer import DDPMScheduler, DiffusionInferer
    from medical_image_generation_amd.trainer import DDPMTrainer
    c, _, net_a = _nets("unet3d")
    _, _, net_b = _nets("unet3d")
    x0 = synth.ellipsoid_volume(S, "x0", c["shape"]).cuda()
    t = torch.tensor(c["timesteps"]).cuda()
    tra, trb = DDPMTrainer(net_a, lr=cases.STEP_LR), DDPMTrainer(net_b, lr=cases.STEP_LR)
    noise = [synth.tensor(S, f"noise{k}", c["shape"]).cuda() for k in range(3)]
    tra.capture(x0, noise[0], t)
    la = [float(tra.step_graph(x0, noise[0], t))]
    lb = [float(trb.step(x0, noise[0], t))]
    sch = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205)
    sch.set_timesteps(4)
    inf = DiffusionInferer(sch)
    big = (5,) + tuple(c["shape"][1:3]) + (24, 24)  # another batch size AND another spatial extent: new plans, a larger workspace
    inf.sample(torch.randn(big, device="cuda"), net_a, sch, verbose=False)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    for k in (1, 2):
        la.append(float(tra.step_graph(x0, noise[k], t)))
        lb.append(float(trb.step(x0, noise[k], t)))
    pa, pb = tra.arena.data[:tra.arena.n_trainable], trb.arena.data[:trb.arena.n_trainable]
    e = float((pa - pb).norm() / pb.norm())
    print(f"\n[graph lifetime] losses graph {la} eager {lb}; parameter rel-L2 {e:.2e}")
    # (the broken replay froze the parameters: losses off by 25 %, parameters by 2e-2; healthy runs differ by Adam's amplification
    # of fp32-atomic noise on zero-gradient tensors: ~1e-4 in the loss, ~1e-3 in the parameters after three steps at lr 1e-3)
    assert all(abs(u - v) <= 2e-3 * abs(v) for u, v in zip(la, lb)) and e <= 5e-3
    with pytest.raises(ValueError):
        tra.step(x0.double(), noise[0], t)           # wrong dtype is refused, not misread
    with pytest.raises(ValueError):
        tra.step(x0, noise[0], t.to(torch.int32))    # timesteps must be int64 [N]
    tra.step(x0, noise[0], torch.tensor([-5, 5000], device="cuda"))  # out-of-range timesteps are clamped to the schedule
    assert bool(torch.isfinite(tra.loss).all())


def test_ldm_step_matches_oracle_composition():
    """LDMTrainer = the 'vae' branch of train_ldm.py:154-180: no-grad AutoencoderKL.encode_stage_2_inputs -> * scale_factor ->
    q-sample -> UNet -> MSE -> backward, against the same composition of the CPU restatements; scale_factor = 1/std(z) of the first
    batch (train_ldm.py:110-112).  Also LatentDiffusionInferer.__call__ (the autograd-edge form of the same forward)."""
    from medical_image_generation_amd.autoencoderkl import AutoencoderKL
    from medical_image_generation_amd.inferer import DDPMScheduler, LatentDiffusionInferer
    from medical_image_generation_amd.trainer import LDMTrainer
    from medical_image_generation_amd.unet import DiffusionModelUNet
    ca = cases.AEKL_CASES["aekl_c3a"]
    ae_ref = nets.AutoencoderKL(**ca["kwargs"])
    ae_sd = synth.state_dict({k: tuple(v.shape) for k, v in ae_ref.state_dict().items()}, S)
    ae_ref.load_state_dict(ae_sd)
    ae = AutoencoderKL(**ca["kwargs"])
    ae.load_state_dict(ae_sd)
    ae = ae.cuda()
    ukw = dict(cases.UNET_CASES["unet_ldm"]["kwargs"])  # in/out 8 = the AE's latent channels
    u_ref = nets.DiffusionModelUNet(**ukw)
    u_sd = synth.state_dict({k: tuple(v.shape) for k, v in u_ref.state_dict().items()}, S + 1)
    u_ref.load_state_dict(u_sd)
    unet = DiffusionModelUNet(**ukw)
    unet.load_state_dict(u_sd)
    unet = unet.cuda()
    x = synth.ellipsoid_volume(S, "x", (2, 1, 32, 32, 32))
    with torch.no_grad():
        mu, sigma = ae_ref.encode(x)
    eps, noise = synth.tensor(S, "eps", mu.shape), synth.tensor(S, "lnoise", mu.shape)
    t = torch.tensor([250, 750])
    with torch.no_grad():
        z = mu + eps * sigma
    scale_ref = float(1 / torch.std(z))
    sched = step.DDPMSchedule()
    loss_ref, _ = step.ddpm_loss(u_ref, sched, z * scale_ref, noise, t)
    loss_ref.backward()
    tr = LDMTrainer(unet, ae, lr=1e-4)
    with pytest.raises(RuntimeError):
        tr.forward_backward(x.cuda(), eps.cuda(), noise.cuda(), t.cuda())  # scale factor not set yet
    sf = tr.estimate_scale_factor(x.cuda(), eps.cuda())
    assert abs(sf - scale_ref) <= 1e-2 * scale_ref
    tr.forward_backward(x.cuda(), eps.cuda(), noise.cuda(), t.cuda())
    names = [n for n, p in u_ref.named_parameters() if p.grad is not None]
    g_ref = torch.cat([dict(u_ref.named_parameters())[n].grad.flatten() for n in names])
    g_hip = torch.cat([tr.arena.gview(n).cpu().flatten() for n in names])
    e = float((g_hip - g_ref).norm() / g_ref.norm())
    print(f"\n[LDM step] scale_factor {sf:.5f} vs {scale_ref:.5f}; loss {float(tr.loss):.6f} vs {float(loss_ref):.6f}; gradient rel-L2 {e:.3e}")
    assert abs(float(tr.loss) - float(loss_ref)) <= 1.5e-2 * float(loss_ref) and e <= 5e-2
    assert float(tr.ae_arena.grad.abs().max()) == 0.0  # the autoencoder is frozen: nothing reached its gradient arena
    # graph form: encoder + UNet step in one capture
    tr.capture(x.cuda(), eps.cuda(), noise.cuda(), t.cuda())
    lg = float(tr.step_graph())
    assert abs(lg - float(loss_ref)) <= 1.5e-2 * float(loss_ref)
    # the autograd-edge form of the same forward (generative.inferers.LatentDiffusionInferer.__call__)
    inferer = LatentDiffusionInferer(DDPMScheduler(1000, "scaled_linear_beta", beta_start=0.0015, beta_end=0.0205), scale_factor=sf)
    ae.sampling = lambda m, s: m + eps.cuda() * s  # pin the sampling noise
    unet2 = DiffusionModelUNet(**ukw)
    unet2.load_state_dict(u_sd)
    pred = inferer(inputs=x.cuda(), autoencoder_model=ae, diffusion_model=unet2.cuda(), noise=noise.cuda(), timesteps=t.cuda())
    loss2 = torch.nn.functional.mse_loss(pred.float(), noise.cuda())
    assert abs(float(loss2) - float(loss_ref)) <= 1.5e-2 * float(loss_ref)
