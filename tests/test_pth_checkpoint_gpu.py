"""Checkpoints in the reference's .pth layout (train_ldm.py:466-505) cross the boundary in both directions: a run of the oracle with
torch.optim.AdamW resumes on the fused HIP optimizer, and a run of the HIP trainer resumes under torch.optim.AdamW."""
import pytest
import torch

from oracle import cases, nets, step, synth

pytestmark = pytest.mark.gpu
S = cases.SEED
NAME = "unet3d"


def _pair():
    from medical_image_generation_amd.unet import DiffusionModelUNet
    c = cases.UNET_CASES[NAME]
    ref = nets.DiffusionModelUNet(**c["kwargs"])
    sd = synth.state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, S)
    ref.load_state_dict(sd)
    net = DiffusionModelUNet(**c["kwargs"])
    net.load_state_dict(sd)
    return c, ref, net.cuda()


def _inputs(c, k):
    return synth.ellipsoid_volume(S, "x0", c["shape"]), synth.tensor(S, f"noise{k}", c["shape"]), (torch.tensor(c["timesteps"]) + 37 * k) % 1000


def _cos(a, b):
    return float(torch.dot(a, b) / (a.norm() * b.norm()))


def _flat(sd, names):
    return torch.cat([sd[n].detach().cpu().flatten().float() for n in names])


def test_reference_checkpoint_resumes_on_the_fused_optimizer(tmp_path):
    from medical_image_generation_amd import checkpoint as ck
    from medical_image_generation_amd.trainer import DDPMTrainer
    c, ref, net = _pair()
    opt = torch.optim.AdamW(ref.parameters(), lr=cases.STEP_LR)
    sched = step.DDPMSchedule()
    for k in range(2):
        step.ddpm_train_step(ref, opt, sched, *_inputs(c, k), max_norm=1.0)
    path = tmp_path / "last_model.pth"  # what train_ldm.py:466-479 writes
    torch.save({"epoch": 4, "network_state_dict": ref.state_dict(), "optimizer_state_dict": opt.state_dict(), "validation_loss": 0.5}, path)
    tr = DDPMTrainer(net, lr=123.0, optimizer="AdamW", max_grad_norm=1.0)  # lr comes from the checkpoint's param group
    assert ck.load_model(tr, str(path), for_training=True) == 5
    assert tr.lr == cases.STEP_LR and float(tr.step_count) == 2.0
    names = [n for n, _ in ref.named_parameters() if ".proj_attn." not in n]
    before = _flat(ref.state_dict(), names)
    assert torch.equal(_flat(net.state_dict(), names), before)
    # the moments landed where the fused kernel reads them
    sd_o = ck.optimizer_state_dict(tr)
    idx = {n: i for i, (n, _) in enumerate(ref.named_parameters())}
    for n in names[::17]:
        assert torch.equal(sd_o["state"][idx[n]]["exp_avg"], opt.state_dict()["state"][idx[n]]["exp_avg"])
    assert not any(idx[n] in sd_o["state"] for n, _ in ref.named_parameters() if ".proj_attn." in n)
    # third step on both sides
    x0, noise, t = _inputs(c, 2)
    step.ddpm_train_step(ref, opt, sched, x0, noise, t, max_norm=1.0)
    tr.step(x0.cuda(), noise.cuda(), t.cuda())
    upd_ref, upd_hip = _flat(ref.state_dict(), names) - before, _flat(net.state_dict(), names) - before
    cos = _cos(upd_ref, upd_hip)
    print(f"\nthird-step update after resume: cosine {cos:.4f} |ref| {float(upd_ref.norm()):.5f} |hip| {float(upd_hip.norm()):.5f}")
    # with two steps of shared Adam history the third update is dominated by the loaded moments: a lost or misplaced state shows as cos << 1
    assert cos > 0.98 and abs(float(upd_hip.norm()) / float(upd_ref.norm()) - 1) < 0.05


def test_hip_checkpoint_resumes_under_torch_optim(tmp_path):
    from medical_image_generation_amd import checkpoint as ck
    from medical_image_generation_amd.trainer import DDPMTrainer
    c, ref, net = _pair()
    tr = DDPMTrainer(net, lr=cases.STEP_LR, optimizer="AdamW", max_grad_norm=1.0)
    assert ck.optimizer_state_dict(tr)["state"] == {}  # like torch before the first update
    for k in range(2):
        x0, noise, t = _inputs(c, k)
        tr.step(x0.cuda(), noise.cuda(), t.cuda())
    last = ck.save_model(tr, str(tmp_path), epoch=7, validation_loss=0.25)
    assert last.endswith("checkpoints/last_model.pth") and (tmp_path / "checkpoints" / "best_model.pth").is_file()
    ck.save_model(tr, str(tmp_path), epoch=8, validation_loss=0.9)  # worse: best_model.pth keeps epoch 7 (train_ldm.py:482-488)
    assert torch.load(tmp_path / "checkpoints" / "best_model.pth", weights_only=True)["epoch"] == 7
    assert torch.load(last, weights_only=True)["epoch"] == 8
    ckpt = torch.load(last, weights_only=True)  # the reference's load_model (train_ldm.py:492-505) on a torch module + torch optimizer
    assert set(ckpt) == {"epoch", "network_state_dict", "optimizer_state_dict", "validation_loss"}
    ref.load_state_dict(ckpt["network_state_dict"])
    opt = torch.optim.AdamW(ref.parameters(), lr=1.0)
    opt.load_state_dict(ckpt["optimizer_state_dict"])
    assert opt.param_groups[0]["lr"] == cases.STEP_LR and set(opt.state_dict()["param_groups"][0]) == set(ckpt["optimizer_state_dict"]["param_groups"][0])
    names = [n for n, _ in ref.named_parameters() if ".proj_attn." not in n]
    before = _flat(ref.state_dict(), names)
    x0, noise, t = _inputs(c, 2)
    step.ddpm_train_step(ref, opt, step.DDPMSchedule(), x0, noise, t, max_norm=1.0)
    tr.step(x0.cuda(), noise.cuda(), t.cuda())
    upd_ref, upd_hip = _flat(ref.state_dict(), names) - before, _flat(net.state_dict(), names) - before
    cos = _cos(upd_ref, upd_hip)
    print(f"\nthird-step update, torch.optim on the HIP trainer's state: cosine {cos:.4f}")
    assert cos > 0.98 and abs(float(upd_hip.norm()) / float(upd_ref.norm()) - 1) < 0.05
    # a checkpoint of another network is refused like torch refuses it
    with pytest.raises(ValueError):
        bad = ckpt["optimizer_state_dict"]
        bad["param_groups"][0]["params"] = bad["param_groups"][0]["params"][:-1]
        ck.load_optimizer_state_dict(tr, bad)
