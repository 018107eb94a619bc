"""Layout of the optimizer state the package writes into the reference's .pth checkpoints (train_ldm.py:466-505): it is what
torch.optim.AdamW.state_dict() would hold for the same module, and it round-trips through a real torch optimizer.  (Host logic only:
the flat moment buffers are stood in for by CPU tensors; the GPU tests run the real trainers.)"""
import math

import torch

from oracle import cases


class _FakeTrainer:
    def __init__(self, net, step):
        self.model, self.arena = net, net.arena(torch.device("cpu"))
        n = self.arena.n_trainable
        g = torch.Generator().manual_seed(0)
        self.exp_avg, self.exp_avg_sq = torch.randn(n, generator=g), torch.rand(n, generator=g)
        self.step_count = torch.tensor([float(step)])
        self.lr, self.betas, self.eps, self.weight_decay, self.decoupled = 1e-4, (0.9, 0.999), 1e-8, 0.01, True


def test_optimizer_state_dict_is_torch_adamw_layout():
    from medical_image_generation_amd import checkpoint as ck
    from medical_image_generation_amd.unet import DiffusionModelUNet
    net = DiffusionModelUNet(**cases.UNET_CASES["unet3d"]["kwargs"])
    t = _FakeTrainer(net, 3)
    sd = ck.optimizer_state_dict(t)
    names = [n for n, _ in net.named_parameters()]
    unused = {i for i, n in enumerate(names) if ".proj_attn." in n}
    assert sd["param_groups"][0]["params"] == list(range(len(names))) and set(sd["state"]) == set(range(len(names))) - unused
    opt = torch.optim.AdamW(net.parameters(), lr=1.0)
    assert set(opt.state_dict()["param_groups"][0]) == set(sd["param_groups"][0])
    opt.load_state_dict(sd)  # torch accepts it: shapes, ids and group sizes fit
    assert opt.param_groups[0]["lr"] == 1e-4 and all(tuple(opt.state[p]["exp_avg"].shape) == tuple(p.shape) for p in opt.state)
    want = t.exp_avg.clone()
    t.exp_avg, t.exp_avg_sq, t.step_count = torch.zeros_like(want), torch.zeros_like(want), torch.zeros(1)
    ck.load_optimizer_state_dict(t, opt.state_dict())
    covered = torch.zeros(len(want), dtype=torch.bool)  # alignment padding between tensors belongs to no parameter
    for name, _, trainable in net._entries:
        if trainable:
            o = t.arena.offsets[name]
            covered[o:o + math.prod(t.arena.shapes[name])] = True
    assert torch.equal(t.exp_avg[covered], want[covered]) and float(t.step_count) == 3.0
    assert ck.optimizer_state_dict(_FakeTrainer(net, 0))["state"] == {}


def test_checkpoint_with_numpy_validation_loss_loads(tmp_path):
    """The reference writes `validation_loss` as its validation loop returned it: a numpy float after np.mean.  The safe loader takes it."""
    import numpy as np
    from medical_image_generation_amd import checkpoint as ck
    path = tmp_path / "best_model.pth"
    torch.save({"epoch": 3, "validation_loss": np.float64(0.25), "network_state_dict": {"w": torch.ones(2)}}, path)
    got = ck._load(str(path))
    assert got["epoch"] == 3 and float(got["validation_loss"]) == 0.25 and torch.equal(got["network_state_dict"]["w"], torch.ones(2))
