"""Generate tests/golden/*.safetensors by running the REFERENCE's own model files.

ORACLE TOOLING -- build container only (needs /root/reference; never runs on the GPU
box).  Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.tools.gen_golden

What comes from where:
  * network code, constructor logic, init, forward/backward: the reference's
    `diffusion_model_unet_with_strides.py` / `autoencoderkl_with_strides.py`, unmodified,
    loaded by path (they only need the 4-symbol `monai` stand-in, oracle/tools/monai_standin.py);
  * weights / inputs / upstream grads: oracle.synth (seeded by name), every tensor randomised;
  * optimizer + clipping: torch.optim.Adam[W] + clip_grad_norm_ (what T-LDM:121,177 call);
  * q-sample / KL closed forms: oracle.step (third-party `generative` is absent -> unpinned).
"""
from __future__ import annotations

import importlib.util
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import torch
from safetensors.torch import save_file

from oracle import cases, step, synth
from oracle.tools import monai_standin

REF = "/root/reference/medimgen"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def _load(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _shapes(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def _grads(module):
    return {n: p.grad.detach().clone() for n, p in module.named_parameters() if p.grad is not None}


def _save(name, tensors, meta=None):
    os.makedirs(OUT, exist_ok=True)
    tensors = {k: v.detach().contiguous() for k, v in tensors.items()}
    save_file(tensors, os.path.join(OUT, name + ".safetensors"), metadata=meta or {})
    size = sum(v.numel() * v.element_size() for v in tensors.values())
    print(f"  {name}: {len(tensors)} tensors, {size / 1e6:.2f} MB")


def known_answers(ru):
    t = torch.tensor([0, 1, 999])
    out = {"timestep_embedding_8": ru.get_timestep_embedding(t, 8),
           "timestep_embedding_7": ru.get_timestep_embedding(t, 7),
           "timestep_embedding_32": ru.get_timestep_embedding(torch.tensor([17, 903]), 32)}
    # pristine-init facts: output identically zero; proj_attn never gets a grad
    c = cases.UNET_CASES["unet3d"]
    torch.manual_seed(0)
    net = ru.DiffusionModelUNet(**c["kwargs"])
    x = synth.tensor(cases.SEED, "x", c["shape"])
    y = net(x, torch.tensor(c["timesteps"]))
    out["pristine_out_absmax"] = y.abs().max().reshape(1)
    (y.sum() + sum(p.sum() for p in net.parameters()) * 0).backward()
    net.zero_grad()
    net.load_state_dict(synth.state_dict(_shapes(net), cases.SEED))
    net(x, torch.tensor(c["timesteps"])).square().mean().backward()
    gradless = sorted(n for n, p in net.named_parameters() if p.grad is None)
    _save("known_answers", out, {"gradless_params": "\n".join(gradless)})


def unet_case(ru, name):
    c = cases.UNET_CASES[name]
    net = ru.DiffusionModelUNet(**c["kwargs"])
    net.load_state_dict(synth.state_dict(_shapes(net), cases.SEED))
    x = synth.tensor(cases.SEED, "x", c["shape"])
    t = torch.tensor(c["timesteps"])
    extra = {}
    if "class_labels" in c:
        extra["class_labels"] = torch.tensor(c["class_labels"])
    if "context" in c:
        extra["context"] = synth.tensor(cases.SEED, "context", c["context"])
    pred = net(x, t, **extra)
    gy = synth.tensor(cases.SEED, "grad_out", pred.shape)
    xg = x.clone().requires_grad_(True)
    net.zero_grad()
    net(xg, t, **extra).backward(gy)
    out = {"pred": pred, "dx": xg.grad}
    g = _grads(net)
    s = synth.summarise(g, cases.SEED)
    out["grad_norm"], out["grad_dot"] = s["norm"], s["dot"]
    # full gradients of a few structurally distinct tensors (debugging aid for the HIP path)
    for key in sorted(g):
        if g[key].numel() <= 4096:
            out["grad:" + key] = g[key]
    _save(name, out, {"grad_names": "\n".join(sorted(g))})


def unet_steps(ru, name, opt_name):
    c = cases.UNET_CASES[name]
    net = ru.DiffusionModelUNet(**c["kwargs"])
    net.load_state_dict(synth.state_dict(_shapes(net), cases.SEED))
    sched = step.DDPMSchedule()
    opt = getattr(torch.optim, opt_name)(net.parameters(), lr=cases.STEP_LR)
    x0 = synth.ellipsoid_volume(cases.SEED, "x0", c["shape"])
    t = torch.tensor(c["timesteps"])
    losses = []
    for k in range(cases.STEP_COUNT):
        noise = synth.tensor(cases.SEED, f"noise{k}", c["shape"])
        loss, _ = step.ddpm_train_step(net, opt, sched, x0, noise, (t + 37 * k) % 1000, max_norm=1.0)
        losses.append(loss)
    s = synth.summarise({k: v.detach() for k, v in net.state_dict().items()}, cases.SEED)
    _save(name + "_steps", {"losses": torch.stack(losses), "param_norm": s["norm"], "param_dot": s["dot"]},
          {"optimizer": opt_name, "names": "\n".join(sorted(net.state_dict()))})


def aekl_case(ra, name):
    c = cases.AEKL_CASES[name]
    net = ra.AutoencoderKL(**c["kwargs"])
    net.load_state_dict(synth.state_dict(_shapes(net), cases.SEED))
    x = synth.ellipsoid_volume(cases.SEED, "x", c["shape"])
    z_mu, z_sigma = net.encode(x)
    eps = synth.tensor(cases.SEED, "eps", z_mu.shape)
    # AutoencoderKL.forward with the sampling noise made explicit (AEKL:786-787, 821-825)
    recon = net.decode(z_mu + eps * z_sigma)
    loss = torch.nn.functional.l1_loss(recon, x) + step.kl_loss(z_mu, z_sigma) * cases.KL_WEIGHT
    net.zero_grad()
    loss.backward()
    g = _grads(net)
    s = synth.summarise(g, cases.SEED)
    out = {"z_mu": z_mu, "z_sigma": z_sigma, "recon": recon, "loss": loss.reshape(1),
           "kl": step.kl_loss(z_mu, z_sigma).reshape(1), "grad_norm": s["norm"], "grad_dot": s["dot"]}
    for key in sorted(g):
        if g[key].numel() <= 4096:
            out["grad:" + key] = g[key]
    _save(name, out, {"grad_names": "\n".join(sorted(g))})


def block_cases(ru, ra):
    """Per-block vectors with FULL tensors (small), for op-level checks of the HIP kernels."""
    S = cases.SEED

    def run(tag, mod, inputs, fwd):
        mod.load_state_dict(synth.state_dict(_shapes(mod), S))
        leaves = {k: v.clone().requires_grad_(True) for k, v in inputs.items()}
        y = fwd(mod, leaves)
        gy = synth.tensor(S, tag + ":gy", y.shape)
        y.backward(gy)
        out = {"in:" + k: v for k, v in inputs.items()}
        out["out"] = y
        for k, v in leaves.items():
            out["din:" + k] = v.grad
        for n, p in mod.named_parameters():
            if p.grad is not None:
                out["dparam:" + n] = p.grad
        _save("block_" + tag, out)

    x = synth.tensor(S, "bx", (2, 32, 8, 8, 8))
    emb = synth.tensor(S, "bemb", (2, 128))
    run("resnet3d_32_64", ru.ResnetBlock(3, 32, 128, 64, norm_num_groups=32), {"x": x, "emb": emb},
        lambda m, i: m(i["x"], i["emb"]))
    run("resnet3d_32_32", ru.ResnetBlock(3, 32, 128, 32, norm_num_groups=32), {"x": x, "emb": emb},
        lambda m, i: m(i["x"], i["emb"]))
    xa = synth.tensor(S, "bxa", (2, 64, 4, 4, 4))
    run("attn3d_64_h32", ru.AttentionBlock(3, 64, num_head_channels=32, norm_num_groups=32), {"x": xa},
        lambda m, i: m(i["x"]))
    run("attn3d_64_h64", ru.AttentionBlock(3, 64, num_head_channels=64, norm_num_groups=32), {"x": xa},
        lambda m, i: m(i["x"]))
    run("down3d_32", ru.Downsample(3, 32, use_conv=True, out_channels=32, stride=[2] * 3, kernel_size=[3] * 3,
                                   padding=[1] * 3), {"x": x}, lambda m, i: m(i["x"]))
    run("up3d_32", ru.Upsample(3, 32, use_conv=True, out_channels=32, stride=[2] * 3, padding=[1] * 3),
        {"x": synth.tensor(S, "bxu", (2, 32, 4, 4, 4))}, lambda m, i: m(i["x"]))
    run("ae_res3d_16_32", ra.ResBlock(3, 16, 8, 1e-6, 32), {"x": synth.tensor(S, "bxr", (2, 16, 8, 8, 8))},
        lambda m, i: m(i["x"]))
    x2 = synth.tensor(S, "bx2", (2, 32, 16, 16))
    run("resnet2d_32_64", ru.ResnetBlock(2, 32, 128, 64, norm_num_groups=32), {"x": x2, "emb": emb},
        lambda m, i: m(i["x"], i["emb"]))


def main():
    assert os.path.isdir(REF), "reference not mounted: goldens can only be generated in the build container"
    monai_standin.install()
    torch.set_num_threads(os.cpu_count() or 1)
    ru = _load("_ref_unet", "diffusion_model_unet_with_strides.py")
    ra = _load("_ref_aekl", "autoencoderkl_with_strides.py")
    print("golden vectors ->", OUT)
    only = sys.argv[1:]  # optional: regenerate just the named UNet cases (new cases leave the committed fixtures untouched)
    if only:
        for name in only:
            unet_case(ru, name)
        return
    known_answers(ru)
    for name in cases.UNET_CASES:
        unet_case(ru, name)
    for name, opt in cases.STEP_CASES.items():
        unet_steps(ru, name, opt)
    for name in cases.AEKL_CASES:
        aekl_case(ra, name)
    block_cases(ru, ra)


if __name__ == "__main__":
    main()
