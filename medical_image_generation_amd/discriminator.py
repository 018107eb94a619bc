"""PatchDiscriminator and the least-squares adversarial loss on the MI355X HIP path.

The reference's GAN step (medimgen/train_autoencoder.py:371-397 `train_discriminator_step`, :416-423 the generator's adversarial term,
:41 `PatchAdversarialLoss(criterion="least_squares")`, :600 `PatchDiscriminator(**config['discriminator_params'])` with the planner's
`{'spatial_dims', 'in_channels', 'out_channels': 1, 'num_channels': 64, 'num_layers_d': 3}`, configuration.py:966-967) takes both
classes from the third-party `generative` package, whose source is not under /root/reference: the classes here mirror upstream's
constructor surface, `state_dict()` names (`initial_conv.conv.*`, `<l>.conv.weight`, `<l>.adn.N.*`, `final_conv.conv.*`), initialisation
(conv weights N(0, 0.02), BatchNorm weights N(1, 0.02), biases 0) and forward (the list of every layer's output) as restated in
oracle/disc.py -- PARITY UNPINNED against the original.

Kernels: csrc/disc.hip -- the k4 convs (stride 2 / 1, padding 1) run as patch matrix + the library's bf16 NT GEMM, BatchNorm in
training mode as GroupNorm with one channel per group on the [1, N*V, C] view (activation code 2 = LeakyReLU(0.2)), the loss as one
small kernel.  3-D only (the 2-D planner configs keep upstream's torch discriminator through AETrainer(extra_loss=...)).
"""
from __future__ import annotations

import torch
from torch import nn

from . import engine as E
from . import hipops as ops
from ._lib import call, ptr
from .unet import HipModule, ParamSpec, _NetFn

BF16, F32 = torch.bfloat16, torch.float32


def _up8(n):
    return (n + 7) // 8 * 8


def conv_gemm(ctx: E.Ctx, x, name, k, s, p, bias: bool, need_dx=True, param_grads=True):
    """k^3 conv with stride s / padding p on NDHWC bf16 through im2col + NT GEMM (weight `name.weight` [Cout, Cin, k, k, k]).  Returns
    y [N, Do, Ho, Wo, Cout rounded up to 8] (pad channels zero: zero weight rows, zero bias)."""
    n, d, h, w, cin = x.shape
    wt = ctx.p(name + ".weight")
    cout, taps = wt.shape[0], k * k * k
    cop, kk = _up8(cout), taps * cin
    do, ho, wo = ((v + 2 * p - k) // s + 1 for v in (d, h, w))
    m = n * do * ho * wo
    dev = x.device
    w2 = torch.empty((cop, kk), dtype=BF16, device=dev)
    w2t = torch.empty((kk, cop), dtype=BF16, device=dev)
    call("mi_disc_pack_weights", ptr(wt), ptr(w2), ptr(w2t), cout, cop, cin, taps)
    xd = ops.dense(x)
    patches = torch.empty((m, kk), dtype=BF16, device=dev)
    call("mi_im2col3d", ptr(xd), cin, ptr(patches), n, d, h, w, cin, k, s, p)
    b = None
    if bias:
        b = ctx.p(name + ".bias")
        if cop != cout:  # padded copy of the bias (the final conv: one output channel)
            b8 = torch.empty((1, cop), dtype=F32, device=dev)
            ops.zero_f32_2d_(b8)
            ops.add_f32_(b8[:, :cout], b.view(1, cout))
            b = b8
    y = torch.empty((n, do, ho, wo, cop), dtype=BF16, device=dev)
    E._gemm(patches, kk, 0, 0, w2, kk, 0, 0, y, cop, 0, 0, m, cop, kk, 1, 1, bias=b)
    ctx.count(2 * m * cout * kk, dgrad=need_dx, wgrad=param_grads)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            dy = ops.dense(dy)
            if param_grads:
                if bias:
                    cs = ops.colsum(dy, merge_batch=True)  # [1, cop]
                    ops.add_f32_(ctx.g(name + ".bias").view(1, cout), cs[:, :cout])
                mp = _up8(m)  # the GEMM reduces over the voxel axis: pitch rounded up to 8, pad columns zeroed by the transpose kernel
                dy_t = torch.empty((cop, mp), dtype=BF16, device=dev)
                p_t = torch.empty((kk, mp), dtype=BF16, device=dev)
                call("mi_transpose_bf16", ptr(dy), cop, 0, 0, ptr(dy_t), mp, 0, 0, m, cop, 1, 1)
                call("mi_transpose_bf16", ptr(patches), kk, 0, 0, ptr(p_t), mp, 0, 0, m, kk, 1, 1)
                dw2 = torch.empty((cop, kk), dtype=F32, device=dev)
                E._gemm(dy_t, mp, 0, 0, p_t, mp, 0, 0, dw2, kk, 0, 0, cop, kk, mp, 1, 1)
                call("mi_disc_wgrad_unpack", ptr(dw2), ptr(ctx.g(name + ".weight")), cout, cin, taps)
            if need_dx:
                dpatches = torch.empty((m, kk), dtype=BF16, device=dev)
                E._gemm(dy, cop, 0, 0, w2t, cop, 0, 0, dpatches, kk, 0, 0, m, kk, cop, 1, 1)
                dx = torch.empty((n, d, h, w, cin), dtype=BF16, device=dev)
                call("mi_col2im3d", ptr(dpatches), ptr(dx), cin, n, d, h, w, cin, k, s, p)
                tape.put(x, dx)

        tape.record(bwd)
    return y


def leaky_relu(ctx: E.Ctx, x, slope=0.2):
    y = torch.empty_like(x)
    call("mi_leaky_relu_fwd", ptr(x), ptr(y), x.numel(), float(slope))
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            dy = tape.take(y)
            if dy is not None:
                dx = torch.empty_like(x)
                call("mi_leaky_relu_bwd", ptr(x), ptr(ops.dense(dy)), ptr(dx), x.numel(), float(slope))
                tape.put(x, dx)

        tape.record(bwd)
    return y


def batchnorm_leaky(ctx: E.Ctx, x, name, buffers, eps=1e-5, momentum=0.1, param_grads=True):
    """nn.BatchNorm3d in TRAINING mode (batch statistics over N, D, H, W; the running buffers are updated) followed by LeakyReLU(0.2):
    GroupNorm with one channel per group on the [1, N*D, H, W, C] view of the channels-last tensor."""
    n, d, h, w, c = x.shape
    xv = x.view(1, n * d, h, w, c)
    gamma, beta = ctx.p(name + ".weight"), ctx.p(name + ".bias")
    st = ops.gn_stats(xv, c, eps, gamma, beta)
    y = ops.gn_apply(xv, st, 2).view(x.shape)
    rm, rv, nbt = buffers
    call("mi_bn_running_update", ptr(st.mean_rstd), ptr(rm), ptr(rv), ptr(nbt), c, float(eps), float(momentum), n * d * h * w)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            g = tape.take(y)
            if g is None:
                return
            gv = ops.dense(g).view(xv.shape)
            # (with the discriminator frozen -- the generator step -- the parameter gradients land in scratch that nobody reads)
            dgam = ctx.g(name + ".weight") if param_grads else torch.empty(c, dtype=F32, device=x.device)
            dbet = ctx.g(name + ".bias") if param_grads else torch.empty(c, dtype=F32, device=x.device)
            dx = ops.gn_bwd(gv, xv, st, gamma, 2, dgam, dbet)
            tape.put(x, dx.view(x.shape))

        tape.record(bwd)
    return y


class PatchDiscriminator(HipModule):
    """`generative.networks.nets.PatchDiscriminator(spatial_dims, num_channels, in_channels, out_channels=1, num_layers_d=3, kernel_size=4,
    activation=LeakyReLU(0.2), norm="BATCH", bias=False, padding=1, dropout=0.0, last_conv_kernel_size=None)`; forward returns the list
    of the outputs of initial_conv, layers 0 .. num_layers_d-1 and final_conv (the reference takes `[-1]`, the patch logits)."""

    def __init__(self, spatial_dims: int, num_channels: int, in_channels: int, out_channels: int = 1, num_layers_d: int = 3, kernel_size: int = 4,
                 activation=("LEAKYRELU", {"negative_slope": 0.2}), norm="BATCH", bias: bool = False, padding: int = 1, dropout: float = 0.0,
                 last_conv_kernel_size: int | None = None) -> None:
        super().__init__()
        if spatial_dims != 3:
            raise NotImplementedError("the HIP PatchDiscriminator is 3-D (2-D configs: upstream's torch module through AETrainer(extra_loss=))")
        if str(norm).upper() != "BATCH" or dropout != 0.0 or bias or out_channels != 1:
            raise NotImplementedError("only the configuration the reference builds: norm='BATCH', bias=False, dropout=0, out_channels=1")
        act_name = activation[0] if isinstance(activation, (tuple, list)) else activation
        slope = activation[1].get("negative_slope", 0.01) if isinstance(activation, (tuple, list)) and len(activation) > 1 else 0.01
        if str(act_name).upper() != "LEAKYRELU" or abs(slope - 0.2) > 1e-12:
            raise NotImplementedError("only LeakyReLU(0.2) (upstream's default)")
        self.spatial_dims, self.in_channels, self.out_channels = spatial_dims, in_channels, out_channels
        self.num_channels, self.num_layers_d = num_channels, num_layers_d
        self.kernel_size, self.padding = kernel_size, padding
        self.last_k = kernel_size if last_conv_kernel_size is None else last_conv_kernel_size
        spec = ParamSpec(self, 3)

        def conv(name, cin, cout, k, with_bias):
            w = torch.empty(cout, cin, k, k, k)
            nn.init.normal_(w, 0.0, 0.02)  # initialise_weights: Conv3d weights N(0, 0.02); the bias keeps nn.Conv3d's default
            spec._add(name + ".weight", w)
            if with_bias:
                bound = 1 / (cin * k ** 3) ** 0.5
                spec._add(name + ".bias", torch.empty(cout).uniform_(-bound, bound))

        conv("initial_conv.conv", in_channels, num_channels, kernel_size, True)
        self._layers = []  # (name, cin, cout, stride)
        cin, cout = num_channels, num_channels * 2
        for l in range(num_layers_d):
            conv(f"{l}.conv", cin, cout, kernel_size, False)
            spec._add(f"{l}.adn.N.weight", torch.empty(cout).normal_(1.0, 0.02))
            spec._add(f"{l}.adn.N.bias", torch.zeros(cout))
            mod = self._modules[str(l)]._modules["adn"]._modules["N"]
            mod.register_buffer("running_mean", torch.zeros(cout))
            mod.register_buffer("running_var", torch.ones(cout))
            mod.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
            self._layers.append((str(l), cin, cout, 1 if l == num_layers_d - 1 else 2))
            cin, cout = cout, cout * 2
        conv("final_conv.conv", cin, out_channels, self.last_k, True)
        self._init_plumbing(spec, [])

    def _buffers_of(self, l):
        mod = self._modules[l]._modules["adn"]._modules["N"]
        return mod.running_mean, mod.running_var, mod.num_batches_tracked

    def _run(self, c: E.Ctx, x_cl, need_dx=True, param_grads=True):
        """-> [outputs]; the last one is the logits tensor [N, d, h, w, 8] whose channel 0 (.. out_channels) is real, the rest padding."""
        k, p = self.kernel_size, self.padding
        outs = []
        h = conv_gemm(c, x_cl, "initial_conv.conv", k, 2, p, True, need_dx=need_dx, param_grads=param_grads)
        h = leaky_relu(c, h, 0.2)
        outs.append(h)
        for name, _, _, stride in self._layers:
            h = conv_gemm(c, h, name + ".conv", k, stride, p, False, param_grads=param_grads)
            h = batchnorm_leaky(c, h, name + ".adn.N", self._buffers_of(name), param_grads=param_grads)
            outs.append(h)
        outs.append(conv_gemm(c, h, "final_conv.conv", self.last_k, 1, (self.last_k - 1) // 2, True, param_grads=param_grads))
        return outs

    def forward(self, x: torch.Tensor):
        if x.shape[1] != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {x.shape[1]}")
        if not x.is_cuda:
            raise RuntimeError("medical_image_generation_amd runs on MI355X only: move the module and inputs to 'cuda' (no CPU fallback)")
        oc = self.out_channels

        def runner(c, xin, need_dx):
            x_cl = ops.to_channels_last(xin.contiguous().float())
            outs = self._run(c, x_cl, need_dx=need_dx)
            if outs[-1].shape[-1] != oc:  # drop the padding channels of the logits
                outs[-1] = _slice_first(c, outs[-1], oc)
            return tuple(outs), {"x_cl": x_cl}

        grad_enabled = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        return list(_NetFn.apply(self, runner, self.num_layers_d + 2, grad_enabled, x, None, *self.parameters()))


def _slice_first(ctx: E.Ctx, x, nc):
    """The logits channel of the padded [N, d, h, w, 8] tensor as a dense [N, d, h, w, 1] tensor (a transpose: row 0 of [8, voxels])."""
    assert nc == 1
    n, d, h, w, c = x.shape
    m = n * d * h * w
    t = torch.empty((c, m), dtype=BF16, device=x.device)
    call("mi_transpose_bf16", ptr(x), c, 0, 0, ptr(t), m, 0, 0, m, c, 1, 1)
    y = t[:1].view(n, d, h, w, 1)
    if ctx.tape is not None:
        tape = ctx.tape

        def bwd():
            g = tape.take(y)
            if g is not None:  # [1, m] -> [m, 8]: the transpose kernel zero-fills the pad columns 1..7
                back = torch.empty((m, c), dtype=BF16, device=x.device)
                call("mi_transpose_bf16", ptr(ops.dense(g)), m, 0, 0, ptr(back), c, 0, 0, 1, m, 1, 1)
                tape.put(x, back.view(x.shape))

        tape.record(bwd)
    return y


class PatchAdversarialLoss:
    """`generative.losses.PatchAdversarialLoss(criterion="least_squares")` on the logits of ONE discriminator: MSE between
    LeakyReLU(0.05)(logits) (upstream's activation in front of the least-squares criterion; `no_activation_leastsq=True` drops it) and a
    constant target -- 1 for real / for the generator, 0 for fake.  `hip(logits_cl, ...)` is the fused form on a channels-last logits
    tensor (one kernel: loss accumulation + gradient); `__call__` the torch-tensor form for callers outside the fused trainers."""

    def __init__(self, criterion: str = "least_squares", no_activation_leastsq: bool = False):
        if criterion != "least_squares":
            raise NotImplementedError("only criterion='least_squares' (train_autoencoder.py:41)")
        self.slope = 1.0 if no_activation_leastsq else 0.05
        self.real_label, self.fake_label = 1.0, 0.0

    def hip(self, logits_cl, target_is_real: bool, loss_acc, weight: float, want_grad=True):
        """*loss_acc += weight * loss; returns d(weight * loss)/d(logits) (channels-last, padding channels zero) or None."""
        n, d, h, w, cs = logits_cl.shape
        dl = torch.empty_like(logits_cl) if want_grad else None
        call("mi_ls_gan_loss", ptr(logits_cl), cs, n * d * h * w, self.real_label if target_is_real else self.fake_label, self.slope, ptr(dl),
             ptr(loss_acc), float(weight))
        return dl

    def __call__(self, logits, target_is_real: bool, for_discriminator: bool = False):
        a = torch.nn.functional.leaky_relu(logits.float(), self.slope) if self.slope != 1.0 else logits.float()
        return torch.nn.functional.mse_loss(a, torch.full_like(a, self.real_label if target_is_real else self.fake_label))
