"""Kernel-level parity: every HIP kernel family, through the C ABI, against a plain PyTorch fp32 reference of the
same op on the same bf16-rounded inputs.  Tolerance: one bf16 rounding of the output (2^-8 relative) plus
accumulation-order noise -> |err| <= 1e-2 * max|ref| unless stated otherwise."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

dev = torch.device("cuda")


@pytest.fixture(scope="module")
def ops():
    from medical_image_generation_amd import hipops
    return hipops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()  # bf16-representable fp32


def cl(x):  # NCDHW fp32 (cpu) -> NDHWC bf16 (gpu), via torch (test-side only)
    return x.permute(0, 2, 3, 4, 1).contiguous().to(dev, torch.bfloat16)


def cf(y):  # NDHWC bf16 (gpu) -> NCDHW fp32 (cpu)
    return y.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def check(got, ref, tol=1e-2, what=""):
    scale = float(ref.abs().max()) + 1e-12
    err = float((got - ref).abs().max())
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g}"


def test_layout_roundtrip(ops):
    for shape in [(2, 1, 5, 6, 7), (1, 8, 4, 4, 4), (2, 32, 3, 5, 9)]:
        x = rnd(*shape)
        y = ops.to_channels_last(x.to(dev))
        assert torch.equal(y.cpu(), x.permute(0, 2, 3, 4, 1).bfloat16())
        assert torch.equal(ops.to_channels_first(y).cpu(), x)
    x2 = rnd(2, 3, 8, 8)
    assert torch.equal(ops.to_channels_first(ops.to_channels_last(x2.to(dev)), 2).cpu(), x2)


def test_elementwise_family(ops):
    a, b = rnd(2, 16, 4, 5, 6), rnd(2, 16, 4, 5, 6, seed=1)
    check(cf(ops.add(cl(a), cl(b))), (a + b).bfloat16().float(), 1e-6, "add")
    c = rnd(2, 24, 4, 5, 6, seed=2)
    cat = ops.concat_channels(cl(a), cl(c))
    assert torch.equal(cf(cat), torch.cat([a, c], 1))
    assert torch.equal(cf(ops.slice_channels(cat, 16, 24)), c)
    for f in [(2, 2, 2), (2, 2, 1), (1, 2, 2)]:
        up = ops.upsample_nearest(cl(a), f)
        ref = F.interpolate(a, scale_factor=tuple(float(v) for v in f), mode="nearest")
        assert torch.equal(cf(up), ref)
        g = rnd(*ref.shape, seed=3)
        av = a.clone().requires_grad_(True)
        F.interpolate(av, scale_factor=tuple(float(v) for v in f), mode="nearest").backward(g)
        check(cf(ops.upsample_nearest_bwd(cl(g), f)), av.grad, 1e-2, "upsample bwd")
    t = torch.tensor([0, 1, 17, 999], device=dev)
    for dim in (8, 7, 32, 128):
        from oracle.nets import timestep_embedding
        check(ops.timestep_embedding(t, dim).cpu(), timestep_embedding(t.cpu(), dim), 5e-5, f"timestep embedding {dim}")
    v = rnd(3, 40)
    check(ops.silu_f32(v.to(dev)).cpu(), F.silu(v), 1e-5, "silu")
    vv = v.clone().requires_grad_(True)
    F.silu(vv).backward(torch.ones_like(v) * 0.5)
    vd, hd = v.to(dev), torch.full_like(v, 0.5).to(dev)
    check(ops.silu_bwd_f32(vd, hd).cpu(), vv.grad, 1e-5, "silu bwd")
    x = rnd(2, 64, 3, 7, 5)
    check(ops.colsum(cl(x)).cpu(), x.sum(dim=(2, 3, 4)), 1e-3, "colsum")
    x1 = rnd(2, 1, 9, 7, 5)
    check(ops.colsum(cl(x1)).cpu(), x1.sum(dim=(2, 3, 4)), 1e-3, "colsum ragged")


@pytest.mark.parametrize("shape,groups", [((1, 256, 16, 16, 16), 32), ((2, 512, 16, 16, 16), 32), ((2, 256, 5, 6, 7), 32), ((3, 128, 1, 40, 40), 16),
                                          ((1, 512, 4, 4, 4), 32)])
@pytest.mark.parametrize("act", [0, 1])
def test_single_launch_groupnorm_equals_three_launches(ops, shape, groups, act, monkeypatch):
    """Small tensors can take statistics -> coefficients -> apply in ONE launch (mi_gn_small_fwd inside gn_apply, MI_GN_SMALL=1); the records
    it leaves for the backward and the activated tensor must be those of the three-launch path up to the summation order of the
    statistics (fp32 partials, fp64 finish in both)."""
    monkeypatch.setattr(ops, "GN_SMALL", True)
    n, c = shape[0], shape[1]
    v = shape[2] * shape[3] * shape[4]
    assert ops._lib.call_raw("mi_gn_small_supported", n, v, c, groups) == 1
    assert ops._lib.call_raw("mi_gn_small_supported", 1, 32 ** 3, 128, 32) == 0 and ops._lib.call_raw("mi_gn_small_supported", 1, 4096, 384, 32) == 0
    x = (rnd(*shape, scale=1.5) + 0.3).bfloat16().float()
    gamma, beta = (1 + 0.2 * rnd(c, seed=5)).to(dev), (0.1 * rnd(c, seed=6)).to(dev)
    xc = cl(x)
    st1 = ops.gn_stats(xc, groups, 1e-6, gamma, beta)
    assert st1._pending is not None
    y1 = ops.gn_apply(xc, st1, act)
    assert st1._pending is None
    st3 = ops.gn_stats(xc, groups, 1e-6, gamma, beta, allow_small=False)
    y3 = ops.gn_apply(xc, st3, act)
    assert torch.allclose(st1.mean_rstd, st3.mean_rstd, rtol=2e-6, atol=1e-7)
    assert torch.allclose(st1.scale_shift, st3.scale_shift, rtol=2e-6, atol=1e-6)
    assert float((y1 != y3).float().mean()) <= 2e-3, "more than rounding flips between the two forms"
    assert float((y1.float() - y3.float()).abs().max()) <= 2e-2 * float(y3.float().abs().max())
    # a reader of the records before any apply gets them from the plain statistics pass
    st2 = ops.gn_stats(xc, groups, 1e-6, gamma, beta)
    assert torch.equal(st2.scale_shift, st3.scale_shift) and st2._pending is None


@pytest.mark.parametrize("shape,groups", [((2, 32, 4, 6, 5), 32), ((1, 96, 8, 8, 8), 32), ((2, 64, 1, 16, 16), 16),
                                          ((1, 16, 5, 5, 5), 8), ((1, 512, 4, 4, 4), 32),
                                          # coarse-level shapes (8 / 16 channels per group, few voxels)
                                          ((2, 256, 5, 6, 7), 32), ((1, 256, 16, 16, 16), 32), ((1, 512, 16, 16, 16), 32)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_fwd_bwd(ops, shape, groups, silu):
    x = rnd(*shape, scale=1.5) + 0.3
    x = x.bfloat16().float()
    c = shape[1]
    gamma, beta = 1 + 0.2 * rnd(c, seed=5), 0.1 * rnd(c, seed=6)
    eps = 1e-6
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.group_norm(xr, groups, gr, br, eps)
    if silu:
        y = F.silu(y)
    g = rnd(*shape, seed=7)
    y.backward(g)
    xc = cl(x)
    st = ops.gn_stats(xc, groups, eps, gamma.to(dev), beta.to(dev))
    check(cf(ops.gn_apply(xc, st, silu)), y.detach(), 1e-2, "gn fwd")
    dgamma, dbeta = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    other, other2 = rnd(*shape, seed=8), rnd(*shape, seed=9)
    dx = ops.gn_bwd(cl(g), xc, st, gamma.to(dev), silu, dgamma, dbeta, add=cl(other), add2=cl(other2))
    check(cf(dx), xr.grad + other + other2, 1.5e-2, "gn dx")
    check(dgamma.cpu(), gr.grad, 1e-2, "gn dgamma")
    check(dbeta.cpu(), br.grad, 1e-2, "gn dbeta")
    # the two-launch form (fp64 atomic sums, coefficients derived inside the apply pass): the same dx bit for bit up to the summation
    # order of the block totals, the same parameter gradients
    dgamma2, dbeta2 = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    sums = torch.zeros(ops.GN_FUSED_REPLICAS * 2 * shape[0] * c, dtype=torch.float64, device=dev)  # replicated records of the atomic sums
    dx2 = ops.gn_bwd(cl(g), xc, st, gamma.to(dev), silu, dgamma2, dbeta2, add=cl(other), add2=cl(other2), sums=sums)
    check(cf(dx2), xr.grad + other + other2, 1.5e-2, "gn dx (fused)")
    assert float((dx2.float() - dx.float()).abs().max()) <= 2e-2 * float(dx.float().abs().max())
    assert float((dx2 != dx).float().mean()) <= 1e-3, "more than rounding flips between the two forms"
    check(dgamma2.cpu(), gr.grad, 1e-2, "gn dgamma (fused)")
    check(dbeta2.cpu(), br.grad, 1e-2, "gn dbeta (fused)")


@pytest.mark.parametrize("m,n,k,z", [(128, 128, 64, 1), (300, 200, 96, 2), (5, 512, 128, 1), (4096, 64, 4096, 2), (64, 70, 8, 3)])
def test_gemm_nt(ops, m, n, k, z):
    a, b = rnd(z, m, k), rnd(z, n, k, seed=1)
    bias, res = rnd(n, seed=2), rnd(z, m, n, seed=3)
    ref = 0.5 * a @ b.transpose(1, 2) + bias + res
    out = ops.gemm_nt(a.to(dev, torch.bfloat16), b.to(dev, torch.bfloat16), bias=bias.to(dev), res=res.to(dev, torch.bfloat16), alpha=0.5)
    check(out.float().cpu(), ref, 1e-2, "gemm bf16 out")
    acc = torch.ones(z, m, n, device=dev)
    ops.gemm_nt(a.to(dev, torch.bfloat16), b.to(dev, torch.bfloat16), out=acc, accumulate=True)
    check(acc.cpu(), 1 + a @ b.transpose(1, 2), 2e-3, "gemm f32 accumulate")
    # strided operands: column slices of a wider matrix (how q/k/v heads are addressed)
    wide = rnd(z, m, 3 * k, seed=4).to(dev, torch.bfloat16)
    out2 = ops.gemm_nt(wide[:, :, k:2 * k], b.to(dev, torch.bfloat16), out_f32=True)
    check(out2.cpu(), wide[:, :, k:2 * k].float().cpu() @ b.transpose(1, 2), 2e-3, "gemm strided A")


def test_transpose_softmax(ops):
    x = rnd(3, 70, 130).to(dev, torch.bfloat16)
    assert torch.equal(ops.transpose(x), x.transpose(1, 2).contiguous())
    s = rnd(2, 50, 777, scale=3.0)
    p = ops.softmax_fwd(s.to(dev))
    check(p.float().cpu(), torch.softmax(s, -1), 1e-2, "softmax")
    dp = rnd(2, 50, 777, seed=9)
    pr = p.float().cpu()
    ref = pr * (dp - (dp * pr).sum(-1, keepdim=True)) * 0.25
    check(ops.softmax_bwd(p, dp.to(dev), 0.25).float().cpu(), ref, 1e-2, "softmax bwd")


CONV_CASES = [
    # (N, Cin, Cout, dims, kernel, stride, padding)
    (1, 32, 32, (8, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 64, 96, (5, 9, 11), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 96, 32, (4, 8, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 1, 32, (8, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 32, 1, (8, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 1, 16, (5, 9, 11), (3, 3, 3), (1, 1, 1), (1, 1, 1)),    # single-channel streaming kernels (conv_c1.hip), ragged tiles, 2 images
    (2, 24, 1, (6, 7, 10), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 64, 1, (4, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 8, 8, (4, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 96, 64, (4, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 64, 32, (8, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0)),    # streaming 1x1 kernels: per-image dy column sums, 2 images
    (1, 192, 64, (4, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 32, 96, (4, 8, 9), (1, 1, 1), (1, 1, 1), (0, 0, 0)),    # ragged voxel count
    (1, 512, 512, (4, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),   # one tile = one split, 256 (cout block, cin chunk) pairs
    (1, 96, 544, (4, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),    # ... 17 cout blocks x 3 cin chunks
    (1, 40, 72, (4, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # ... ragged last chunk (8 of 32 channels) and last cout block (8 of 32 rows)
    (1, 512, 256, (4, 8, 8), (1, 1, 1), (1, 1, 1), (0, 0, 0)),  # weights too wide for the streaming kernel's LDS: forward / dgrad on the NT GEMM
    (2, 384, 128, (5, 9, 11), (1, 1, 1), (1, 1, 1), (0, 0, 0)),  # ... ragged voxel count, two images
    (1, 256, 768, (4, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 32, 32, (8, 8, 8), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 64, 64, (6, 10, 12), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 32, 32, (8, 8, 4), (3, 3, 1), (2, 2, 1), (1, 1, 0)),
    # k3 s2 p1 on all three axes = the phase kernels (convph.hip): odd extents (the last output reads a padded input voxel), 3 input
    # chunks, 32 / 64 / 128 output channels, 2 images, a tile grid with ragged tiles in every axis
    (2, 96, 64, (7, 9, 11), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 64, 32, (10, 18, 34), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 128, 128, (16, 16, 16), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (2, 32, 32, (24, 40, 36), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (2, 32, 64, (1, 16, 16), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (1, 64, 64, (1, 20, 12), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    (1, 16, 48, (4, 6, 6), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 256, 128, (4, 4, 4), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # 80 tiles x 2 channel blocks: the only case here large enough for the 64-channels-per-workgroup variants of the k3 s1 kernels
    # (forward AND data gradient) -- every smaller case runs the 32-channel variants
    (1, 128, 128, (20, 32, 32), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"n{c[0]}_{c[1]}to{c[2]}_{'x'.join(map(str, c[3]))}_k{c[4][0]}{c[4][2]}s{c[5][0]}{c[5][2]}")
def test_conv_fwd_dgrad_wgrad(ops, case):
    n, cin, cout, dims, k, s, p = case
    x = rnd(n, cin, *dims)
    w = rnd(cout, cin, *k, scale=1.0 / math.sqrt(cin * k[0] * k[1] * k[2]))
    bias = rnd(cout, seed=3)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv3d(xr, wr, bias, stride=s, padding=p)
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    plan = ops.ConvPlan(n, dims, cin, cout, k, s, p)
    assert plan.out_dims == tuple(y.shape[2:])
    plan.pack(w.to(dev))
    xc, gc = cl(x), cl(g)
    check(cf(plan.fwd(xc, addvec=bias.to(dev))), y.detach(), 1e-2, "conv fwd")
    check(cf(plan.dgrad(gc)), xr.grad, 1e-2, "conv dgrad")
    dw = torch.ones_like(w).to(dev)  # wgrad accumulates
    cs = torch.ones((n, cout), device=dev)
    plan.wgrad(xc, gc, dw, colsum=cs)
    check(dw.cpu() - 1, wr.grad, 1e-2, "conv wgrad")
    check(cs.cpu() - 1, g.sum(dim=(2, 3, 4)), 1e-2, "fused dy column sums (bias / temb gradient)")
    cb = torch.ones(cout, device=dev)  # 1-D target: summed over the batch as well (= the bias gradient)
    dw2 = torch.ones_like(w).to(dev)
    plan.wgrad(xc, gc, dw2, colsum=cb)
    check(cb.cpu() - 1, g.sum(dim=(0, 2, 3, 4)), 1e-2, "fused bias gradient")
    check(dw2.cpu() - 1, wr.grad, 1e-2, "conv wgrad (batch-summed bias gradient variant)")


UPCONV_CASES = [
    # (N, Cin, Cout, coarse dims): one / two / three / four input chunks (two chunks: the halo image of a tile serves all 8 phases),
    # 32 / 64 / 128 output channels, ragged coarse tile grids, 2 images
    (1, 64, 64, (8, 8, 8)),
    (2, 32, 32, (4, 8, 8)),
    (1, 64, 64, (5, 9, 11)),
    (2, 96, 64, (6, 10, 12)),
    (1, 128, 128, (8, 16, 16)),
    (1, 32, 64, (3, 5, 7)),
    (1, 64, 32, (12, 20, 24)),
    (1, 64, 64, (16, 32, 32)),   # 128 coarse tiles: several tiles per workgroup, the 64-channel variant
]


@pytest.mark.parametrize("case", UPCONV_CASES, ids=lambda c: f"n{c[0]}_{c[1]}to{c[2]}_{'x'.join(map(str, c[3]))}")
def test_upsample_conv_fwd_dgrad_wgrad(ops, case):
    """Upsample.forward (UNet:569-588) as one op -- nearest x2 + k3 s1 p1 conv by 8 phase convolutions on the coarse tensor -- against
    F.interpolate(mode="nearest") + F.conv3d autograd in fp32 on the bf16-rounded inputs.  The phase weights are sums of up to 8 taps
    rounded to bf16 once, the reference multiplies the taps one by one: both are within bf16 rounding of the exact result."""
    n, cin, cout, dims = case
    x = rnd(n, cin, *dims)
    w = rnd(cout, cin, 3, 3, 3, scale=1.0 / math.sqrt(cin * 27))
    bias = rnd(cout, seed=3)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv3d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, bias, padding=1)
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    plan = ops.UpConvPlan(n, dims, cin, cout)
    assert plan.out_dims == tuple(y.shape[2:])
    plan.pack(w.to(dev))
    xc, gc = cl(x), cl(g)
    check(cf(plan.fwd(xc, addvec=bias.to(dev))), y.detach(), 1e-2, "upsample + conv fwd")
    # into a channel-slice view of a wider buffer (how the U-Net's up path receives it: the first channels of a concat buffer)
    wide = torch.zeros((n,) + plan.out_dims + (cout + 32,), dtype=torch.bfloat16, device=dev)
    plan.fwd(xc, addvec=bias.to(dev), out=wide[..., :cout])
    check(cf(wide[..., :cout].contiguous()), y.detach(), 1e-2, "upsample + conv fwd into a concat slot")
    assert float(wide[..., cout:].abs().max()) == 0
    check(cf(plan.dgrad(gc)), xr.grad, 1e-2, "upsample + conv dgrad (coarse dx)")
    check(cf(plan.dgrad(wide_view(gc))), xr.grad, 1e-2, "upsample + conv dgrad from a strided dy view")
    dw = torch.ones_like(w).to(dev)  # wgrad accumulates
    cb = torch.ones(cout, device=dev)
    plan.wgrad(xc, gc, dw, colsum=cb)
    check(dw.cpu() - 1, wr.grad, 1e-2, "upsample + conv wgrad")
    check(cb.cpu() - 1, g.sum(dim=(0, 2, 3, 4)), 1e-2, "upsample + conv bias gradient")


def wide_view(t):
    """t [N, D, H, W, C] as the LAST C channels of a buffer with 32 more channels (a d(concat) slice)."""
    buf = torch.zeros(t.shape[:-1] + (t.shape[-1] + 32,), dtype=t.dtype, device=t.device)
    buf[..., 32:] = t
    return buf[..., 32:]


@pytest.mark.parametrize("n,cin,cout,dims,groups", [(2, 32, 32, (8, 8, 8), 32), (1, 64, 96, (5, 9, 11), 32), (1, 32, 64, (8, 16, 16), 16),
                                                    (3, 16, 48, (4, 6, 6), 8),
                                                    # >= 4 tiles along D: the rolling-halo variant (column segments, runs that end inside a column)
                                                    (1, 32, 32, (16, 16, 16), 32), (2, 32, 32, (22, 16, 24), 32), (3, 32, 32, (32, 8, 8), 8)])
def test_conv_emits_groupnorm_sums(ops, n, cin, cout, dims, groups):
    """The k3 s1 p1 forward kernel also emits per-channel sums of its (bf16) output; GroupNorm statistics built from them -- alone
    and as the first / second half of a channel concatenation -- must equal the statistics pass over the stored tensor."""
    x, w = rnd(n, cin, *dims), rnd(cout, cin, 3, 3, 3, scale=1.0 / math.sqrt(27 * cin))
    bias, res = rnd(cout, seed=3), cl(rnd(n, cout, *dims, seed=8))
    plan = ops.ConvPlan(n, dims, cin, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert plan.stats_chunks > 0
    plan.pack(w.to(dev))
    y, sums = plan.fwd(cl(x), addvec=bias.to(dev), res=res, want_sums=True)
    assert sums is not None and torch.equal(y, plan.fwd(cl(x), addvec=bias.to(dev), res=res))
    v = dims[0] * dims[1] * dims[2]
    gamma, beta = (1 + 0.2 * rnd(cout, seed=5)).to(dev), (0.1 * rnd(cout, seed=6)).to(dev)
    ref = ops.gn_stats(y, groups, 1e-6, gamma, beta)
    got = ops.gn_stats_from_sums(sums, None, n, v, groups, 1e-6, gamma, beta)
    check(got.scale_shift.cpu(), ref.scale_shift.cpu(), 1e-4, "scale/shift from conv sums")
    check(got.mean_rstd.cpu(), ref.mean_rstd.cpu(), 1e-4, "mean/rstd from conv sums")
    # concatenation [y | y2]: two sources with different chunk counts
    plan2 = ops.ConvPlan(n, dims, cin, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    plan2.pack(rnd(32, cin, 3, 3, 3, seed=11, scale=0.1).to(dev))
    y2, sums2 = plan2.fwd(cl(x), want_sums=True)
    cat = ops.concat_channels(y, y2)
    ct = cout + 32
    gam2, bet2 = torch.ones(ct, device=dev), torch.zeros(ct, device=dev)
    for g2 in sorted({g for g in (32, 16, 8, 4) if ct % g == 0}):  # includes groupings whose groups straddle the two halves
        ref2 = ops.gn_stats(cat, g2, 1e-6, gam2, bet2)
        got2 = ops.gn_stats_from_sums(sums, sums2, n, v, g2, 1e-6, gam2, bet2)
        check(got2.scale_shift.cpu(), ref2.scale_shift.cpu(), 1e-4, f"scale/shift from two sources, {g2} groups")


@pytest.mark.parametrize("n,cout,dims,groups", [(2, 32, (8, 8, 8), 32), (1, 32, (20, 16, 24), 8), (3, 16, (5, 9, 11), 4), (1, 24, (64, 64, 64), 8)])
def test_input_conv_emits_groupnorm_sums(ops, n, cout, dims, groups):
    """The 1 -> C forward kernel of the network's input conv (conv_c1.hip) emits the same per-channel sums: statistics built from them
    equal the statistics pass over the stored tensor, for several images, ragged tiles and more tiles than persistent workgroups."""
    x, w = rnd(n, 1, *dims), rnd(cout, 1, 3, 3, 3, scale=1.0 / math.sqrt(27))
    bias = rnd(cout, seed=3)
    plan = ops.ConvPlan(n, dims, 1, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert plan.stats_chunks > 0
    plan.pack(w.to(dev))
    y, sums = plan.fwd(cl(x), addvec=bias.to(dev), want_sums=True)
    assert sums is not None and torch.equal(y, plan.fwd(cl(x), addvec=bias.to(dev)))
    v = dims[0] * dims[1] * dims[2]
    gamma, beta = (1 + 0.2 * rnd(cout, seed=5)).to(dev), (0.1 * rnd(cout, seed=6)).to(dev)
    ref = ops.gn_stats(y, groups, 1e-6, gamma, beta)
    got = ops.gn_stats_from_sums(sums, None, n, v, groups, 1e-6, gamma, beta)
    check(got.scale_shift.cpu(), ref.scale_shift.cpu(), 1e-4, "scale/shift from the input conv's sums")
    check(got.mean_rstd.cpu(), ref.mean_rstd.cpu(), 1e-4, "mean/rstd from the input conv's sums")


@pytest.mark.parametrize("c", [32, 64])
def test_conv_residual_at_size(ops, c):
    """The residual pieces of the k3 s1 p1 kernel are loaded asynchronously a tap group ahead of their use; a load that is consumed
    (or whose register is copied) before it has landed only shows at sizes where HBM latency is real -- 8^3 cases pass by luck.
    2 x 64^3: 2048 tiles, both register-blocking variants (c = 32: one 32-channel block per wave -- the rolling-halo kernel --, c = 64:
    two).  The epilogue adds the residual to the bf16-rounded accumulator, so the result is bit-exact against conv-without-residual +
    residual.  (Adding it in fp32 inside the accumulator init of the 32-channel kernel was built and measured: 174 -> 145 us stand-alone,
    nothing in the step -- there the launch is HBM-bound on x + residual + y -- and one rounding instead of the reference's two; dropped.)"""
    n, d = 2, 64
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, d, d, d, c, generator=g).to(dev, torch.bfloat16)
    res = torch.randn(n, d, d, d, c, generator=g).to(dev, torch.bfloat16)
    w = (torch.randn(c, c, 3, 3, 3, generator=g) / math.sqrt(27 * c)).to(dev)
    bias = torch.randn(c, generator=g).to(dev)
    plan = ops.ConvPlan(n, (d, d, d), c, c, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    plan.pack(w)
    base = plan.fwd(x, addvec=bias)
    want = (base.float() + res.float()).to(torch.bfloat16)
    for with_sums in (False, True):
        r = plan.fwd(x, addvec=bias, res=res, want_sums=with_sums)
        y = r[0] if with_sums else r
        assert torch.isfinite(y.float()).all()
        assert torch.equal(y, want), f"residual epilogue differs (sums={with_sums}): {int((y != want).sum())} elements"


@pytest.mark.parametrize("c", [32, 64])
def test_conv27_at_size_vs_fp32_reference(ops, c):
    """The MFMA loops of k_conv27 read their fragments with inline-asm `ds_read_b128` consumed behind hand-counted `lgkmcnt` waits (the
    idiom class of the asynchronous-load bug fixed in round 1).  Evidence instead of argument: both register-blocking variants
    (c = 32: <1,.>, c = 64: <2,.>), forward (with and without the GroupNorm-sum epilogue) AND data gradient, against fp32
    F.conv3d on the CPU at 2 x 64^3 -- 2048 tiles per image pair, HBM latency real, every workgroup persistent over many tiles."""
    n, d = 2, 64
    g = torch.Generator().manual_seed(100 + c)
    x = torch.randn(n, c, d, d, d, generator=g).bfloat16().float()
    w = (torch.randn(c, c, 3, 3, 3, generator=g) / math.sqrt(27 * c))
    bias = torch.randn(c, generator=g)
    gy = torch.randn(n, c, d, d, d, generator=g).bfloat16().float()
    torch.set_num_threads(16)
    wb = w.bfloat16().float()  # the kernel multiplies bf16-rounded weights
    y_ref = F.conv3d(x, wb, bias, padding=1)
    dx_ref = F.conv_transpose3d(gy, wb, None, padding=1)  # data gradient of a stride-1 conv
    plan = ops.ConvPlan(n, (d, d, d), c, c, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    plan.pack(w.to(dev))
    xc, gc = cl(x), cl(gy)
    y0 = plan.fwd(xc, addvec=bias.to(dev))
    y1, sums = plan.fwd(xc, addvec=bias.to(dev), want_sums=True)
    assert torch.equal(y0, y1)
    # bf16 output rounding is the only error source: inputs and weights are exactly representable, accumulation is fp32
    check(cf(y0), y_ref, 4e-3, f"conv27 fwd c={c} @2x64^3")
    assert float((cf(y0) - y_ref).abs().max()) <= 2.0 ** -7 * float(y_ref.abs().max())  # every element within bf16 rounding of fp32
    dx = plan.dgrad(gc)
    check(cf(dx), dx_ref, 4e-3, f"conv27 dgrad c={c} @2x64^3")
    assert float((cf(dx) - dx_ref).abs().max()) <= 2.0 ** -7 * float(dx_ref.abs().max())
    if sums is not None:  # the emitted per-channel sums equal the sums of the stored tensor
        got = sums.partial.view(n, c, -1, 2).sum(2).cpu()
        yf = y0.float().cpu()
        check(got[..., 0], yf.sum(dim=(1, 2, 3)), 1e-3, "emitted channel sums")
        check(got[..., 1], (yf * yf).sum(dim=(1, 2, 3)), 1e-3, "emitted channel sums of squares")


def test_conv_fused_prologue_epilogue(ops):
    """GroupNorm-affine + SiLU prologue, per-sample add vector (bias + temb) and residual in the epilogue."""
    n, cin, cout, dims = 2, 64, 32, (4, 8, 8)
    x = rnd(n, cin, *dims, scale=1.3) + 0.2
    x = x.bfloat16().float()
    w = rnd(cout, cin, 3, 3, 3, scale=1 / math.sqrt(27 * cin))
    gamma, beta = 1 + 0.2 * rnd(cin, seed=5), 0.1 * rnd(cin, seed=6)
    addv, res = rnd(n, cout, seed=7), rnd(n, cout, *dims, seed=8)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    act = F.silu(F.group_norm(xr, 32, gamma, beta, 1e-6))
    act.retain_grad()
    y = F.conv3d(act, wr, None, padding=1) + addv[:, :, None, None, None] + res
    g = rnd(*y.shape, seed=9)
    y.backward(g)
    plan = ops.ConvPlan(n, dims, cin, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    plan.pack(w.to(dev))
    xc = cl(x)
    st = ops.gn_stats(xc, 32, 1e-6, gamma.to(dev), beta.to(dev))
    check(cf(plan.fwd(xc, st, True, addvec=addv.to(dev), res=cl(res))), y.detach(), 1e-2, "fused fwd")
    dw = torch.zeros_like(w).to(dev)
    plan.wgrad(xc, cl(g), dw, st, True)
    check(dw.cpu(), wr.grad, 1e-2, "fused wgrad")
    check(cf(plan.dgrad(cl(g))), act.grad, 1e-2, "dgrad wrt activated input")


def test_train_glue(ops):
    from medical_image_generation_amd._lib import call, ptr
    n, c, dims = 2, 4, (3, 4, 5)
    v = 60
    x0, noise = rnd(n, c, *dims), rnd(n, c, *dims, seed=1)
    acp = torch.cumprod(1 - torch.linspace(0.0015 ** 0.5, 0.0205 ** 0.5, 1000) ** 2, 0)
    t = torch.tensor([3, 900])
    out = torch.empty((n, *dims, c), dtype=torch.bfloat16, device=dev)
    sa, so = acp.sqrt().to(dev), (1 - acp).sqrt().to(dev)
    x0d, nd, td = x0.to(dev), noise.to(dev), t.to(dev)  # keep alive: calls are asynchronous
    call("mi_qsample", ptr(x0d), ptr(nd), ptr(sa), ptr(so), ptr(td), None, 0, ptr(out), None, n, c, v, 1000)
    ref = acp[t].sqrt().view(n, 1, 1, 1, 1) * x0 + (1 - acp[t]).sqrt().view(n, 1, 1, 1, 1) * noise
    check(cf(out), ref, 1e-2, "qsample")
    vel = torch.empty_like(x0d)  # v-prediction target (scheduler.get_velocity, T-LDM:163-165)
    call("mi_qsample", ptr(x0d), ptr(nd), ptr(sa), ptr(so), ptr(td), None, 0, ptr(out), ptr(vel), n, c, v, 1000)
    check(vel.cpu(), acp[t].sqrt().view(n, 1, 1, 1, 1) * noise - (1 - acp[t]).sqrt().view(n, 1, 1, 1, 1) * x0, 1e-6, "velocity target")
    # mode="concat": two un-noised condition channels behind the c noised ones (pitch c + 2), batch 2
    cond = rnd(n, 2, *dims, seed=5)
    condd = cond.to(dev)
    out2 = torch.empty((n, *dims, c + 2), dtype=torch.bfloat16, device=dev)
    call("mi_qsample", ptr(x0d), ptr(nd), ptr(sa), ptr(so), ptr(td), ptr(condd), 2, ptr(out2), None, n, c, v, 1000)
    check(cf(out2), torch.cat([ref, cond], dim=1), 1e-2, "qsample + concat condition")
    pred = rnd(n, c, *dims, seed=2)
    pr = pred.clone().requires_grad_(True)
    loss_ref = F.mse_loss(pr, noise)
    loss_ref.backward()
    loss, dpred = torch.zeros(1, device=dev), torch.empty_like(out)
    predd = cl(pred)
    call("mi_mse_fwd_bwd", ptr(predd), ptr(nd), ptr(dpred), ptr(loss), n, c, v, 1.0)
    check(loss.cpu(), loss_ref.detach().reshape(1), 1e-4, "mse loss")
    check(cf(dpred), pr.grad, 1e-2, "mse grad")


@pytest.mark.parametrize("decoupled,wd", [(1, 0.01), (0, 0.0), (0, 0.1)])
def test_adam_matches_torch(ops, decoupled, wd):
    from medical_image_generation_amd._lib import call, ptr
    n = 10007
    p0 = torch.randn(n)
    grads = [torch.randn(n) * (3.0 if i == 0 else 0.01) for i in range(3)]
    pr = p0.clone().requires_grad_(True)
    opt = (torch.optim.AdamW([pr], lr=1e-3, weight_decay=wd) if decoupled else torch.optim.Adam([pr], lr=1e-3, weight_decay=wd))
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    step, sumsq = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    for g in grads:
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = g.clone().to(dev)
        call("mi_sumsq_f32", ptr(gd), n, ptr(sumsq), 0)
        # the gradient buffer holds 4x the gradient and grad_scale = 1/4 undoes it (the data-parallel SUM all-reduce over 4 ranks)
        gd4 = gd * 4
        call("mi_sumsq_f32", ptr(gd4), n, ptr(sumsq), 0)
        call("mi_adam_step", ptr(p), ptr(gd4), ptr(m), ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, wd, decoupled, ptr(sumsq), 1.0, 0.25, ptr(step))
    assert float(step) == 3.0
    check(p.cpu(), pr.detach(), 1e-5, "adam params")


@pytest.mark.parametrize("rows,fin,fout", [(4096, 256, 768), (16384, 512, 1536), (1000, 768, 2304), (27, 64, 192), (4096, 96, 32), (515, 32, 32)])
def test_linear_weight_gradient_matches_fp32(rows, fin, fout):
    """mi_linear_wgrad_bf16 (the q/k/v Linear of AttentionBlock, UNet:379-381, and every 1x1 conv's weight gradient): dW += dy^T x,
    db += column sums of dy, on bf16 operands with fp32 accumulation -- over few or many voxel splits (the launch picks 4-32 of them;
    both ends of that range are hit by these shapes), ragged row counts and channel counts that are not multiples of 64."""
    from medical_image_generation_amd._lib import call, ptr
    x, dy = rnd(rows, fin, seed=1), rnd(rows, fout, seed=2)
    dw0, db0 = rnd(fout, fin, seed=3), rnd(fout, seed=4)
    dw, db = dw0.to(dev).contiguous(), db0.to(dev).contiguous()
    xb, dyb = x.to(dev).bfloat16().contiguous(), dy.to(dev).bfloat16().contiguous()  # (held: a temporary's block would be reused by the next one)
    call("mi_linear_wgrad_bf16", ptr(xb), fin, fin, ptr(dyb), fout, fout, rows, ptr(dw), ptr(db))
    torch.cuda.synchronize()
    want_w = dw0.double() + dy.double().t() @ x.double()
    want_b = db0.double() + dy.double().sum(0)
    check(dw.cpu().double(), want_w, 2e-3, "linear dW")
    check(db.cpu().double(), want_b, 2e-3, "linear db")


@pytest.mark.parametrize("rows,fin,fout", [(4100, 256, 512), (700, 64, 96)])
def test_linear_weight_gradient_on_channel_slices(rows, fin, fout):
    """The same entry point on operands that are channel slices of wider buffers (row pitch > features: the 1x1 shortcut conv of a
    ResnetBlock reads a slice of the skip-concat buffer), wide and narrow kernels."""
    from medical_image_generation_amd._lib import call, ptr
    ldx, ldy = fin + 64, fout + 128
    xw, dyw = rnd(rows, ldx, seed=5), rnd(rows, ldy, seed=6)
    xb, dyb = xw.to(dev).bfloat16().contiguous(), dyw.to(dev).bfloat16().contiguous()
    dw, db = torch.zeros(fout, fin, device=dev), torch.zeros(fout, device=dev)
    call("mi_linear_wgrad_bf16", xb.data_ptr() + 2 * 32, ldx, fin, dyb.data_ptr() + 2 * 64, ldy, fout, rows, ptr(dw), ptr(db))
    torch.cuda.synchronize()
    x, dy = xw[:, 32:32 + fin].double(), dyw[:, 64:64 + fout].double()
    check(dw.cpu().double(), dy.t() @ x, 2e-3, "linear dW (slices)")
    check(db.cpu().double(), dy.sum(0), 2e-3, "linear db (slices)")
