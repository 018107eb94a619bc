"""Debug aid: run one eager forward+backward of the C3a AutoencoderKL (batch 2, 128^3) and report the first op whose output has a NaN."""
import sys, torch
sys.path.insert(0, "/root/repo")
from medical_image_generation_amd import engine as E, hipops as ops
from medical_image_generation_amd.autoencoderkl import AutoencoderKL
from medical_image_generation_amd.trainer import AETrainer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
down = [[[1] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3], [[2] * 3, [3] * 3, [1] * 3]]
kw = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=8, num_res_blocks=2, num_channels=[32, 64, 128],
          attention_levels=[False] * 3, norm_num_groups=16, with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False,
          downsample_parameters=down, upsample_parameters=list(reversed(down))[:-1])
dev = torch.device("cuda")
torch.manual_seed(0)
net = AutoencoderKL(**kw).to(dev)
tr = AETrainer(net, lr=5e-5, kl_weight=1e-7)
first = [None]
def bad(t):
    return t is not None and t.is_floating_point() and not bool(torch.isfinite(t.float()).all())
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        r = orig(*a, **k)
        outs = r if isinstance(r, tuple) else (r,)
        for i, o in enumerate(outs):
            t = o if torch.is_tensor(o) else getattr(o, "scale_shift", None) if o is not None else None
            if t is None and o is not None and hasattr(o, "partial"): t = o.partial
            if torch.is_tensor(t) and bad(t) and first[0] is None:
                first[0] = (name, i, [tuple(x.shape) for x in a if torch.is_tensor(x)], [x for x in a if isinstance(x, str)])
                print("FIRST NON-FINITE:", first[0], flush=True)
        return r
    setattr(mod, name, f)
for n in ("gn_stats", "gn_stats_from_sums", "gn_apply", "gn_bwd", "add", "upsample_nearest", "upsample_nearest_bwd"):
    wrap(ops, n)
_of = ops.ConvPlan.fwd
def fwd_dbg(self, x, st=None, silu=False, addvec=None, res=None, out=None, want_sums=False):
    r = _of(self, x, st, silu, addvec=addvec, res=res, out=out, want_sums=want_sums)
    y = r[0] if isinstance(r, tuple) else r
    if first[0] is None and bad(y):
        first[0] = True
        print("FIRST NON-FINITE conv fwd:", self.cin, "->", self.cout, self.kernel, self.stride, "x", tuple(x.shape), "x finite", not bad(x),
              "addvec", None if addvec is None else (tuple(addvec.shape), not bad(addvec)), "res", None if res is None else (tuple(res.shape), not bad(res)),
              "out view", out is not None, "sums", isinstance(r, tuple) and r[1] is not None, flush=True)
        yy = y.float()
        nb = ~torch.isfinite(yy)
        print("  non-finite count", int(nb.sum()), "of", yy.numel(), "per image", nb.reshape(nb.shape[0], -1).sum(1).tolist(),
              "channels", nb.reshape(-1, nb.shape[-1]).any(0).nonzero().flatten().tolist()[:40], flush=True)
    return r
ops.ConvPlan.fwd = fwd_dbg
wrap(ops.ConvPlan, "dgrad")
x = torch.rand((B, 1, S, S, S), device=dev)
eps = torch.randn((B, 8, S // 4, S // 4, S // 4), device=dev)
for it in range(2):
    loss = tr.step(x, eps)
    torch.cuda.synchronize()
    print("step", it, "loss", float(loss), "grad finite", bool(torch.isfinite(tr.arena.grad).all()), "param finite", bool(torch.isfinite(tr.arena.data).all()), flush=True)
