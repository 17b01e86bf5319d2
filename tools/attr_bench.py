"""Stand-alone timing of the attribute networks' kernels (csrc/attr.hip vs mlp.hip's gathered kernels) on a fixed synthetic work
list: 16,384 rays x 768 samples, a given fraction of them listed.  usage: [L4D_LIB=...] python tools/attr_bench.py [keep_fraction]"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from lidar4d_amd import ops, _lib

dev = "cuda"
keep = float(sys.argv[1]) if len(sys.argv) > 1 else 0.95
n_rays, T, n_geo, in_pad, nh, ls = 16384, 768, 15, 96, 2, 128.0
P = n_rays * T
g = torch.Generator(device=dev).manual_seed(0)
dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
denc = ops.freq_fwd(((dirs + 1) / 2).contiguous(), 12)
h = (torch.rand(P, 16, device=dev, generator=g) * 2 - 1).half()
z = torch.linspace(0.1, 1.0, T, device=dev).repeat(n_rays, 1).contiguous()
sigma = (torch.rand(n_rays, T, device=dev, generator=g) < keep).float() * 0.5 + 1e-9
_, _, _, idx_p, counts = ops.composite_fwd_padded(sigma, z, 0.9 / T, 1.0, False)
_, _, _, _, idx_c, count_c = ops.composite_fwd(sigma, z, 0.9 / T, 1.0, False, want_mask=False, want_idx=True)
print("rows (padded) %d, samples %d of %d" % (int(counts[0]), int(counts[1]), P))
wR = (torch.rand(64 * in_pad + 64 * 64 + 16 * 64, device=dev, generator=g) * 0.6 - 0.3).half()
wI = (torch.rand(64 * in_pad + 64 * 64 + 16 * 64, device=dev, generator=g) * 0.6 - 0.3).half()
a, c = torch.zeros(P, 2, device=dev), torch.zeros(P, 2, device=dev)
d_attr = torch.rand(P, 2, device=dev, generator=g) * 2 - 1
dh = torch.zeros(P, 16, dtype=torch.float16, device=dev)
gR, gI = torch.zeros(wR.numel(), device=dev), torch.zeros(wR.numel(), device=dev)


def run_new():
    rt = ops.attr_nets_fwd(idx_p, counts[0:1], P, n_rays, T, denc, h, n_geo, in_pad, wR, wI, nh, a, c)
    ops.attr_nets_bwd(idx_p, counts[0:1], P, n_rays, T, denc, h, n_geo, in_pad, wR, wI, nh, rt, d_attr, c, ls, dh, gR, gI, 1.0 / ls)


def run_old():
    for ch, w in ((0, wR), (1, wI)):
        ops.attr_mlp_fwd(idx_c, count_c, P, T, denc, h, n_geo, in_pad, w, nh, save_act=False, attr_dense=a, attr_compact=c, channel=ch)
    for ch, (w, gw) in enumerate(((wR, gR), (wI, gI))):
        ops.attr_mlp_bwd_gathered(idx_c, count_c, P, T, denc, h, n_geo, in_pad, None, None, w, nh, gw, 1.0 / ls, d_attr=d_attr,
                                  attr_compact=c, channel=ch, loss_scale=ls, dh16=dh, accumulate=(1 if ch == 1 else 0) | 2)


for name, fn in (("hoisted pair (attr.hip)", run_new), ("gathered (mlp.hip)", run_old)):
    fn()
    torch.cuda.synchronize()
    _lib.profile_start()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = {}
    for k, ms in _lib.profile_stop():
        tot[k] = tot.get(k, 0.0) + ms / 3
    print(name + ": " + "  ".join("%s %.3f" % (k.split("<")[0][:22], v) for k, v in tot.items() if v > 0.02) + "   sum %.3f ms" % sum(tot.values()))
