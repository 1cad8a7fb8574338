"""fused vs two-kernel backward of the first hidden layer through the whole jet call (bf16 mode): where do the parameter
gradients differ?  GPU box: python tools/micro/dbg_fc1_fused_pipeline.py [npts]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from space_time_pde_amd import implicit_net, lig_jet

dev = torch.device("cuda:0")
lig_jet.set_mlp_precision("bf16")
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(3)
net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32, activation=torch.nn.Softplus).to(dev)
lat0 = 0.5 * torch.randn(1, 6, 9, 7, 32, device=dev)
pts = 0.02 + 0.96 * torch.rand(1, npts, 3, device=dev)
combo = {(0, 0): 0.7, (1, 1): 1.3}
cot = None
res = {}
for rep in range(2):
    for fused in ("1", "0"):
        os.environ["STPDE_FC1_FUSED"] = fused
        lat = lat0.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        jets, _ = lig_jet.lig_jets(net, lat, pts, 0., 1., True, (), combo=combo)
        if cot is None:
            cot = torch.randn_like(jets)
        (jets * cot).sum().backward()
        torch.cuda.synchronize()
        res[(rep, fused)] = [p.grad.clone() for p in net.parameters()]
names = [n for n, _ in net.named_parameters()]
for key in ((0, "1"), (1, "1")):
    print("run", key, "fused vs old (rep 0)")
    for n, a, b in zip(names, res[key], res[(0, "0")]):
        d = (a - b).abs()
        tol = 2e-5 * b.abs().max()
        bad = (d > tol).nonzero()
        if bad.numel():
            print("  %-12s %s: %d elements off, max %.4g; first: %s" % (n, tuple(a.shape), bad.shape[0], d.max().item(),
                  [(tuple(i.tolist()), round(a[tuple(i)].item(), 5), round(b[tuple(i)].item(), 5)) for i in bad[:6]]))
print("old vs old:", max((a - b).abs().max().item() for a, b in zip(res[(0, "0")], res[(1, "0")])))
