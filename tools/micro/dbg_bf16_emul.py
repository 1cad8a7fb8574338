"""Debug aid: which bf16 emulation of the oracle matches the library's bf16 mode (per output stream)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from oracle import jet_ref as J
from space_time_pde_amd import lig_jet
import test_gpu_lig_jet as T

act = sys.argv[1] if len(sys.argv) > 1 else "softplus"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
pts = 0.02 + 0.96 * torch.rand(2, 150, 3, generator=g)
pairs = ((1, 1), (2, 2))
net = T._net(act, nf=32).to(dev)
jets, pp = lig_jet.lig_jets(net, lat.to(dev), pts.to(dev), 0., 1., True, pairs, chunk_points=128, precision="bf16")
jets = jets.double().cpu()
def emu(layers, tang):
    e = J.lig_jets(T._params64(net), act, lat.double(), pts.double(), 0., 1., second=tuple(pp), bf16_layers=layers,
                   bf16_pre_tangents=tang)
    return e.permute(0, 3, 1, 2).reshape(e.shape[0], 4, -1)
for layers, tang in [((1, 2), (1, 2)), ((1, 2, 3, 4, 5), (1, 2, 3, 4)), ((1, 2, 3, 4, 5), (1, 2)), ((1, 2), (1, 2, 3, 4)),
                     ((1, 2, 3), (1, 2, 3, 4)), ((1, 2), (1, 2)), ((), ())]:
    e = emu(layers, tang)
    print(layers, tang, ["%.2e" % T._normerr(jets[s], e[s]) for s in range(e.shape[0])])
kw = dict(second=tuple(pp), bf16_layers=(1, 2, 3, 4, 5), bf16_pre_tangents=(1, 2, 3, 4))
e64 = emu((1, 2, 3, 4, 5), (1, 2, 3, 4))
e32 = J.lig_jets([(w.float(), b.float()) for w, b in T._params64(net)], act, lat, pts, 0., 1., **kw)
e32 = e32.double().permute(0, 3, 1, 2).reshape(e32.shape[0], 4, -1)
print("emu32 vs emu64", ["%.2e" % T._normerr(e32[s], e64[s]) for s in range(e64.shape[0])], torch.get_num_threads())
print("kernel vs emu32", ["%.2e" % T._normerr(jets[s], e32[s]) for s in range(e64.shape[0])])
print(torch.__config__.show()[:600])
