"""Stage-by-stage comparison of the fused residual block (unet3d._ResBlockHip) with torch fp64 autograd."""
import sys, copy, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from space_time_pde_amd import unet3d

shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1, 32, 128, 128)
ci, cn, co = (int(v) for v in sys.argv[5:8]) if len(sys.argv) > 7 else (16, 16, 32)
dev = torch.device("cuda:0")
torch.manual_seed(1)
blk = unet3d.ResBlock3D(ci, cn, co).to(dev).train()
x = torch.randn(*shape, ci, device=dev) + 0.5
cot = torch.randn(*shape, co, device=dev)
ref = copy.deepcopy(blk).double()
xr = x.double().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
t = {}
def keep(name, v):
    v.retain_grad(); t[name] = v; return v
y1 = keep("y1", ref.conv1(xr)); h1 = keep("h1", torch.relu(ref.bn1(y1)))
y2 = keep("y2", ref.conv2(h1)); h2 = keep("h2", torch.relu(ref.bn2(y2)))
y3 = keep("y3", ref.conv3(h2)); sc = keep("sc", ref.shortcut(xr))
out = torch.relu(ref.bn3(y3) + sc)
(out.permute(0, 2, 3, 4, 1) * cot.double()).sum().backward()
cl = lambda v: v.permute(0, 2, 3, 4, 1)
unet3d._ResBlockHip.debug = dbg = {}
xx = x.clone().requires_grad_(True)
y = blk.forward_cl(xx)
(y * cot).sum().backward()
torch.cuda.synchronize()
def rel(a, b):
    return (a.double() - b).abs().max().item() / b.abs().max().item()
print("done flag", dbg["done"])
print("out", rel(y, cl(out)))
for k in ("y1", "h1", "y2", "y3"):
    print(k, rel(dbg[k], cl(t[k])))
print("dy3", rel(dbg["dy3"], cl(t["y3"].grad)))
print("dsc", rel(dbg["dsc"], cl(t["sc"].grad)))
print("dz2", rel(dbg["dz2"], cl(t["h2"].grad * (t["h2"] > 0))))
print("dy2", rel(dbg["dy2"], cl(t["y2"].grad)))
print("dz1", rel(dbg["dz1"], cl(t["h1"].grad * (t["h1"] > 0))))
print("dy1", rel(dbg["dy1"], cl(t["y1"].grad)))
print("dx", rel(dbg["dx"], cl(xr.grad)))
