import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from space_time_pde_amd import implicit_net, lig_jet, nonlinearities, physics, unet3d
from space_time_pde_amd.train_step import sharded_step
dev = torch.device('cuda:0')
torch.manual_seed(1)
net = implicit_net.ImNet(nf=32, activation=torch.nn.Softplus).to(dev)
unet = unet3d.UNet3d(in_features=4, out_features=32, igres=(32, 128, 128), nf=16, mf=256).to(dev); unet.train()
N = int(sys.argv[1])
crop, pts, tgt = bench.make_inputs(N, dev)
layer = physics.get_rb2_pde_layer(**bench.RB2)
lig_jet.DEFAULT_CHUNK = 1 << 18
def step():
    for p in list(net.parameters()) + list(unet.parameters()): p.grad = None
    return sharded_step(unet, net, layer, crop, pts, tgt, N, 1.0, 0.0125, "l1")
for _ in range(2): step()
torch.cuda.synchronize()
cpu, tot = [], []
for _ in range(4):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    cpu.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print('N', N, 'cpu enqueue ms', [round(c, 1) for c in cpu], 'total ms', [round(c, 1) for c in tot])
# UNet only
def unet_only():
    for p in unet.parameters(): p.grad = None
    y = unet(crop); y.sum().backward()
for _ in range(2): unet_only()
torch.cuda.synchronize()
t0 = time.perf_counter(); unet_only(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('unet fwd+bwd: cpu', round((t1 - t0) * 1e3, 1), 'total', round((t2 - t0) * 1e3, 1))
# step without the UNet (latent grid as leaf) -> UNet share of the step by difference
lat = unet(crop).detach().permute(0, 2, 3, 4, 1).contiguous()
from space_time_pde_amd import local_implicit_grid as lig
def step_nounet():
    for p in net.parameters(): p.grad = None
    l = lat.clone().requires_grad_(True)
    layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, l, q, 0., 1.))
    pred, res = layer(pts, return_residue=True)
    loss = (pred - tgt).abs().sum() / (N * 4) + 0.0125 * torch.stack(list(res.values()), 0).abs().sum() / (N * 4)
    loss.backward()
for _ in range(2): step_nounet()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): step_nounet()
torch.cuda.synchronize()
print('step without UNet ms', round((time.perf_counter() - t0) * 250, 1))
