import sys, time, torch, copy
sys.path.insert(0, '/root/repo')
from space_time_pde_amd import unet3d
dev = torch.device('cuda:0')
torch.manual_seed(0)
igres = (32, 128, 128)
net = unet3d.UNet3d(in_features=4, out_features=32, igres=igres, nf=16, mf=256).to(dev).train()
net2 = copy.deepcopy(net)
x = torch.randn(1, 4, *igres, device=dev)
cot = torch.randn(1, 32, *igres, device=dev)

def run(m, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for p in m_params[id(m)]: p.grad = None
        y = m(x)
        y.backward(cot)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y

m_params = {id(net): list(net.parameters()), id(net2): list(net2.parameters())}
run(net, 2)
t_eager, y_e = run(net, 10)
g_e = [p.grad.clone() for p in net.parameters()]
print("eager ms/iter", t_eager)
net.deferred_weight_grads = True          # weight / bias gradients on a side stream, .grad assigned after the backward pass
run(net, 2)
t_def, y_d = run(net, 10)
g_d = [p.grad for p in net.parameters()]
print("eager, deferred weight gradients ms/iter", t_def)
print("  out rel diff", ((y_d - y_e).abs().max() / y_e.abs().max()).item(),
      " max grad-norm rel diff", max(((a - b).norm() / (b.norm() + 1e-20)).item() for a, b in zip(g_d, g_e)))
net.deferred_weight_grads = False
try:
    gnet = torch.cuda.make_graphed_callables(net2, (x,))
    m_params[id(gnet)] = list(net2.parameters())
    run(gnet, 2)
    t_g, y_g = run(gnet, 10)
    print("graphed ms/iter", t_g)
    # compare (BN running stats differ by number of calls; outputs in train mode only depend on batch stats)
    print("out rel diff", ((y_g - y_e).abs().max() / y_e.abs().max()).item())
    g_g = [p.grad for p in net2.parameters()]
    print("max grad-norm rel diff", max(((a - b).norm() / (b.norm() + 1e-20)).item() for a, b in zip(g_g, g_e)))
except Exception as e:
    import traceback; traceback.print_exc()
