import sys, time, json, torch
sys.path.insert(0, '.')
import bench
from space_time_pde_amd import implicit_net, lig_jet, nonlinearities, physics, unet3d
from space_time_pde_amd.train_step import sharded_step, GraphedStep
dev = torch.device("cuda:0")
out = {}
for name, igres, prec in (("configs[3]", (64, 256, 256), "bf16"), ("configs[1]", (32, 128, 128), "fp32")):
    torch.manual_seed(1)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32, activation=nonlinearities.NONLINEARITIES["softplus"]).to(dev)
    unet = unet3d.UNet3d(in_features=4, out_features=32, igres=igres, nf=16, mf=256).to(dev).train()
    layer = physics.get_rb2_pde_layer(**bench.RB2)
    crop, pts, tgt = bench.make_inputs(1 << 20, dev, igres=igres)
    lig_jet.set_mlp_precision(prec)
    params = list(unet.parameters()) + list(net.parameters())
    def eager():
        for p in params: p.grad = None
        return sharded_step(unet, net, layer, crop, pts, tgt, 1 << 20, 1.0, 0.0125, "l1")
    def timed(fn, n=8):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
    rec = dict(eager_ms=timed(eager))
    try:
        g = GraphedStep(unet, net, layer, crop, pts, tgt, 1 << 20, 1.0, 0.0125, "l1")
        rec["graph_ms"] = timed(lambda: g())
        rec["loss_graph"] = float(g()[0]); rec["loss_eager"] = float(eager()[0])
    except Exception as e:
        rec["graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    rec["peak_GB"] = torch.cuda.max_memory_allocated() / 2**30
    out[name] = rec
    del unet, net, crop, pts, tgt
    g = None
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
