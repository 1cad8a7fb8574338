"""Build-container only: time the IMPORTED reference and the CPU restatement (oracle/cpu_ref.py) on the same chunks of
BASELINE configs[1] (latent [1,32,128,128,32], 4096-point chunks, softplus and leaky-relu, full RB2 set, backward to the
IM-NET parameters and the latent grid) to establish the restatement/reference ratio that bench.py's ``cpu_baseline``
("port") relies on.  Writes profiles/r2_cpu_ref_vs_port.json.

    python tools/ref_vs_port_timing.py [nchunks]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
np.int = int
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "experiments", "rb2d"))
import implicit_net as r_imnet  # noqa: E402
import local_implicit_grid as r_lig  # noqa: E402
import nonlinearities as r_nl  # noqa: E402
import physics as r_phys  # noqa: E402

from oracle import cpu_ref  # noqa: E402

MEAN, STD = (0.01, 0.0, 0.02, -0.01), (0.05, 0.3, 0.15, 0.12)
RB2 = dict(mean=MEAN, std=STD, t_crop=2., z_crop=1., x_crop=1., use_continuity=True)


def main():
    nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(0)
    latent = 0.5 * torch.randn(1, 32, 128, 128, 32, generator=g)
    out = {"threads": torch.get_num_threads(), "chunk": 4096, "nchunks": nchunks, "host": os.uname().nodename}
    for act in ("softplus", "leakyrelu"):
        params = cpu_ref.imnet_init(nf=32, seed=1)
        net = r_imnet.ImNet(dim=3, in_features=32, out_features=4, nf=32, activation=r_nl.NONLINEARITIES[act])
        with torch.no_grad():
            for k, (w, b) in enumerate(params):
                net.fc[k].weight.copy_(w)
                net.fc[k].bias.copy_(b)
        layer = r_phys.get_rb2_pde_layer(**RB2)
        pde = cpu_ref.rb2_oracle(**RB2)
        t_ref, t_port = [], []
        for c in range(nchunks + 1):          # chunk 0 = warm-up for both
            pts = torch.rand(1, 4096, 3, generator=g)
            tgt = torch.randn(1, 4096, 4, generator=g)
            lat = latent.clone().requires_grad_(True)
            t0 = time.perf_counter()
            layer.update_forward_method(lambda p: r_lig.query_local_implicit_grid(net, lat, p, 0., 1.))
            pred, res = layer(pts.clone(), return_residue=True)
            reg = torch.nn.functional.l1_loss(pred, tgt)
            st = torch.stack(list(res.values()), 0)
            (reg + 0.0125 * torch.nn.functional.l1_loss(st, torch.zeros_like(st))).backward()
            t1 = time.perf_counter()
            cpu_ref.lig_pde_step(params, act, latent, pts, tgt, pde, 1.0, 0.0125)
            t2 = time.perf_counter()
            net.zero_grad()
            if c:
                t_ref.append(t1 - t0)
                t_port.append(t2 - t1)
        mr, mp = float(np.median(t_ref)), float(np.median(t_port))
        out[act] = dict(reference_s_per_chunk=mr, port_s_per_chunk=mp, reference_pts_per_s=4096 / mr,
                        port_pts_per_s=4096 / mp, port_over_reference_time=mp / mr)
        print(act, out[act], flush=True)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r2_cpu_ref_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
