cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests/test_gpu_lig_jet.py -m gpu -q -k retain_graph 2>&1 | grep -E "passed|failed"; done
python - <<'PY'
import torch, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from space_time_pde_amd import lig_jet, implicit_net, nonlinearities
dev=torch.device("cuda:0")
g=torch.Generator().manual_seed(9)
lat=0.5*torch.randn(1,4,5,6,32,generator=g); pts=0.02+0.96*torch.rand(1,130,3,generator=g)
torch.manual_seed(0)
net=implicit_net.ImNet(dim=3,in_features=32,out_features=4,nf=16,activation=nonlinearities.NONLINEARITIES["softplus"]).to(dev)
latd=lat.to(dev).requires_grad_(True)
jets,_=lig_jet.lig_jets(net,latd,pts.to(dev),0.,1.,True,((1,1),(2,2)),chunk_points=64)
cot=torch.randn(jets.shape,generator=g).to(dev)
loss=(jets*cot).sum()
loss.backward(retain_graph=True)
w=[p.grad.clone() for p in net.parameters()]
loss.backward()
for k,(p,w1) in enumerate(zip(net.parameters(),w)):
    print(k, ((p.grad-2*w1).abs().max()/w1.abs().max()).item())
PY
