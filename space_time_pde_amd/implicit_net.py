"""IM-NET decoder (mirrors src/implicit_net.py:8-54 of the reference: same constructor, attributes, state_dict).

The module itself is ordinary PyTorch so that it can be trained, checkpointed and called on arbitrary
``[rows, dim+in_features]`` inputs; on the hot path (``query_local_implicit_grid`` + ``PDELayer`` on CUDA
tensors) its parameters are consumed directly by the HIP jet kernels and ``forward`` is never called.
"""
import torch
import torch.nn as nn


class ImNet(nn.Module):
    """MLP that re-concatenates its raw input after each of the first four layers."""

    def __init__(self, dim=3, in_features=32, out_features=4, nf=32, activation=torch.nn.LeakyReLU):
        super().__init__()
        self.dim = dim
        self.in_features = in_features
        self.dimz = dim + in_features
        self.out_features = out_features
        self.nf = nf
        self.activ = activation()
        widths = [nf * 16, nf * 8, nf * 4, nf * 2, nf]
        fan_in = [self.dimz] + [w + self.dimz for w in widths[:-1]]
        for k, (fi, fo) in enumerate(zip(fan_in, widths)):
            setattr(self, "fc%d" % k, nn.Linear(fi, fo))
        self.fc5 = nn.Linear(nf, out_features)
        # the reference registers every layer twice (fc0.. and fc.0..); checkpoints carry both key sets
        self.fc = nn.ModuleList([self.fc0, self.fc1, self.fc2, self.fc3, self.fc4, self.fc5])

    def forward(self, x):
        h = x
        for k in range(4):
            h = torch.cat([self.activ(self.fc[k](h)), x], dim=-1)
        return self.fc5(self.activ(self.fc4(h)))
