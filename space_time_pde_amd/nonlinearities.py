"""Activations selectable by name (mirrors src/nonlinearities.py:5-22 of the reference)."""
import torch
import torch.nn as nn


class Swish(nn.Module):
    """x * sigmoid(beta * x) with a learnable scalar beta (reference nonlinearities.py:5-12)."""

    def __init__(self):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(1.0))

    def forward(self, x):
        return x * torch.sigmoid(self.beta * x)


NONLINEARITIES = {
    "tanh": nn.Tanh,
    "relu": nn.ReLU,
    "softplus": nn.Softplus,
    "elu": nn.ELU,
    "swish": Swish,
    "leakyrelu": nn.LeakyReLU,
}
