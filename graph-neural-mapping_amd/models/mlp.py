"""Parameter container with the interface of the reference's per-layer update MLP
(/root/reference models/mlp.py:6-49).

Contract kept: MLP(num_layers, input_dim, hidden_dim, output_dim); ValueError for
num_layers < 1 (mlp.py:21-22); submodule names and creation order -- `linear` for a
single layer (mlp.py:25), else `linears` then `batch_norms` (mlp.py:32-38) -- so a seeded
construction consumes the RNG exactly as the reference does and state_dict keys match.

Inside GIN_InfoMaxReg the hot path never calls forward(): the HIP kernels read these
tensors directly (gnm/core.py).  forward() is the plain definition for stand-alone use
and for the max-pooling fallback, both outside the accelerated path.
"""
import torch.nn as nn
import torch.nn.functional as F


class MLP(nn.Module):
    def __init__(self, num_layers, input_dim, hidden_dim, output_dim):
        super().__init__()
        if num_layers < 1:
            raise ValueError("number of layers should be positive!")
        self.num_layers = num_layers
        self.linear_or_not = num_layers == 1
        if self.linear_or_not:
            self.linear = nn.Linear(input_dim, output_dim)
            return
        widths = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.linears = nn.ModuleList(nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))
        self.batch_norms = nn.ModuleList(nn.BatchNorm1d(hidden_dim) for _ in range(num_layers - 1))

    def forward(self, x):
        if self.linear_or_not:
            return self.linear(x)
        for lin, bn in zip(self.linears[:-1], self.batch_norms):
            x = F.relu(bn(lin(x)))
        return self.linears[-1](x)
