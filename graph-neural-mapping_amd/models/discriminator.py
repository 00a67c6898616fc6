"""Parameter container with the interface of the reference's Infomax scorer
(/root/reference models/discriminator.py:5-38): one nn.Bilinear(n_h, n_h, 1) called
`f_k`, xavier-uniform weight and zero bias (discriminator.py:13-17), so state_dict keys
are disc.f_k.weight / disc.f_k.bias and seeded construction matches.

Inside GIN_InfoMaxReg the scores come from the HIP row-dot kernels (csrc/disc.hip), which
read f_k.weight / f_k.bias.  forward() is the literal bilinear definition for
stand-alone use and the max-pooling fallback, both outside the accelerated path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Discriminator(nn.Module):
    def __init__(self, n_h):
        super().__init__()
        self.f_k = nn.Bilinear(n_h, n_h, 1)
        self.weights_init(self.f_k)

    def weights_init(self, m):
        if isinstance(m, nn.Bilinear):
            nn.init.xavier_uniform_(m.weight.data)
            if m.bias is not None:
                m.bias.data.zero_()

    def forward(self, c, h_pl, h_mi, s_bias1=None, s_bias2=None):
        # every graph summary is repeated N // B times (integer division, discriminator.py:24)
        c_x = torch.repeat_interleave(c, h_pl.shape[0] // c.shape[0], dim=0)
        scores = []
        for h, extra in ((h_pl, s_bias1), (h_mi, s_bias2)):
            sc = F.bilinear(h, c_x, self.f_k.weight, self.f_k.bias)
            scores.append(sc if extra is None else sc + extra)
        return torch.cat(scores, 0)
