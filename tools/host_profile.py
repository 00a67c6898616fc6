#!/usr/bin/env python3
"""cProfile of the host side of a training step at a tiny batch (GPU time negligible)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch, torch.nn.functional as F
from gnm import synth
from gnm.parallel import DataParallelGIN
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
pool = synth.make_pool("dense_fc", 32)
torch.manual_seed(0)
model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).train()
dp = DataParallelGIN(model)
arena = model.arena()
gids = np.array([arena.add(g) for g in pool], dtype=np.int64)
bt = arena.batch_from_gids(gids)
lab = torch.zeros(32, dtype=torch.int64, device=dev)
dl = torch.cat([torch.ones(bt.N, 1), torch.zeros(bt.N, 1)]).to(dev)
def step():
    dp.zero_grad()
    c, d = model.forward_batch(bt)
    loss = F.cross_entropy(c, lab) + 0.05 * F.binary_cross_entropy_with_logits(d, dl)
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
