"""Per-kernel timing of one training step at the bench's configs[1] shape, every launch (incl. the untimed
small kernels) through rocprofv3-free HIP events: python tools/time_small.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import synth
from gnm._cabi import lib, check
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
B = 1024
pool = synth.make_pool("dense_fc", 256)
torch.manual_seed(0)
m = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).train()
ar = m.arena()
gids = np.array([ar.add(g) for g in pool], dtype=np.int64)
bt = ar.batch_from_gids(gids[np.random.default_rng(0).integers(0, 256, B)])
N, L, H = bt.N, 5, 64
hs = [torch.randn(N, H, device=dev) for _ in range(L)]
U = torch.randn(B, L * H, device=dev)
perm = torch.randperm(B, device=dev).to(torch.int32)
bias = torch.zeros(1, device=dev)
d = torch.empty(2 * N, device=dev)
import ctypes as C
hp = (C.c_void_p * L)(*[h.data_ptr() for h in hs])
st = torch.cuda.current_stream().cuda_stream
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("disc_score_fwd us", t(lambda: check(lib.gnm_disc_score_fwd(hp, None, None, H, L, H, U.data_ptr(), L * H, perm.data_ptr(), bias.data_ptr(), bt.node_off.data_ptr(), N, B, d.data_ptr(), st), "f")))
dU = torch.empty_like(U); s2 = torch.empty(B, device=dev)
print("disc_score_bwd us", t(lambda: check(lib.gnm_disc_score_bwd(hp, None, None, H, L, H, d.data_ptr(), perm.data_ptr(), bt.node_off.data_ptr(), N, B, dU.data_ptr(), L * H, s2.data_ptr(), None, None, st), "b")))
