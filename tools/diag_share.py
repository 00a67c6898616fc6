"""Diagnostic: where does the time go when two ranks share one GPU over gloo (test hook of bench.py)?"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from gnm import synth
from gnm.parallel import DataParallelGIN
from gnm.graphs import CapturedTrainStep
from gnm.train import infomax_loss
from models.graphcnn import GIN_InfoMaxReg
B = 128
pool = synth.make_pool("dense_fc", B, first=rank * B, n=400, f0=7)
torch.manual_seed(0)
model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).train()
dp = DataParallelGIN(model)
arena = model.arena()
gids = np.array([arena.add(g) for g in pool], dtype=np.int64)
bt = arena.batch_from_gids(gids)
lab = torch.tensor([g.label for g in pool], device=dev)
loss_fn = lambda c, d, l: infomax_loss(c, d, l, 0.05)[0]
def t(f, n=10):
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def eager():
    c, d = model.forward_batch(bt); loss_fn(c, d, lab).backward()
eager(); 
print(rank, "eager ms", t(eager), flush=True)
cap = CapturedTrainStep(model, bt, loss_fn, zero_grad=dp.zero_grad)
print(rank, "replay ms", t(lambda: cap.run(bt, lab)), flush=True)
print(rank, "allreduce ms", t(lambda: dp.allreduce_gradients()), flush=True)
def both():
    cap.run(bt, lab); dp.allreduce_gradients()
print(rank, "replay+allreduce ms", t(both), flush=True)
x = torch.zeros(142288, device=dev)
print(rank, "allreduce(fresh tensor) ms", t(lambda: dist.all_reduce(x)), flush=True)
xc = torch.zeros(142288)
print(rank, "allreduce(cpu tensor) ms", t(lambda: dist.all_reduce(xc)), flush=True)
dist.destroy_process_group()
