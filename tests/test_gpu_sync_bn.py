"""sync_bn: two data-parallel ranks with cross-rank BatchNorm statistics reproduce ONE process
on the union batch -- logits, every parameter gradient after the flat all-reduce, BatchNorm running buffers.
Two processes share the one GPU of the test box and talk over gloo (RCCL needs one GPU per rank); everything
else is the product path: DataParallelGIN(sync_bn=True), the HIP kernels, the gradient sink, the flat all-reduce.
The loss is the cross entropy only: the Infomax negatives index rows of the LOCAL batch (graphcnn.py:198-201,242),
which no sharding can reproduce; default (local) BatchNorm is checked to differ, so the test has teeth."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graph-neural-mapping_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _graphs(cfg, d):
    from helpers import edge_mat_of

    class G:
        pass
    out = []
    for g in range(cfg["B"]):
        o = G()
        o.g = list(range(cfg["n"]))
        o.edge_mat = torch.from_numpy(edge_mat_of(d[f"und_{g}"]))
        o.node_features = torch.from_numpy(d[f"feat_{g}"])
        o.label = int(d["labels"][g])
        out.append(o)
    return out


def _model(cfg, state):
    from models.graphcnn import GIN_InfoMaxReg
    dev = torch.device(DEV)
    m = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, cfg["learn_eps"], cfg["gpool"],
                       cfg["npool"], dev)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return m.to(dev).train()


def _step(model, dp, graphs):
    dp.zero_grad()
    np.random.seed(11)
    c_logit, _ = model(graphs)
    labels = torch.tensor([g.label for g in graphs], device=DEV)
    torch.nn.functional.cross_entropy(c_logit, labels).backward()
    dp.allreduce_gradients()
    return c_logit


def _worker(rank, world, port, case, sync_bn, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import load_case
        from gnm.parallel import DataParallelGIN
        cfg, state, d = load_case(case)
        model = _model(cfg, state)
        dp = DataParallelGIN(model, sync_bn=sync_bn)
        mine = dp.shard(_graphs(cfg, d))
        for _ in range(2):                                   # two steps: running buffers accumulate
            c_logit = _step(model, dp, mine)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), c_logit=c_logit.detach().cpu().numpy(),
                 flat_grad=dp.fp.flat_grad.cpu().numpy(),
                 **{"buf_" + k: v.cpu().numpy() for k, v in model.named_buffers()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["tiny_s1_eps1_gsum_nsum", "tiny_s1_eps0_gaverage_naverage", "tiny_s3_mlp3"])
def test_two_ranks_with_sync_bn_equal_one_process_on_the_union_batch(case, tmp_path):
    from helpers import load_case, rel_err
    from gnm.parallel import DataParallelGIN
    cfg, state, d = load_case(case)
    assert cfg["B"] % 2 == 0
    # single process, union batch
    model = _model(cfg, state)
    dp = DataParallelGIN(model)
    graphs = _graphs(cfg, d)
    for _ in range(2):
        c_ref = _step(model, dp, graphs)
    ref_flat = dp.fp.flat_grad.cpu().numpy()
    ref_buf = {k: v.cpu().numpy() for k, v in model.named_buffers()}
    c_ref = c_ref.detach().cpu().numpy()
    scale = float(np.abs(ref_flat).max())
    results = {}
    for sync_bn in (True, False):
        out = tmp_path / ("sync" if sync_bn else "local")
        out.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), case, sync_bn, str(out)), nprocs=2, join=True)
        r = [dict(np.load(out / f"rank{k}.npz")) for k in range(2)]
        assert np.array_equal(r[0]["flat_grad"], r[1]["flat_grad"])          # the all-reduce left both ranks equal
        results[sync_bn] = r
    r = results[True]
    c = np.concatenate([r[0]["c_logit"], r[1]["c_logit"]], 0)
    assert rel_err(c, c_ref) <= 1e-5
    assert np.max(np.abs(r[0]["flat_grad"] - ref_flat)) <= 5e-5 * scale
    for k, v in ref_buf.items():
        if k.endswith("num_batches_tracked"):
            assert int(r[0]["buf_" + k]) == int(v) == 2
        else:
            assert rel_err(r[0]["buf_" + k], v) <= 1e-5, k
            assert np.array_equal(r[0]["buf_" + k], r[1]["buf_" + k]), k
    # without sync_bn the shards normalise with their own statistics: visibly different numbers
    rl = results[False]
    cl = np.concatenate([rl[0]["c_logit"], rl[1]["c_logit"]], 0)
    assert rel_err(cl, c_ref) > 1e-3


def _fused_worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import load_case
        from gnm.train import FusedTrainStep
        cfg, state, d = load_case(case)
        torch.manual_seed(50 + rank)                      # replicas start different: the step object must sync them
        model = _model(cfg, state)
        if rank == 1:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.5)
        step = FusedTrainStep(model, lr=0.01, capture=False)
        graphs = _graphs(cfg, d)
        mine = step.dp.shard(graphs)
        arena = model.arena()
        batch = arena.batch_from_gids(np.array([arena.add(g) for g in mine], dtype=np.int64))
        labels = torch.tensor([g.label for g in mine], device=DEV)
        for s in range(3):
            step.run(batch, labels, np.arange(len(mine)))
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=step.dp.fp.flat.cpu().numpy(),
                 grad=step.dp.fp.flat_grad.cpu().numpy(), steps=step.optimizer.step_count.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_fused_train_step_keeps_two_ranks_in_lockstep(tmp_path):
    """FusedTrainStep over a process group: parameters broadcast from rank 0 at construction, gradients SUM-all-reduced
    and scaled by 1/world inside the Adam launch -> after three steps both ranks hold bitwise the same parameters."""
    case = "tiny_s1_eps1_gsum_nsum"
    mp.spawn(_fused_worker, args=(2, _free_port(), case, str(tmp_path)), nprocs=2, join=True)
    r = [dict(np.load(tmp_path / f"rank{k}.npz")) for k in range(2)]
    assert np.array_equal(r[0]["flat"], r[1]["flat"]) and np.array_equal(r[0]["grad"], r[1]["grad"])
    assert int(r[0]["steps"][0]) == 3 and np.isfinite(r[0]["flat"]).all()
    from helpers import load_case
    _, state, _ = load_case(case)
    moved = np.abs(r[0]["flat"]).sum() != 0 and not np.array_equal(
        r[0]["flat"][:3], np.asarray(state["eps"], dtype=np.float32)[:3])
    assert moved                                           # the optimizer did move the parameters
