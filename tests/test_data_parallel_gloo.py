"""Data-parallel path on CPU with gloo, world_size 2 (the N > 1 path of bench.py uses the
same DataParallelGIN with backend nccl = RCCL).

The HIP kernels cannot run here, so each rank produces its local gradients with the CPU
oracle (tests may use it as a stand-in producer) and pushes them through the REAL product
code: FlatParams (parameters and grads re-homed into flat buffers), shard(), broadcast,
and the single flat all-reduce.  Checked:
  * rank r gets graphs [r*B/W, (r+1)*B/W) of the global batch -- a graph is never split;
  * after allreduce_gradients() every rank holds the MEAN of the per-rank gradients
    (bitwise identical across ranks), equal to the oracle's per-shard grads averaged;
  * parameters broadcast from rank 0 overwrite a diverged replica;
  * eps keeps a zero (not missing) slot in the flat buffer when learn_eps is False.
"""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, learn_eps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import edge_mat_of, load_case
        from gnm.parallel import DataParallelGIN
        from models.graphcnn import GIN_InfoMaxReg
        from oracle import gin_oracle as O

        cfg, state, d = load_case(case)
        torch.manual_seed(100 + rank)                        # replicas start DIFFERENT on purpose
        model = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, learn_eps, cfg["gpool"],
                               cfg["npool"], torch.device("cpu"))
        if rank == 0:
            model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        dp = DataParallelGIN(model)
        assert dp.world == world and dp.rank == rank
        assert dp.fp.total == sum(p.numel() for p in model.parameters())
        dp.broadcast_parameters(src=0)
        for k, v in model.state_dict().items():              # every replica now equals rank 0's
            assert np.array_equal(v.numpy(), np.asarray(state[k])), k

        graphs = [O.OGraph(cfg["n"], edge_mat_of(d[f"und_{g}"]), d[f"feat_{g}"], int(d["labels"][g]))
                  for g in range(cfg["B"])]
        mine = dp.shard(graphs)
        per = cfg["B"] // world
        assert [id(g) for g in mine] == [id(g) for g in graphs[rank * per:(rank + 1) * per]]

        # local gradients from the oracle (stand-in for the HIP backward on this CPU box)
        om = O.OracleGIN(state, cfg["L"], cfg["m"], learn_eps, cfg["gpool"], cfg["npool"], dtype=np.float64)
        perm = np.random.default_rng(7 + rank).permutation(len(mine))
        local = om.train_step_grads(mine, perm, update_running=False)["grads"]
        dp.zero_grad()
        for name, p in model.named_parameters():
            if name in local:
                p.grad.add_(torch.from_numpy(np.asarray(local[name], dtype=np.float32).reshape(p.shape)))
        for name, p in model.named_parameters():             # grads are views of the flat buffer
            assert p.grad.data_ptr() >= dp.fp.flat_grad.data_ptr()
        before = dp.fp.flat_grad.clone()
        dp.allreduce_gradients()
        assert dp.last_needs_scale is False                  # the synchronous form returns the mean
        # the asynchronous form has ONE contract: un-scaled sums + last_needs_scale, on every backend
        mean_sync = dp.fp.flat_grad.clone()
        dp.fp.flat_grad.copy_(before)
        work = dp.allreduce_gradients(async_op=True)
        assert work is not None and dp.last_needs_scale is True
        work.wait()
        assert torch.allclose(dp.fp.flat_grad / world, mean_sync, rtol=1e-6, atol=0)
        dp.fp.flat_grad.copy_(mean_sync)
        np.save(os.path.join(out_dir, f"flat_{rank}.npy"), dp.fp.flat_grad.numpy())
        np.save(os.path.join(out_dir, f"local_{rank}.npy"),
                np.concatenate([np.asarray(local.get(n, np.zeros(p.shape)), dtype=np.float64).reshape(-1)
                                for n, p in model.named_parameters()]))
        if not learn_eps:
            off = 0
            for name, p in model.named_parameters():
                if name == "eps":
                    assert torch.all(dp.fp.flat_grad[off:off + p.numel()] == 0)
                off += p.numel()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("learn_eps", [True, False])
def test_flat_allreduce_world2(tmp_path, learn_eps):
    case = "tiny_s1_eps1_gsum_nsum" if learn_eps else "tiny_s1_eps0_gsum_nsum"    # B = 4 -> 2 graphs per rank
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, learn_eps, str(tmp_path)), nprocs=world, join=True)
    flats = [np.load(tmp_path / f"flat_{r}.npy") for r in range(world)]
    locals_ = [np.load(tmp_path / f"local_{r}.npy") for r in range(world)]
    assert np.array_equal(flats[0], flats[1])                # bitwise identical on all ranks
    mean = (locals_[0] + locals_[1]) / world
    scale = np.abs(mean).max()
    assert np.abs(flats[0] - mean).max() <= 1e-6 * scale


def test_shard_requires_divisible_batch():
    from gnm.parallel import DataParallelGIN
    from models.graphcnn import GIN_InfoMaxReg
    m = GIN_InfoMaxReg(2, 2, 4, 32, 2, 0.0, True, "sum", "sum", torch.device("cpu"))
    dp = DataParallelGIN(m)                                   # no process group: world 1
    assert dp.shard(list(range(5))) == list(range(5))
    dp.world = 2
    with pytest.raises(ValueError):
        dp.shard(list(range(5)))


def test_flat_params_keep_module_semantics():
    """Re-homing parameters into the flat buffer must not change values, state_dict keys,
    or optimizer behaviour (Adam steps the views in place)."""
    from gnm.parallel import FlatParams
    from models.graphcnn import GIN_InfoMaxReg
    torch.manual_seed(3)
    m = GIN_InfoMaxReg(2, 2, 4, 32, 2, 0.0, True, "sum", "sum", torch.device("cpu"))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    opt = torch.optim.Adam(m.parameters(), lr=0.1)
    fp.zero_grad()
    for p in m.parameters():
        p.grad.add_(1.0)
    flat_before = fp.flat.clone()
    opt.step()
    assert not torch.equal(fp.flat, flat_before)              # the step wrote through the views
    off = 0
    for p in m.parameters():
        assert torch.equal(p.data.reshape(-1), fp.flat[off:off + p.numel()])
        off += p.numel()


def test_autograd_accumulates_into_flat_buffer():
    """loss.backward() must ADD into the flat gradient views (AccumulateGrad's in-place path), otherwise the
    all-reduce would see zeros.  No forward runs on this CPU box (the model has no CPU path): the loss is a function
    of the parameters themselves, which exercises the same AccumulateGrad nodes."""
    from gnm.parallel import FlatParams
    from models.graphcnn import GIN_InfoMaxReg
    from helpers import load_case

    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    torch.manual_seed(0)
    m = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, True, "sum", "sum", torch.device("cpu"))
    fp = FlatParams(m)
    for step in range(2):                       # second step: views must still be attached
        fp.zero_grad()
        sum((p * (k + 1.0)).square().sum() for k, p in enumerate(m.parameters())).backward()
        assert fp.flat_grad.abs().sum() > 0
        off = 0
        for p in m.parameters():
            assert p.grad.data_ptr() == fp.flat_grad.data_ptr() + 4 * off
            assert torch.equal(p.grad.reshape(-1), fp.flat_grad[off:off + p.numel()])
            off += p.numel()
