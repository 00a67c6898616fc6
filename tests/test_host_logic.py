"""CPU-only checks of host logic added in round 2 (no device compute): the per-device launch-configuration
guard, KernelTimer bookkeeping when a fused entry point declines a shape, device selection for launches,
gradient-view re-attachment of the data-parallel wrapper under a stock torch optimizer, constructor shape limits."""
import numpy as np
import pytest
import torch

from helpers import load_case


def test_per_device_configure_once_guard():
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per (kernel, device): the guard every launcher keeps must
    fire once PER DEVICE, not once per process (a second GPU driven from the same process would otherwise launch
    its > 64 KB-LDS kernels unconfigured)."""
    from gnm._cabi import lib
    assert lib.gnm_debug_device_once(0, 1) == 1        # device 0, first use
    assert lib.gnm_debug_device_once(0, 0) == 0
    assert lib.gnm_debug_device_once(1, 0) == 1        # device 1 in the same process: configured on its own
    assert lib.gnm_debug_device_once(1, 0) == 0
    assert lib.gnm_debug_device_once(0, 0) == 0
    assert lib.gnm_debug_device_once(63, 0) == 1
    assert lib.gnm_debug_device_once(64, 0) == 1       # outside the table: never cached, always (re)configured
    assert lib.gnm_debug_device_once(64, 0) == 1
    assert lib.gnm_debug_device_once(-1, 0) == 1
    assert lib.gnm_debug_device_once(0, 1) == 1        # reset forgets


class _FakeEvent:
    clock = 0.0

    def record(self):
        self.t = _FakeEvent.clock

    def elapsed_time(self, other):
        return other.t - self.t


def test_kernel_timer_ignores_declined_fused_call(monkeypatch):
    """The unsupported-then-fallback sequence of encoder_forward: gnm_agg_fwd_bnrelu returns -2 without
    launching, then gnm_agg runs.  Only the launch that happened may be timed, with its own meta (round 1
    averaged an empty interval in and kept the fused label: a 2x inflated roofline for F != 64)."""
    from gnm import core
    monkeypatch.setattr(core, "_new_event", _FakeEvent)
    timer = core.KernelTimer(("agg_fwd_F128",))
    monkeypatch.setattr(core, "TIMER", timer)
    for _ in range(3):
        with core._timed("agg_fwd_F128", F=128, B=256, N=256000, fused_bnrelu=1) as tm:
            rc = -2                                       # declined: nothing launched
            if rc != 0:
                tm.cancel()
        with core._timed("agg_fwd_F128", F=128, B=256, N=256000):
            _FakeEvent.clock += 0.177                     # the real launch
        with core._timed("lin_fwd_K128_H128", N=1):       # not selected by the prefixes
            _FakeEvent.clock += 1.0
    summ = timer.summary()
    assert list(summ) == ["agg_fwd_F128"]
    c, ms, meta = summ["agg_fwd_F128"]
    assert c == 3 and abs(ms - 0.177) < 1e-9
    assert "fused_bnrelu" not in meta

    # a block that raises records nothing either
    with pytest.raises(RuntimeError):
        with core._timed("agg_fwd_F128", F=128, B=1, N=1):
            raise RuntimeError("launch failed")
    assert timer.summary()["agg_fwd_F128"][0] == 3

    # launches of the same tag but different kernels (meta) are reported apart, never averaged together
    with core._timed("agg_fwd_F128", F=128, B=256, N=256000, fused_bnrelu=1):
        _FakeEvent.clock += 0.5
    summ = timer.summary()
    assert set(summ) == {"agg_fwd_F128|", "agg_fwd_F128|fused_bnrelu=1"}
    assert summ["agg_fwd_F128|"][0] == 3 and abs(summ["agg_fwd_F128|fused_bnrelu=1"][1] - 0.5) < 1e-9


class _FakeTensor:
    def __init__(self, device):
        self.device = torch.device(device)
        self.is_cuda = self.device.type == "cuda"


def test_launch_device_selection():
    """Kernels are launched on the stream of the device the TENSORS live on (not torch's current device);
    CPU tensors and replicas spread over two devices are refused before any pointer reaches the library."""
    from gnm import core
    from gnm._cabi import GnmError
    assert core.launch_device(_FakeTensor("cuda:1"), None, _FakeTensor("cuda:1")) == torch.device("cuda:1")
    with pytest.raises(GnmError, match="GPU only"):
        core.launch_device(torch.zeros(1))
    with pytest.raises(GnmError, match="different devices"):
        core.launch_device(_FakeTensor("cuda:0"), _FakeTensor("cuda:1"))
    with pytest.raises(GnmError):
        core.launch_device(None)


def test_direct_grad_sink_survives_stock_optimizer_zero_grad():
    """DataParallelGIN in direct mode + torch.optim.Adam, the multi-GPU recipe of INTEGRATION.md:
    optimizer.zero_grad() (set_to_none=True by default) detaches the flat-buffer views; the sink backward hands
    autograd None, so without re-attachment Adam silently skips every parameter."""
    from gnm.parallel import DataParallelGIN
    from models.graphcnn import GIN_InfoMaxReg
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    torch.manual_seed(0)
    model = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, True, "sum", "sum",
                           torch.device("cpu"))
    dp = DataParallelGIN(model)
    assert dp.direct
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for how in ("dp.zero_grad", "allreduce_only", "module.zero_grad"):
        opt.zero_grad()                                   # grads -> None
        assert all(p.grad is None for p in model.parameters())
        if how == "module.zero_grad":
            model.zero_grad()                             # what compute_saliency does (graphcnn.py:256)
        if how != "allreduce_only":
            dp.zero_grad()
            off = 0
            for p in model.parameters():
                assert p.grad is not None and p.grad.data_ptr() == dp.fp.flat_grad.data_ptr() + 4 * off
                off += p.numel()
        # the HIP backward writes every gradient straight into the sink (= views of the flat buffer)
        for name, t in model._spec.grad_sink.items():
            t.fill_(0.25)
        dp.allreduce_gradients()                          # world 1: only re-attaches
        before = dp.fp.flat.clone()
        opt.step()
        moved = (dp.fp.flat - before).abs()
        assert float(moved.min()) > 0, how                # EVERY parameter took a step
    # non-direct mode: zero_grad zeroes and re-attaches
    model2 = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, True, "sum", "sum",
                            torch.device("cpu"))
    dp2 = DataParallelGIN(model2, direct_grads=False)
    for p in model2.parameters():
        p.grad = None
    dp2.fp.flat_grad.fill_(3.0)
    dp2.zero_grad()
    assert float(dp2.fp.flat_grad.abs().max()) == 0.0 and all(p.grad is not None for p in model2.parameters())


def test_constructor_rejects_shapes_outside_the_kernels():
    from gnm._cabi import lib
    from models.graphcnn import GIN_InfoMaxReg
    cpu = torch.device("cpu")
    assert lib.gnm_linear_max_k(64) == 448 and lib.gnm_linear_max_k(128) == 192 and lib.gnm_linear_max_k(129) == 0
    GIN_InfoMaxReg(5, 2, 400, 64, 2, 0.5, True, "sum", "sum", cpu)          # the reference's one_hot default fits
    for bad in (dict(hidden_dim=130), dict(hidden_dim=256), dict(hidden_dim=0), dict(num_layers=17),
                dict(input_dim=449), dict(input_dim=400, hidden_dim=128)):
        kw = dict(num_layers=5, input_dim=7, hidden_dim=64)
        kw.update(bad)
        with pytest.raises(ValueError):
            GIN_InfoMaxReg(kw["num_layers"], 2, kw["input_dim"], kw["hidden_dim"], 2, 0.5, True, "sum", "sum", cpu)
    # "max" neighbour pooling runs through the same Linear / BatchNorm kernels: same limits
    with pytest.raises(ValueError):
        GIN_InfoMaxReg(2, 2, 7, 130, 2, 0.5, True, "sum", "max", cpu)
    GIN_InfoMaxReg(2, 2, 7, 64, 2, 0.5, True, "sum", "max", cpu)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start its own ranks (round 1 exited with a usage
    message, so the driver's N > 1 form could never run).  --selftest-launch stops after the rendezvous: gloo,
    one all-reduce, rank 0's JSON line relayed through the parent -- no GPU involved."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0
    # a world size that contradicts --gpus is an error, not a silent single-rank run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                        capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_add_many_equals_add_and_symmetry_check():
    """One-upload bulk insertion gives the same arena as per-graph add(); the O(E) double-transpose symmetry
    test agrees with the definition (edge multiset of A == of A^T), duplicates included."""
    from gnm._cabi import lib
    from gnm.arena import GraphArena
    from gnm import synth
    gs = [synth.dense_fc_graph(i, n=40, t=32, f0=3) for i in range(40)]
    # two asymmetric graphs: a one-way edge, and a duplicated edge whose reverse is single
    e = gs[5].edge_mat.numpy().copy(); gs[5].edge_mat = torch.from_numpy(np.concatenate([e, [[0], [7]]], 1))
    e = gs[9].edge_mat.numpy().copy(); gs[9].edge_mat = torch.from_numpy(np.concatenate([e, e[:, :1]], 1))
    a1, a2 = GraphArena("cpu"), GraphArena("cpu")
    ids1 = [a1.add(g) for g in gs]
    for g in gs:
        g._gnm_cache = None
    ids2 = a2.add_many(gs, threads=4)
    assert ids1 == ids2 == list(range(40))
    assert a2.add_many(gs) == ids2                                   # cached: nothing is added twice
    assert len(a2) == 40
    for f in ("n", "nnz", "sym", "rp_off", "col_off", "t_rp_off", "t_col_off", "feat_off"):
        assert getattr(a1, f) == getattr(a2, f), f
    assert a1.sym.count(False) == 2 and not a1.sym[5] and not a1.sym[9]
    assert torch.equal(a1.rowptr.buf[:a1.rowptr.size], a2.rowptr.buf[:a2.rowptr.size])
    assert torch.equal(a1.col.buf[:a1.col.size], a2.col.buf[:a2.col.size])
    assert torch.equal(a1.feat.buf[:a1.feat.size], a2.feat.buf[:a2.feat.size])
    # brute-force definition on small random multigraphs
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(1, 7))
        A = rng.integers(0, 3, (n, n))
        if trial % 2:
            A = A + A.T
        src, dst = np.nonzero(A)
        em = np.stack([np.repeat(src, A[src, dst]), np.repeat(dst, A[src, dst])]).astype(np.int64)
        E = em.shape[1]
        rp = np.empty(n + 1, np.int32); col = np.empty(max(E, 1), np.uint16)
        assert lib.gnm_csr_from_edge_mat(np.ascontiguousarray(em).ctypes.data, E, n, rp.ctypes.data, col.ctypes.data) == 0
        assert bool(lib.gnm_csr_is_symmetric(rp.ctypes.data, col.ctypes.data, n)) == bool((A == A.T).all()), A



def test_choose_launch_mode_trial_logic():
    """bench.py's N > 1 launch-mode trial (gnm.parallel.choose_launch_mode): fastest wins, ties go to the earlier
    candidate, a candidate that raises (a captured collective failing at replay) or returns garbage is dropped,
    nothing measurable raises."""
    from gnm.parallel import choose_launch_mode
    t = {"graph+cc": 2.0e-3, "graph": 2.2e-3, "eager": 4.0e-3}
    calls = []

    def measure(name):
        calls.append(name)
        return t[name]
    assert choose_launch_mode(["graph+cc", "graph", "eager"], measure) == ("graph+cc", t)
    assert calls == ["graph+cc", "graph", "eager"]            # every candidate measured once, in order
    t2 = dict(t, eager=1.0e-3)
    assert choose_launch_mode(["graph+cc", "graph", "eager"], lambda n: t2[n])[0] == "eager"
    assert choose_launch_mode(["graph", "eager"], lambda n: 1.0)[0] == "graph"                     # tie -> preference order

    def flaky(name):
        if name == "graph+cc":
            raise RuntimeError("captured collective failed at replay")
        return float("nan") if name == "graph" else 3.0e-3
    mode, times = choose_launch_mode(["graph+cc", "graph", "eager"], flaky)
    assert mode == "eager" and list(times) == ["eager"]
    with pytest.raises(RuntimeError):
        choose_launch_mode(["graph"], lambda n: (_ for _ in ()).throw(ValueError("x")))
    # a candidate that failed on ANOTHER rank is dropped here as well (agree = MIN over the ranks of the success flag),
    # the exception text is kept, and a GPU fault is not swallowed
    errs = {}
    mode, times = choose_launch_mode(["graph+cc", "graph", "eager"], flaky, agree=lambda ok: ok, errors=errs)
    assert mode == "eager" and "captured collective failed" in errs["graph+cc"]
    veto = iter([False, True, True])
    mode, times = choose_launch_mode(["graph+cc", "graph", "eager"], lambda n: 1e-3, agree=lambda ok: ok and next(veto))
    assert mode == "graph" and "graph+cc" not in times
    with pytest.raises(RuntimeError, match="Memory access fault"):
        choose_launch_mode(["graph", "eager"], lambda n: (_ for _ in ()).throw(RuntimeError("Memory access fault by GPU")))


def test_allreduce_gradients_is_a_mean_and_a_noop_for_one_rank():
    """world size 1: no collective, the flat buffer is untouched (and .grad views are re-attached)."""
    import torch
    from gnm.parallel import DataParallelGIN
    lin = torch.nn.Linear(3, 2)
    dp = DataParallelGIN(lin, direct_grads=False)
    dp.zero_grad()
    lin.weight.grad.fill_(2.0)
    lin.weight.grad = None
    assert dp.allreduce_gradients() is None and dp.world == 1
    assert lin.weight.grad is not None and float(lin.weight.grad.sum()) == 12.0


def test_tile_to_wave_map_of_the_linear_kernels_is_a_bijection():
    """csrc/linear.hip lin_first_tile (balanced numbering of the waves that walk the 32-row tiles with a stride of all
    active waves): for every launch shape each index below 4 x rows belongs to exactly one (workgroup, wave) -- a tile
    nobody takes would be a block of rows left unwritten, a tile taken twice a data race -- and on full launches the first
    `nwg` indices fall on `nwg` different workgroups (what the numbering is for)."""
    from gnm._cabi import lib
    f = lib.gnm_debug_lin_first_tile
    for groups in (1, 2, 3):
        for rows in list(range(1, 14)) + [255, 256, 512, 767, 768]:
            nwg = (rows + groups - 1) // groups
            seen = {}
            for b in range(nwg):
                for wave in range(4 * groups):
                    t = f(wave, groups, rows, nwg, b)
                    active = (b * groups + wave // 4) < rows
                    if not active:
                        assert t >= 4 * rows, (groups, rows, b, wave, t)      # takes no tile
                        continue
                    assert 0 <= t < 4 * rows and t not in seen, (groups, rows, b, wave, t, seen.get(t))
                    seen[t] = (b, wave)
            assert len(seen) == 4 * rows
            if rows == groups * nwg and nwg > 1:
                assert len({seen[t][0] for t in range(nwg)}) == nwg
