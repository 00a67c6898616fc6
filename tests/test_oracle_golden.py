"""Pin the CPU oracle (oracle/gin_oracle.py) against golden vectors captured from
the real reference (tests/golden/make_goldens.py).  CPU only.

Integer structures: bit-exact.  Floats: <= 1e-5 relative (max-norm) for the
float32 oracle; the float64 oracle must also sit within 1e-5 of the reference's
float32 outputs (it is the build's only fp64 cross-check)."""
import numpy as np
import pytest

from helpers import (ORACLE32_FACTOR, ORACLE32_GRAD_RTOL, RTOL, TRUE_SHAPE_GRAD_RTOL, Calibrated, assert_close, assert_grad_true_shape,
                     edge_mat_of, golden_cases, grad_floor,
                     load_case, neighbors_of)
from oracle import gin_oracle as O

CASES = golden_cases()


def make_batch(cfg, d):
    return [O.OGraph(cfg["n"], edge_mat_of(d[f"und_{g}"]), d[f"feat_{g}"], int(d["labels"][g]),
                     neighbors=neighbors_of(d[f"und_{g}"], cfg["n"])) for g in range(cfg["B"])]


def make_model(cfg, state, dtype):
    return O.OracleGIN(state, cfg["L"], cfg["m"], cfg["learn_eps"], cfg["gpool"], cfg["npool"], dtype=dtype)


def test_golden_inventory():
    assert len(CASES) >= 15
    assert any(c.startswith("true_") for c in CASES)


@pytest.mark.parametrize("case", CASES)
def test_integer_structures_bit_exact(case):
    cfg, state, d = load_case(case)
    batch = make_batch(cfg, d)
    if cfg["npool"] == "max":
        padded = O.build_padded_neighbors(batch, cfg["learn_eps"])
        assert padded.dtype == np.int64 and np.array_equal(padded, d["padded_neighbors"])
    else:
        idx, val, shape = O.build_adj_block(batch, cfg["learn_eps"])
        assert idx.dtype == np.int64
        assert np.array_equal(idx, d["adj_indices"])          # same order, same values
        assert np.array_equal(val, d["adj_values"])
    gidx, gval, gshape = O.build_graph_pool(batch, cfg["gpool"])
    assert np.array_equal(gidx, d["gp_indices"])
    assert np.array_equal(gval, d["gp_values"])            # fp32 1/n bit pattern
    assert np.array_equal(O.shuffle_index(batch, d["perm"]), d["shuffle_idx"])
    # the numpy global RNG draw the module must reproduce (graphcnn.py:199)
    np.random.seed(cfg["np_seed"])
    assert np.array_equal(np.random.permutation(cfg["B"]), d["perm"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", CASES)
def test_eval_forward(case, dtype):
    cfg, state, d = load_case(case)
    batch = make_batch(cfg, d)
    model = make_model(cfg, state, dtype)
    c_logit, d_logit, cache = model.forward(batch, d["perm"], training=False)
    rs = slice(None, None, cfg["row_stride"])
    for l in range(cfg["L"]):
        assert_close(cache["layers"][l]["pooled"][rs], d[f"eval_pooled_{l}"], what=f"pooled {l}")
        assert_close(cache["hidden"][l][rs], d[f"eval_hidden_{l}"], what=f"hidden {l}")
    assert_close(c_logit, d["eval_c_logit"], what="c_logit")
    assert_close(d_logit, d["eval_d_logit"], what="d_logit")
    assert_close(model.latent(batch, d["perm"]), d["eval_latent"], what="latent")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", CASES)
def test_train_step(case, dtype):
    cfg, state, d = load_case(case)
    batch = make_batch(cfg, d)
    model = make_model(cfg, state, dtype)
    out = model.train_step_grads(batch, d["perm"], beta=0.05)
    rs = slice(None, None, cfg["row_stride"])
    cache = out["cache"]
    if case.startswith("true_"):
        # deep true-shape activations: calibrated against the fp64 oracle (helpers.TRUE_SHAPE_FACTOR)
        truth_model = make_model(cfg, state, np.float64)
        truth_out = truth_model.train_step_grads(batch, d["perm"], beta=0.05)
        truth = truth_out["cache"]
        cal = Calibrated(factor=ORACLE32_FACTOR if dtype == np.float32 else 1.0)
        for l in range(cfg["L"]):
            cal.check(cache["layers"][l]["pooled"][rs], d[f"train_pooled_{l}"],
                      truth["layers"][l]["pooled"][rs], what=f"pooled {l}")
            cal.check(cache["hidden"][l][rs], d[f"train_hidden_{l}"], truth["hidden"][l][rs], what=f"hidden {l}")
    else:
        for l in range(cfg["L"]):
            assert_close(cache["layers"][l]["pooled"][rs], d[f"train_pooled_{l}"], what=f"pooled {l}")
            assert_close(cache["hidden"][l][rs], d[f"train_hidden_{l}"], what=f"hidden {l}")
    losses = np.array([out["loss"], out["c_loss"], out["d_loss"]])
    if case.startswith("true_"):
        cal.check(out["c_logit"], d["train_c_logit"], truth_out["c_logit"], what="c_logit")
        cal.check(out["d_logit"], d["train_d_logit"], truth_out["d_logit"], what="d_logit")
        cal.check(losses, d["train_loss"],
                  np.array([truth_out["loss"], truth_out["c_loss"], truth_out["d_loss"]]), what="loss")
    else:
        assert_close(out["c_logit"], d["train_c_logit"], what="c_logit")
        assert_close(out["d_logit"], d["train_d_logit"], what="d_logit")
        assert_close(losses, d["train_loss"], what="loss")
    g = out["grads"]
    checked = 0
    # Gradients accumulate fp32 rounding of the reference's own backward through
    # up to 5 layers x 2 BatchNorms, so they get 5x the activation tolerance.
    gtol = 5 * RTOL
    floor = grad_floor(d)
    for key in d:
        if key.startswith("grad_"):
            name = key[len("grad_"):]
            if case.startswith("true_"):
                assert_grad_true_shape(g[name].reshape(d[key].shape), d[key],
                                       truth_out["grads"][name].reshape(d[key].shape), what=key, floor=floor,
                                       rtol=ORACLE32_GRAD_RTOL if dtype == np.float32 else TRUE_SHAPE_GRAD_RTOL)
            else:
                assert_close(g[name].reshape(d[key].shape), d[key], rtol=gtol, what=key, floor=floor)
            checked += 1
        if key.startswith("gradnone_"):
            assert key[len("gradnone_"):] not in g or not cfg["learn_eps"]
    if "gradproj_r" in d:
        if case.startswith("true_"):
            # (random projections of the [L H, L H] discriminator gradient cancel heavily: the REFERENCE's own fp32
            #  value sits 2.7e-3 from the fp64 oracle on the configs[3]-shaped case, so this anchor keeps 5e-3)
            gtol = ORACLE32_GRAD_RTOL
        W = g["disc.f_k.weight"][0].astype(np.float64)
        assert_close(W @ d["gradproj_r"], d["gradproj_Wr"], rtol=gtol, what="disc dW r")
        assert_close(d["gradproj_s"] @ W, d["gradproj_sW"], rtol=gtol, what="disc s dW")
        assert_close(np.array(np.linalg.norm(W)), d["gradproj_fro"], rtol=gtol, what="disc |dW|")
    assert checked >= 10
    # eps has no gradient when learn_eps is False (graph never touches it)
    if not cfg["learn_eps"]:
        assert "gradnone_eps" in d and "eps" not in g
    # BN running statistics after the step
    for key in d:
        if key.startswith("bufafter_"):
            name = key[len("bufafter_"):]
            if name.endswith("num_batches_tracked"):
                assert int(model.p[name]) == int(d[key])
            elif case.startswith("true_"):
                cal.check(model.p[name], d[key], truth_model.p[name], what=key)
            else:
                assert_close(model.p[name], d[key], what=key)


@pytest.mark.parametrize("case", [c for c in CASES if c.startswith("tiny_")])
def test_saliency(case):
    cfg, state, d = load_case(case)
    batch = make_batch(cfg, d)
    model = make_model(cfg, state, np.float32)
    for cls in (0, 1):
        sal = model.compute_saliency(batch[0], cls)
        assert_close(sal, d[f"saliency_cls{cls}"], rtol=5 * RTOL, what=f"saliency {cls}")


def test_isolated_node_average_is_nan():
    """graphcnn.py:157-158: average pooling divides by the degree; an isolated
    node under learn_eps=True gives 0/0 = NaN in the reference.  Reproduced."""
    cfg, state, d = load_case("tiny_s1_eps1_gsum_naverage")
    und = d["und_0"]
    und = und[(und[:, 0] != 3) & (und[:, 1] != 3)]      # isolate node 3
    g = O.OGraph(cfg["n"], edge_mat_of(und), d["feat_0"])
    model = make_model(cfg, state, np.float32)
    _, _, cache = model.forward([g], np.array([0]), training=False)
    assert np.isnan(cache["layers"][0]["pooled"][3]).all()
    assert not np.isnan(cache["layers"][0]["pooled"][0]).any()
