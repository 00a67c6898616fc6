"""Pin oracle/gin_torch_cpu.py -- the torch-CPU restatement bench.py times as the `cpu_baseline` SURVEY.md 8(d)
specifies -- against the golden vectors captured from the real reference.  CPU only."""
import numpy as np
import pytest

from helpers import RTOL, assert_close, edge_mat_of, golden_cases, grad_floor, load_case
from oracle import gin_oracle as O
from oracle.gin_torch_cpu import TorchCpuGIN

CASES = [c for c in golden_cases("tiny_") if "nmax" not in c]


@pytest.mark.parametrize("case", CASES)
def test_train_step_matches_reference_golden(case):
    cfg, state, d = load_case(case)
    batch = [O.OGraph(cfg["n"], edge_mat_of(d[f"und_{g}"]), d[f"feat_{g}"], int(d["labels"][g])) for g in range(cfg["B"])]
    model = TorchCpuGIN(state, cfg["L"], cfg["m"], cfg["learn_eps"], cfg["gpool"], cfg["npool"])
    out = model.train_step(batch, d["perm"], beta=0.05)
    assert_close(out["c_logit"], d["train_c_logit"], what="c_logit")
    assert_close(out["d_logit"], d["train_d_logit"], what="d_logit")
    assert_close(np.array(out["loss"]), d["train_loss"][0], what="loss")
    floor, checked = grad_floor(d), 0
    for key in d:
        if key.startswith("grad_"):
            name = key[len("grad_"):]
            assert_close(out["grads"][name].reshape(d[key].shape), d[key], rtol=5 * RTOL, what=key, floor=floor)
            checked += 1
    assert checked >= 10
    if not cfg["learn_eps"]:
        assert "eps" not in out["grads"]
    for key in d:                                   # BatchNorm running statistics after the step
        if key.startswith("bufafter_"):
            name = key[len("bufafter_"):]
            if name.endswith("num_batches_tracked"):
                assert int(model.buf[name]) == int(d[key])
            else:
                assert_close(model.buf[name].numpy(), d[key], what=key)


def test_encoder_classifier_only_variant():
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    batch = [O.OGraph(cfg["n"], edge_mat_of(d[f"und_{g}"]), d[f"feat_{g}"], int(d["labels"][g])) for g in range(cfg["B"])]
    model = TorchCpuGIN(state, cfg["L"], cfg["m"], cfg["learn_eps"], cfg["gpool"], cfg["npool"])
    out = model.train_step(batch, d["perm"], want_disc=False)
    assert out["d_logit"] is None and "disc.f_k.weight" not in out["grads"]
    assert_close(out["c_logit"], d["train_c_logit"], what="c_logit")
    assert_close(np.array(out["loss"]), d["train_loss"][1], what="CE loss")
