"""Shared test helpers: golden loading and tolerance checks."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: 1e-5 relative fp32 on activations / logits.  Relative to
# the max-norm of the reference tensor (element-wise relative error is not
# meaningful for entries that cancel to ~0).
RTOL = 1e-5

# True-shape (n=400, mean degree 119, 5 layers) per-node activations: measured in
# this container, the REFERENCE's own fp32 result sits up to 2.2e-5 (hidden layer
# 4) from the fp64 oracle -- fp32 rounding amplified by depth, not an
# implementation difference.  No fp32 implementation can be pinned tighter than
# that to another fp32 implementation, so deep true-shape activations are
# checked against the fp64 oracle with the bound
#     max(RTOL, TRUE_SHAPE_FACTOR * err(reference golden, fp64 oracle)).
TRUE_SHAPE_FACTOR = 4.0
# Gradients of the true-shape case: the backward multiplies by the ReLU mask, and
# among 5 x 2 x 51,200 pre-activations a few sit within rounding of zero, so two
# correct fp32 implementations can disagree on a mask bit.  One flipped bit moves
# a bias-gradient sum by ~1e-3 relative (measured: numpy-fp32 vs numpy-fp64 differ
# in exactly one mask element of layer 4 -> 9.8e-4 on batch_norms.4.bias, while
# the reference happened not to flip).  True-shape gradients of the PRODUCT (the HIP path) are therefore
# compared at 1e-3 (round 4; 5e-3 until then -- the worst value ever measured on the GPU is 2.4e-4, see
# profiles/r04_parity_true_shape.md); the tiny cases keep 5e-5.  The numpy-fp32 oracle, a noisier fp32
# implementation that is test infrastructure and not the product, keeps 5e-3 (ORACLE32_GRAD_RTOL).
TRUE_SHAPE_GRAD_RTOL = 1e-3
ORACLE32_GRAD_RTOL = 5e-3
# A mask bit can also flip between fp32 and fp64 arithmetic as such: in true_s0_eps1_gaverage_naverage the
# reference and the numpy-fp32 oracle agree with each other to 1e-4 on every gradient but BOTH sit 1.5e-2 from the
# fp64 oracle (a pre-activation within fp32 rounding of zero: every fp32 implementation takes one side, exact
# arithmetic the other).  A correct fp32 gradient therefore matches the reference's golden OR the fp64 oracle
# (whichever decided its borderline masks the same way) -- assert_grad_true_shape below -- and the two anchors
# may themselves be this far apart:
TRUE_SHAPE_GRAD_ANCHOR_GAP = 5e-2
# The numpy-fp32 oracle is a noisier fp32 implementation than either torch or the HIP path (naive fp32 column
# sums in BatchNorm, sequential fp32 spmm): its own deep-layer activations get this factor instead of
# TRUE_SHAPE_FACTOR.  It is not the product; the pin is the fp64 oracle against the goldens.
ORACLE32_FACTOR = 10.0


def golden_cases(prefix=""):
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    # state_*: seeded state dicts; train_*: training trajectories (tests/test_oracle_train_golden.py)
    return [os.path.basename(f)[:-4] for f in files if not os.path.basename(f).startswith(("state_", "train_"))]


def load_case(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    L, m, f0, H, C, le, B, n = [int(x) for x in d["cfg"]]
    cfg = dict(L=L, m=m, f0=f0, H=H, C=C, learn_eps=bool(le), B=B, n=n,
               gpool=str(d["gpool"]), npool=str(d["npool"]), row_stride=int(d["row_stride"]),
               np_seed=int(d["np_seed"]))
    state = dict(np.load(os.path.join(GOLDEN_DIR, str(d["state_file"]))))
    return cfg, state, d


def edge_mat_of(und):
    """util.py:99-103: all (i,j) pairs then all (j,i) pairs, as a [2,E] int64."""
    e = np.asarray(und, dtype=np.int64).reshape(-1, 2)
    return np.ascontiguousarray(np.concatenate([e, e[:, ::-1]], 0).T)


def neighbors_of(und, n):
    """graph.neighbors as the golden generator's SynthGraph builds them (util.py:86-90: both ends of every undirected
    edge, in edge order); read by the "max" neighbour pooling only (graphcnn.py:59-67)."""
    nb = [[] for _ in range(n)]
    for i, j in np.asarray(und, dtype=np.int64).reshape(-1, 2):
        nb[int(i)].append(int(j))
        nb[int(j)].append(int(i))
    return nb


def rel_err(a, ref, floor=0.0):
    """max|a-ref| / max(max|ref|, floor).  `floor` is an absolute scale for
    quantities that are analytically zero (e.g. the bias gradient of a Linear
    that feeds a train-mode BatchNorm), where both sides are rounding noise."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    nan_a, nan_r = np.isnan(a), np.isnan(ref)
    assert np.array_equal(nan_a, nan_r), "NaN pattern differs"
    scale = np.max(np.abs(ref[~nan_r])) if (~nan_r).any() else 0.0
    scale = max(scale, floor)
    if scale == 0.0:
        return float(np.max(np.abs(a[~nan_a]))) if (~nan_a).any() else 0.0
    return float(np.max(np.abs(a[~nan_a] - ref[~nan_r])) / scale)


def grad_floor(d):
    """2e-2 x the largest gradient entry of the golden case: the scale below
    which a gradient tensor is treated as 'analytically zero'."""
    m = 0.0
    for k in d:
        if k.startswith("grad_") and d[k].size:
            m = max(m, float(np.max(np.abs(d[k]))))
    return 2e-2 * m


def assert_close(a, ref, rtol=RTOL, what="", floor=0.0):
    e = rel_err(a, ref, floor)
    assert e <= rtol, f"{what}: rel err {e:.3e} > {rtol:.1e}"
    return e


def assert_grad_true_shape(a, golden, truth64, what="", floor=0.0, rtol=TRUE_SHAPE_GRAD_RTOL):
    """True-shape gradient check (see TRUE_SHAPE_GRAD_RTOL / TRUE_SHAPE_GRAD_ANCHOR_GAP): `a` must be within rtol of
    the reference's golden or of the fp64 oracle; golden=None (tensor not stored) leaves the fp64 oracle.
    Returns (err vs golden or None, err vs fp64)."""
    e_t = rel_err(a, truth64, floor)
    e_g = rel_err(a, golden, floor) if golden is not None else None
    if golden is not None:
        gap = rel_err(golden, truth64, floor)
        assert gap <= TRUE_SHAPE_GRAD_ANCHOR_GAP, f"{what}: golden is {gap:.2e} from the fp64 oracle"
    best = e_t if e_g is None else min(e_t, e_g)
    assert best <= rtol, f"{what}: err vs golden {e_g}, vs fp64 oracle {e_t:.3e} > {rtol:.1e}"
    return e_g, e_t


class Calibrated:
    """Calibrated comparison for the true-shape case (see TRUE_SHAPE_FACTOR).
    Keeps the running maximum of the reference's own error vs the fp64 oracle
    along the layer sequence, because a tensor inherits the noise of its inputs."""

    def __init__(self, base_rtol=RTOL, factor=TRUE_SHAPE_FACTOR):
        self.base = base_rtol
        self.factor = factor
        self.noise = 0.0
        self.log = []          # (what, error of the checked value vs fp64, the reference's own error, bound applied)
        self.log_golden = []   # error of the checked value vs the reference's golden directly (reported, not asserted)

    def check(self, a, golden, truth64, what="", floor=0.0):
        ref_noise = rel_err(golden, truth64, floor)
        assert ref_noise <= 1e-3, f"{what}: reference golden is {ref_noise:.2e} from the fp64 oracle"
        self.noise = max(self.noise, ref_noise)
        bound = max(self.base, self.factor * self.noise)
        e = rel_err(a, truth64, floor)
        assert e <= bound, f"{what}: err vs fp64 oracle {e:.3e} > {bound:.2e} (reference noise {self.noise:.2e})"
        self.log.append((what, e, ref_noise, bound))
        self.log_golden.append(rel_err(a, golden, floor))
        return e
