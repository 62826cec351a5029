"""CPU: confusion-matrix scores equal sklearn's on the flattened arrays (what the reference calls)."""
import numpy as np
from sklearn.metrics import cohen_kappa_score, f1_score

from drs_amd import metrics as M
from oracle import host_ref as H


def test_scores_match_sklearn_and_reference_formulas():
    rng = np.random.default_rng(0)
    for K, drop in [(6, None), (6, 5), (7, 2), (2, None)]:
        t = rng.integers(0, K, size=5000)
        p = rng.integers(0, K, size=5000)
        if drop is not None:
            t[t == drop] = 0
            p[p == drop] = 1
        cm = np.zeros((K, K), dtype=np.int64)
        np.add.at(cm, (t, p), 1)
        assert abs(M.cohen_kappa(cm) - cohen_kappa_score(t, p)) < 1e-12
        assert abs(M.f1_macro(cm) - f1_score(t, p, average="macro")) < 1e-12
        np.testing.assert_allclose(M.f1_per_class(cm)[0], f1_score(t, p, average=None), atol=1e-12)
        track = np.zeros((K, K), dtype=np.uint32)
        acc, accn, loc = H.calc_accuracy_by_crop(t.reshape(1, 50, 100), p.reshape(1, 50, 100), track, None, K)
        a2, _, n2 = M.overall_and_normalized(cm)
        assert a2 == acc and abs(n2 - accn) < 1e-15
