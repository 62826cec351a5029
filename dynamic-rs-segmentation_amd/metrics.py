"""Scores derived from a K x K confusion matrix (rows = label, cols = prediction).

The reference computes them partly by hand (isprs:526-529, 1298-1303, 1601-1605) and partly with sklearn on the
flattened label/prediction arrays (cohen_kappa_score, f1_score: isprs:1305-1310, 1607-1608); all of them are
functions of the confusion matrix, which is what the device produces.
"""
import numpy as np


def overall_and_normalized(cm):
    """(#correct, overall accuracy, class-normalised accuracy): mean of per-class recalls, classes without
    pixels contribute 0 and the divisor is always K (isprs:526-529)."""
    cm = np.asarray(cm, dtype=np.float64)
    rows = cm.sum(axis=1)
    rec = np.where(rows != 0, np.diag(cm) / np.where(rows != 0, rows, 1), 0.0)
    tot = cm.sum()
    return int(np.trace(cm)), (np.trace(cm) / tot if tot else 0.0), float(rec.sum() / cm.shape[0])


def f1_per_class(cm):
    """sklearn f1_score(average=None) over the labels present in y_true or y_pred."""
    cm = np.asarray(cm, dtype=np.float64)
    tp = np.diag(cm)
    fp = cm.sum(axis=0) - tp
    fn = cm.sum(axis=1) - tp
    present = (cm.sum(axis=0) + cm.sum(axis=1)) > 0
    den = 2 * tp + fp + fn
    f1 = np.where(den > 0, 2 * tp / np.where(den > 0, den, 1), 0.0)
    return f1[present], present


def f1_macro(cm):
    f1, _ = f1_per_class(cm)
    return float(f1.mean()) if len(f1) else 0.0


def cohen_kappa(cm):
    """sklearn cohen_kappa_score(y_true, y_pred) from the confusion matrix."""
    cm = np.asarray(cm, dtype=np.float64)
    n = cm.sum()
    if n == 0:
        return 0.0
    po = np.trace(cm) / n
    pe = float((cm.sum(axis=0) * cm.sum(axis=1)).sum()) / (n * n)
    return float((po - pe) / (1 - pe)) if pe != 1 else 0.0
