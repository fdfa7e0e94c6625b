"""ELEVATER evaluation metrics used by `MVLPT.test` (reference: trainers/vision_benchmark/datasets/metrics.py:1254-1294,
called from trainers/mvlpt.py:1048-1078).  Host-side numpy, like the reference (which leans on sklearn); restated here
without sklearn so the eval path has no dependency the GPU box may lack.  Pinned by tests/golden/metrics.npz
(outputs of the reference's own functions on seeded inputs, oracle/make_golden.py `metrics`).

All four take `(y_true, y_score)` with scores of shape (N, C):
  accuracy          y_true (N,) int                      top-1 hit rate                          (metrics.py:1254-1262, 256-291)
  mean-per-class    y_true (N, C) 0/1 or (N,) int        mean recall over the classes that occur (metrics.py:1271-1274, 839-850)
  11point_mAP       y_true (N, C) 0/1 or (N,) int        11-point interpolated AP, class mean    (metrics.py:1265-1268, 853-895)
  roc_auc           y_true (N, C) 0/1 or (N,) 0/1        macro one-vs-rest ROC-AUC               (metrics.py:1277-1280)
Columns whose target is all zero are dropped from BOTH targets and scores before mean-per-class / mAP are computed
(`filter_out_zero_tgt`, metrics.py:214-230) — so the arg-max of mean-per-class runs over the surviving columns only."""
from __future__ import annotations

import numpy as np


def _target_matrix(y_true: np.ndarray, n_class: int) -> np.ndarray:
    y_true = np.asarray(y_true)
    if y_true.ndim == 1:                                   # metrics.py:122-130
        mat = np.zeros((len(y_true), n_class), dtype=int)
        mat[np.arange(len(y_true)), y_true.astype(int)] = 1
        return mat
    return y_true


def _drop_empty_columns(y_true, y_score):
    y_score = np.asarray(y_score)
    if y_score.size == 0:
        return None, None
    tar = _target_matrix(y_true, y_score.shape[1])
    assert tar.size == y_score.size
    keep = np.where(~np.all(tar == 0, axis=0))[0]
    if keep.size == 0:
        return None, None
    return tar[:, keep], y_score[:, keep]


def accuracy(y_label, y_pred) -> float:
    y_label, y_pred = np.asarray(y_label), np.asarray(y_pred)
    assert len(y_pred) == len(y_label) and y_label.ndim == 1
    if len(y_label) == 0:
        return 0.0
    return float(np.sum(np.argmax(y_pred, axis=1) == y_label)) / len(y_label)


def balanced_accuracy_score(y_label, y_pred) -> float:
    tar, score = _drop_empty_columns(y_label, y_pred)
    if tar is None:
        return 0.0
    t, p = np.argmax(tar, axis=1), np.argmax(score, axis=1)
    recalls = [np.mean(p[t == c] == c) for c in np.unique(t)]      # classes that occur in y_true
    return float(np.mean(recalls))


def _precision_recall(targets: np.ndarray, scores: np.ndarray):
    """Precision/recall pairs ordered from the LOWEST score threshold to the highest, then (1, 0) — the order of
    sklearn.metrics.precision_recall_curve that the interpolation walk of metrics.py:875-881 relies on."""
    order = np.argsort(-scores, kind="mergesort")
    s, y = scores[order], (targets[order] == 1).astype(np.float64)
    last_of_run = np.r_[np.where(np.diff(s))[0], len(s) - 1]     # one operating point per distinct score
    tps = np.cumsum(y)[last_of_run]
    fps = 1 + last_of_run - tps
    precision = tps / (tps + fps)
    recall = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps)
    return np.r_[precision[::-1], 1.0], np.r_[recall[::-1], 0.0]


def map_11_points(y_label, y_pred_proba, n_points: int = 11) -> float:
    tar, score = _drop_empty_columns(y_label, y_pred_proba)
    if tar is None:
        return 0.0
    thresholds = np.linspace(1, 0, n_points, endpoint=True).tolist()
    per_class = []
    for c in range(score.shape[1]):
        precision, recall = _precision_recall(tar[:, c], score[:, c])
        interp = np.empty(len(thresholds))
        idx, best = 0, 0.0
        for i, th in enumerate(thresholds):
            while idx < len(recall) and th <= recall[idx]:
                best = max(best, precision[idx])
                idx += 1
            interp[i] = best
        per_class.append(np.mean(interp))
    return float(np.mean(per_class))


def _binary_auc(y: np.ndarray, s: np.ndarray) -> float:
    pos = y == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    ranks = np.empty(len(s), dtype=np.float64)
    run_start = np.r_[0, np.where(np.diff(ss))[0] + 1]
    run_end = np.r_[run_start[1:], len(ss)]
    for a, b in zip(run_start, run_end):                   # average ranks over ties == trapezoid rule on the ROC curve
        ranks[order[a:b]] = 0.5 * (a + 1 + b)
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def roc_auc(y_true, y_score) -> float:
    y_true, y_score = np.asarray(y_true), np.asarray(y_score, dtype=np.float64)
    if y_true.ndim == 1:
        if y_score.ndim == 2:
            raise ValueError("multiclass roc_auc needs an indicator matrix (the reference passes one-hot rows)")
        return _binary_auc(y_true, y_score)
    return float(np.mean([_binary_auc(y_true[:, c], y_score[:, c]) for c in range(y_true.shape[1])]))


def get_metric(metric_name: str):
    """metrics.py:1283-1294 (an unknown name logs an error and returns None there; here it raises)."""
    table = {"accuracy": accuracy, "mean-per-class": balanced_accuracy_score, "11point_mAP": map_11_points, "roc_auc": roc_auc}
    if metric_name not in table:
        raise KeyError(f"undefined metric {metric_name!r}")
    return table[metric_name]
