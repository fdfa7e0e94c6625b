"""ELEVATER metrics (mvlpt_amd/metrics.py) against the outputs of the reference's own functions
(trainers/vision_benchmark/datasets/metrics.py:1254-1294, run on seeded inputs by oracle/make_golden.py `metrics`),
plus the edge cases the definitions imply.  CPU only."""
import numpy as np
import pytest

from mvlpt_amd import metrics as M
from tests.golden_util import load_npz

TOL = 1e-12     # identical definitions; only float summation order may differ


@pytest.fixture(scope="module")
def golden():
    return load_npz("metrics")


def test_metrics_match_the_reference_outputs(golden):
    g = golden
    for i in g["cases"]:
        s, y, oh = g[f"c{i}_score"], g[f"c{i}_y"], g[f"c{i}_onehot"]
        assert abs(M.accuracy(y, s) - float(g[f"c{i}_accuracy"])) <= TOL
        assert abs(M.balanced_accuracy_score(oh, s) - float(g[f"c{i}_mean_per_class"])) <= TOL
        assert abs(M.map_11_points(oh, s) - float(g[f"c{i}_map11"])) <= TOL
        if f"c{i}_roc_auc" in g:
            assert abs(M.roc_auc(oh, s) - float(g[f"c{i}_roc_auc"])) <= TOL
        else:
            with pytest.raises(ValueError):           # a column with one label value: sklearn raises, so do we
                M.roc_auc(oh, s)


def test_integer_targets_equal_one_hot_targets(golden):
    g = golden
    s, y, oh = g["c0_score"], g["c0_y"], g["c0_onehot"]
    assert M.balanced_accuracy_score(y, s) == M.balanced_accuracy_score(oh, s)
    assert M.map_11_points(y, s) == M.map_11_points(oh, s)


def test_perfect_and_inverted_scores():
    y = np.array([0, 1, 2, 1, 0, 2, 2])
    oh = np.eye(3, dtype=int)[y]
    assert M.accuracy(y, oh.astype(float)) == 1.0
    assert M.balanced_accuracy_score(oh, oh.astype(float)) == 1.0
    assert M.map_11_points(oh, oh.astype(float)) == 1.0
    assert M.roc_auc(oh, oh.astype(float)) == 1.0
    assert M.roc_auc(oh, -oh.astype(float)) == 0.0
    assert M.accuracy(y, -oh.astype(float)) == 0.0


def test_constant_scores_and_empty_inputs():
    y = np.array([0, 1, 1, 0])
    oh = np.eye(2, dtype=int)[y]
    flat = np.zeros((4, 2))
    assert M.roc_auc(oh, flat) == 0.5                       # all ties: chance level
    assert M.accuracy(y, flat) == 0.5                       # arg-max of a tie is column 0
    assert M.accuracy(np.zeros(0, dtype=int), np.zeros((0, 3))) == 0.0
    assert M.map_11_points(np.zeros((0, 3), dtype=int), np.zeros((0, 3))) == 0.0
    assert M.balanced_accuracy_score(np.zeros((0, 3), dtype=int), np.zeros((0, 3))) == 0.0


def test_unused_class_columns_are_dropped_before_the_argmax():
    # class 1 never occurs; its (highest) scores must not count as predictions (metrics.py:214-230)
    y = np.array([0, 2, 0, 2])
    oh = np.eye(3, dtype=int)[y]
    s = np.array([[0.2, 9.0, 0.1], [0.1, 9.0, 0.3], [0.5, 9.0, 0.4], [0.6, 9.0, 0.7]])
    assert M.balanced_accuracy_score(oh, s) == 1.0
    assert M.accuracy(y, s) == 0.0                          # plain top-1 does see the column


def test_get_metric_names():
    for n in ("accuracy", "mean-per-class", "11point_mAP", "roc_auc"):
        assert callable(M.get_metric(n))
    with pytest.raises(KeyError):
        M.get_metric("f1")
