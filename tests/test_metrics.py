"""CPU: J&F metric (swem_amd/metrics.py) -- the reference toolkit's only self-contained known-answer test
(evaluation/pytest/test_evaluation.py:118-128, test_void_masks) plus hand-checkable cases."""
import numpy as np

from swem_amd import metrics as M


def test_void_masks_kat():
    gt = np.zeros((2, 200, 200))
    mask = np.zeros((2, 200, 200))
    void = np.zeros((2, 200, 200))
    gt[:, 100:150, 100:150] = 1
    void[:, 50:100, 100:150] = 1
    mask[:, 50:150, 100:150] = 1
    assert np.mean(M.db_eval_iou(gt, mask, void)) == 1
    assert np.mean(M.db_eval_boundary(gt, mask, void)) == 1


def test_iou_and_boundary_hand_cases():
    a = np.zeros((40, 40))
    b = np.zeros((40, 40))
    a[10:30, 10:30] = 1
    b[10:30, 20:40] = 1                      # half overlap: |A&B| = 200, |A|B| = 600
    assert abs(M.db_eval_iou(a, b) - 200 / 600) < 1e-12
    assert M.db_eval_iou(np.zeros((8, 8)), np.zeros((8, 8))) == 1      # empty union -> 1 (metrics.py:33-36)
    assert M.f_measure(a, a) == 1 and M.f_measure(np.zeros((8, 8)), np.zeros((8, 8))) == 1
    assert M.f_measure(a, np.zeros((40, 40))) == 0                      # prediction without ground truth
    far = np.zeros((40, 40))
    far[0:4, 0:4] = 1
    assert M.f_measure(far, a) < 0.2
    assert M.disk(2).astype(int).tolist() == [[0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0],
                                              [0, 0, 1, 0, 0]]


def test_statistics_and_sequence_protocol():
    m, o, d = M.db_statistics(np.array([1.0, 0.8, 0.6, 0.4, 0.9, 0.7, 0.5, 0.3]))
    assert abs(m - 0.65) < 1e-12 and abs(o - 5 / 8) < 1e-12 and d > 0
    gt = np.zeros((5, 32, 32), dtype=np.int64)
    gt[:, 4:12, 4:12] = 1
    gt[:, 16:28, 16:28] = 2
    pred = gt.copy()
    pred[0] = 0                              # first and last frames are excluded by the protocol (evaluation.py:289-290)
    pred[-1] = 0
    r = M.evaluate_semisupervised(gt, pred)
    assert r['J&F-Mean'] == 1.0 and len(r['J']) == 2
    pred[2, 16:28, 16:28] = 0                # lose object 2 on one of the three scored frames
    r = M.evaluate_semisupervised(gt, pred)
    assert abs(r['J'][1][0] - 2 / 3) < 1e-12 and r['J'][0][0] == 1.0
