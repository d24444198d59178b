"""J&F (region similarity and boundary accuracy) of DAVIS: numpy/scipy restatement of the reference's fork of the
davis2017 toolkit, ``evaluation/davis2017/metrics.py:6-178`` and ``utils.py:136-162``.

The reference needs ``cv2`` and ``skimage`` (absent here) and ``np.bool`` (removed from NumPy >= 1.24); this file
depends on numpy and scipy.ndimage only.  CPU metric code: it scores index maps, it is not on the GPU path.
KAT: the reference's own ``test_void_masks`` (evaluation/pytest/test_evaluation.py:118-128) in tests/test_metrics.py.
"""
import math

import numpy as np
from scipy import ndimage


def db_eval_iou(annotation, segmentation, void_pixels=None):
    """metrics.py:6-37: Jaccard index over the last two axes; empty union counts as 1."""
    assert annotation.shape == segmentation.shape
    annotation = annotation.astype(bool)
    segmentation = segmentation.astype(bool)
    void = np.zeros_like(segmentation) if void_pixels is None else void_pixels.astype(bool)
    inters = np.sum((segmentation & annotation) & ~void, axis=(-2, -1))
    union = np.sum((segmentation | annotation) & ~void, axis=(-2, -1))
    with np.errstate(divide='ignore', invalid='ignore'):
        j = inters / union
    if np.ndim(j) == 0:
        return 1 if np.isclose(union, 0) else j
    j[np.isclose(union, 0)] = 1
    return j


def seg2bmap(seg):
    """metrics.py:122-178 for the same-size case: 1-pixel boundaries offset half a pixel towards the origin."""
    seg = seg.astype(bool)
    assert seg.ndim == 2
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    s[:-1, :] = seg[1:, :]
    se[:-1, :-1] = seg[1:, 1:]
    b = seg ^ e | seg ^ s | seg ^ se
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = 0
    return b


def disk(radius):
    """skimage.morphology.disk: (2r+1)^2 footprint of the points with x^2 + y^2 <= r^2."""
    r = int(radius)
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r


def f_measure(foreground_mask, gt_mask, void_pixels=None, bound_th=0.008):
    """metrics.py:57-119: boundary precision/recall with a disk tolerance of bound_th * image diagonal."""
    assert np.atleast_3d(foreground_mask).shape[2] == 1
    void = np.zeros_like(foreground_mask, dtype=bool) if void_pixels is None else void_pixels.astype(bool)
    bound_pix = bound_th if bound_th >= 1 else math.ceil(bound_th * np.linalg.norm(foreground_mask.shape))
    fg_boundary = seg2bmap(foreground_mask * ~void)
    gt_boundary = seg2bmap(gt_mask * ~void)
    fp = disk(bound_pix)
    fg_dil = ndimage.binary_dilation(fg_boundary, structure=fp)      # cv2.dilate with the same footprint
    gt_dil = ndimage.binary_dilation(gt_boundary, structure=fp)
    gt_match = gt_boundary & fg_dil
    fg_match = fg_boundary & gt_dil
    n_fg, n_gt = np.sum(fg_boundary), np.sum(gt_boundary)
    if n_fg == 0 and n_gt > 0:
        precision, recall = 1, 0
    elif n_fg > 0 and n_gt == 0:
        precision, recall = 0, 1
    elif n_fg == 0 and n_gt == 0:
        precision, recall = 1, 1
    else:
        precision = np.sum(fg_match) / float(n_fg)
        recall = np.sum(gt_match) / float(n_gt)
    return 0 if precision + recall == 0 else 2 * precision * recall / (precision + recall)


def db_eval_boundary(annotation, segmentation, void_pixels=None, bound_th=0.008):
    """metrics.py:40-54."""
    assert annotation.shape == segmentation.shape
    if annotation.ndim == 3:
        return np.array([f_measure(segmentation[i], annotation[i], None if void_pixels is None else void_pixels[i],
                                   bound_th=bound_th) for i in range(annotation.shape[0])])
    if annotation.ndim == 2:
        return f_measure(segmentation, annotation, void_pixels, bound_th=bound_th)
    raise ValueError('db_eval_boundary does not support tensors with %d dimensions' % annotation.ndim)


def db_statistics(per_frame_values):
    """utils.py:136-162: mean, recall (> 0.5) and decay (first minus last quarter)."""
    v = np.asarray(per_frame_values, dtype=float)
    with np.errstate(invalid='ignore'):
        M = np.nanmean(v)
        O = np.nanmean(v > 0.5)
        ids = (np.round(np.linspace(1, len(v), 5) + 1e-10) - 1).astype(np.uint8)
        bins = [v[ids[i]:ids[i + 1] + 1] for i in range(4)]
        D = np.nanmean(bins[0]) - np.nanmean(bins[3])
    return M, O, D


def evaluate_semisupervised(gt_index_maps, pred_index_maps, num_objects=None):
    """evaluation.py:265-322 for one sequence of the semi-supervised task: index maps (T,H,W) incl. the first and
    the last frame, which the protocol excludes (``[:, 1:-1]``); returns per-object J/F statistics and J&F mean."""
    gt = np.asarray(gt_index_maps)[1:-1]
    pr = np.asarray(pred_index_maps)[1:-1]
    n = int(num_objects if num_objects is not None else gt.max())
    out = {'J': [], 'F': []}
    for o in range(1, n + 1):
        j = db_eval_iou(gt == o, pr == o)
        f = db_eval_boundary(gt == o, pr == o)
        out['J'].append(db_statistics(j))
        out['F'].append(db_statistics(f))
    jm = float(np.mean([s[0] for s in out['J']]))
    fm = float(np.mean([s[0] for s in out['F']]))
    out['J&F-Mean'] = (jm + fm) / 2
    return out
