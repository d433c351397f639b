"""Segmentation metrics of the reference (mmseg/core/evaluation/metrics.py:26-131, 280-400) on the GPU: the three
`torch.histc` passes of `intersect_and_union` are ONE counting kernel with exact integer accumulators
(kernels.confusion_counts); totals of several images stay on the device, and the cross-GPU sum
(mmseg's collect_results + CPU sum) is one all-reduce of 3 x num_classes integers.
Return conventions are the reference's: float64 tensors from (total_)intersect_and_union, numpy arrays in eval_metrics."""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from . import kernels as K
from ._lib import S4FError


def f_score(precision, recall, beta=1):
    """metrics.py:9-23"""
    return (1 + beta ** 2) * (precision * recall) / ((beta ** 2 * precision) + recall)


def _as_device_u8(x, device):
    if isinstance(x, str):
        raise S4FError('file-name inputs belong to the dataset layer (outside SURVEY §8): pass arrays')
    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    if t.dtype != torch.uint8:
        if t.numel() and (int(t.min()) < 0 or int(t.max()) > 255):
            raise S4FError('label / prediction ids must fit uint8 (0..255)')
        t = t.to(torch.uint8)
    return t.to(device).contiguous()


def _prepare_label(label, label_map, reduce_zero_label):
    """metrics.py:63-69 on the label map (before it is compared with ignore_index)"""
    if label_map:
        label = label.clone()
        src = label.clone()
        for old_id, new_id in label_map.items():
            label[src == old_id] = new_id
    if reduce_zero_label:
        label = label.clone()
        label[label == 0] = 255
        label = label - 1
        label[label == 254] = 255
    return label


def count_areas(pred_label, label, num_classes, ignore_index, label_map=None, reduce_zero_label=False, counts=None, device=None):
    """accumulate (intersect, prediction, label) pixel counts into `counts` (device int64 [3, num_classes]; created if None)"""
    device = torch.device(device if device is not None else 'cuda')
    pred = _as_device_u8(pred_label, device)
    lab = _prepare_label(_as_device_u8(label, device), label_map, reduce_zero_label)
    if pred.numel() != lab.numel():
        raise S4FError(f'prediction {tuple(pred.shape)} and label {tuple(lab.shape)} differ in size')
    if counts is None:
        counts = torch.zeros(3, num_classes, device=device, dtype=torch.int64)
    K.confusion_counts(pred, lab, num_classes, ignore_index, counts)
    return counts


def _areas(counts):
    c = counts.to('cpu', torch.float64)
    inter, pred, lab = c[0], c[1], c[2]
    return inter, pred + lab - inter, pred, lab


def intersect_and_union(pred_label, label, num_classes, ignore_index, label_map=dict(), reduce_zero_label=False):
    """metrics.py:26-85 -> (area_intersect, area_union, area_pred_label, area_label), float64 CPU tensors [num_classes]"""
    return _areas(count_areas(pred_label, label, num_classes, ignore_index, label_map, reduce_zero_label))


def total_intersect_and_union(results, gt_seg_maps, num_classes, ignore_index, label_map=dict(), reduce_zero_label=False,
                              distributed=False):
    """metrics.py:88-131; distributed=True sums the counts of all ranks (each rank passes ITS images)"""
    counts = None
    for result, gt in zip(results, gt_seg_maps):
        counts = count_areas(result, gt, num_classes, ignore_index, label_map, reduce_zero_label, counts)
    if counts is None:
        counts = torch.zeros(3, num_classes, device='cuda', dtype=torch.int64)
    if distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts)
    return _areas(counts)


def total_area_to_metrics(total_area_intersect, total_area_union, total_area_pred_label, total_area_label, metrics=['mIoU'],
                          nan_to_num=None, beta=1):
    """metrics.py:330-400"""
    if isinstance(metrics, str):
        metrics = [metrics]
    if not set(metrics).issubset({'mIoU', 'mDice', 'mFscore'}):
        raise KeyError('metrics {} is not supported'.format(metrics))
    all_acc = total_area_intersect.sum() / total_area_label.sum()
    ret = OrderedDict({'aAcc': all_acc})
    for metric in metrics:
        if metric == 'mIoU':
            ret['IoU'] = total_area_intersect / total_area_union
            ret['Acc'] = total_area_intersect / total_area_label
        elif metric == 'mDice':
            ret['Dice'] = 2 * total_area_intersect / (total_area_pred_label + total_area_label)
            ret['Acc'] = total_area_intersect / total_area_label
        elif metric == 'mFscore':
            precision = total_area_intersect / total_area_pred_label
            recall = total_area_intersect / total_area_label
            ret['Fscore'] = torch.tensor([f_score(x[0], x[1], beta) for x in zip(precision, recall)])
            ret['Precision'] = precision
            ret['Recall'] = recall
    ret = {k: v.numpy() for k, v in ret.items()}
    if nan_to_num is not None:
        ret = OrderedDict({k: np.nan_to_num(v, nan=nan_to_num) for k, v in ret.items()})
    return ret


def eval_metrics(results, gt_seg_maps, num_classes, ignore_index, metrics=['mIoU'], nan_to_num=None, label_map=dict(),
                 reduce_zero_label=False, beta=1, distributed=False):
    """metrics.py:253-293"""
    return total_area_to_metrics(*total_intersect_and_union(results, gt_seg_maps, num_classes, ignore_index, label_map,
                                                            reduce_zero_label, distributed), metrics, nan_to_num, beta)


def mean_iou(results, gt_seg_maps, num_classes, ignore_index, nan_to_num=None, label_map=dict(), reduce_zero_label=False):
    """metrics.py:133-166"""
    return eval_metrics(results, gt_seg_maps, num_classes, ignore_index, ['mIoU'], nan_to_num, label_map, reduce_zero_label)


def pre_eval_to_metrics(pre_eval_results, metrics=['mIoU'], nan_to_num=None, beta=1):
    """metrics.py:296-327"""
    pre = tuple(zip(*pre_eval_results))
    assert len(pre) == 4
    return total_area_to_metrics(sum(pre[0]), sum(pre[1]), sum(pre[2]), sum(pre[3]), metrics, nan_to_num, beta)
