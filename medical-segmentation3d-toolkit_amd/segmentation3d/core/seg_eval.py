"""Batch Dice evaluation: `cal_dsc_batch` with the reference's signature, console output and CSV layout
(core/seg_eval.py:8-57) -- one row per case holding `label<k>_score` / `label<k>_type` pairs, followed by a `mean` and
a `std` row whose type cells read 'ignore_type'.

What differs underneath: label volumes come from the built-in MetaImage reader (no SimpleITK), every case is counted
in ONE device pass for all labels (utils/metrics.cal_dsc_labels -> csrc/metrics.hip), and the table is assembled from
records with `pandas.concat` (`DataFrame.append`, which the reference calls at :56, was removed from pandas)."""
import os

import pandas as pd

from segmentation3d.utils.metrics import cal_dsc_labels
from segmentation3d.utils.image_io import read_image


def _columns(labels):
    cols = ['filename']
    for label in labels:
        cols += ['label{}_score'.format(label), 'label{}_type'.format(label)]
    return cols


def _score_case(gt_path, seg_path, labels, threshold):
    """one table row: file name, then (score, type) per label; also echoes the reference's progress lines"""
    name = os.path.basename(gt_path)
    results = cal_dsc_labels(read_image(gt_path, dtype=None), read_image(seg_path, dtype=None), labels, threshold)
    row = [name]
    for label, (score, seg_type) in zip(labels, results):
        print('case_name: {}, label: {}, score: {}, type: {}'.format(name, label, score, seg_type))
        row += [score, seg_type]
    return row


def cal_dsc_batch(gt_files, seg_files, labels, threshold, save_csv_file_path):
    """
    :param gt_files, seg_files: equally long lists of label-volume files (.mha / .mhd)
    :param labels: the labels to score
    :param threshold: minimal voxel count for a label to count as present (TN / FP / FN / TP typing)
    :param save_csv_file_path: result csv; None only returns the DataFrame
    """
    assert isinstance(gt_files, list) and isinstance(seg_files, list)
    assert len(gt_files) == len(seg_files)
    cols = _columns(labels)
    cases = pd.DataFrame([_score_case(g, s, labels, threshold) for g, s in zip(gt_files, seg_files)], columns=cols)
    summary = {'mean': ['mean'], 'std': ['std']}
    for label in labels:
        scores = cases['label{}_score'.format(label)]
        mean, std = scores.mean(), scores.std()
        print(mean, std)
        summary['mean'] += [mean, 'ignore_type']
        summary['std'] += [std, 'ignore_type']
    table = pd.concat([cases, pd.DataFrame([summary['mean'], summary['std']], columns=cols)])
    if save_csv_file_path:
        table.to_csv(save_csv_file_path)
    return table
