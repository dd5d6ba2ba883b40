"""Batch Dice evaluation with the reference's entry point and CSV layout (core/seg_eval.py:8-57 `cal_dsc_batch`):
one row per case with `label<k>_score`, `label<k>_type` columns, then a `mean` and a `std` row.

Differences by construction: label volumes are read with the built-in MetaImage reader (no SimpleITK), all labels of
a case are counted in ONE device pass (utils/metrics.py), and the statistics rows are appended with `pandas.concat`
(`DataFrame.append`, used by the reference at :56, no longer exists)."""
import os

import pandas as pd

from segmentation3d.utils.metrics import cal_dsc_labels
from segmentation3d.utils.mha_io import read_mha


def cal_dsc_batch(gt_files, seg_files, labels, threshold, save_csv_file_path):
    """
    gt_files / seg_files: equally long lists of label-volume files (.mha / .mhd)
    labels: labels to score;  threshold: minimal voxel count for a label to count as present
    save_csv_file_path: result csv (None -> only return the DataFrame)
    """
    assert isinstance(gt_files, list) and isinstance(seg_files, list)
    assert len(gt_files) == len(seg_files)
    result_content = []
    for gt_case_path, seg_case_path in zip(gt_files, seg_files):
        gt = read_mha(gt_case_path, dtype=None)
        seg = read_mha(seg_case_path, dtype=None)
        case_name = os.path.basename(gt_case_path)
        content = [case_name]
        for label, (score, seg_type) in zip(labels, cal_dsc_labels(gt, seg, labels, threshold)):
            content.extend([score, seg_type])
            print('case_name: {}, label: {}, score: {}, type: {}'.format(case_name, label, score, seg_type))
        result_content.append(content)
    column = ['filename']
    for label in labels:
        column.extend(['label{}_score'.format(label), 'label{}_type'.format(label)])
    df = pd.DataFrame(data=result_content, columns=column)
    statistics_content = [['mean'], ['std']]
    for label in labels:
        mean, std = df['label{}_score'.format(label)].mean(), df['label{}_score'.format(label)].std()
        print(mean, std)
        statistics_content[0].extend([mean, 'ignore_type'])
        statistics_content[1].extend([std, 'ignore_type'])
    df = pd.concat([df, pd.DataFrame(data=statistics_content, columns=column)])
    if save_csv_file_path:
        df.to_csv(save_csv_file_path)
    return df
