#!/usr/bin/env python
"""Two-stream score fusion (counterpart of the reference ``ensemble.py:13-33``): the joint-stream and bone-stream models
each leave ``<work_dir>/score/epoch<N>_test.pkl`` = {sample_name: logits} (``Processor.save_scores``); the fused
prediction is ``argmax(joint + alpha * bone)`` per sample, reported as top-1 / top-5 accuracy against the label file
``(sample_names, labels)`` of the dataset (``data_gen/ntu_gendata.py:158-173``).

    python ensemble.py --joint-score A.pkl --bone-score B.pkl --label val_label.pkl [--alpha 1.0]

Unlike the reference, the two pickles are matched BY SAMPLE NAME (the reference zips them by position and silently
mis-pairs if the two runs saw the samples in different orders)."""
import argparse
import pickle

import numpy as np


def fuse(joint, bone, alpha=1.0):
    """{name: logits}, {name: logits} -> {name: joint + alpha * bone} over the names both hold"""
    return {k: np.asarray(v) + alpha * np.asarray(bone[k]) for k, v in joint.items() if k in bone}


def accuracy(scores, names, labels, topk=(1, 5)):
    hits = {k: 0 for k in topk}
    total = 0
    for name, lab in zip(names, labels):
        if name not in scores:
            continue
        order = np.argsort(scores[name])
        for k in topk:
            hits[k] += int(int(lab) in order[-k:])
        total += 1
    return {k: hits[k] / max(total, 1) for k in topk}, total


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--joint-score', required=True)
    ap.add_argument('--bone-score', required=True)
    ap.add_argument('--label', required=True, help='(sample_names, labels) pickle of the evaluation split')
    ap.add_argument('--alpha', type=float, default=1.0, help='weight of the bone stream (reference --alpha)')
    arg = ap.parse_args(argv)
    with open(arg.joint_score, 'rb') as f:      # files written by Processor.save_scores / the user's dataset
        joint = pickle.load(f)
    with open(arg.bone_score, 'rb') as f:
        bone = pickle.load(f)
    with open(arg.label, 'rb') as f:
        names, labels = pickle.load(f)
    acc, total = accuracy(fuse(joint, bone, arg.alpha), names, labels)
    print(f'{total} samples  top1 {acc[1]:.4f}  top5 {acc[5]:.4f}')
    return acc


if __name__ == '__main__':
    main()
