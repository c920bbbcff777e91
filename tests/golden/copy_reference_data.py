#!/usr/bin/env python
"""Builds tests/golden/model_parameters.npz from the reference's shipped classifier DATA files
(/root/reference/tbv_slam/model_parameters/; no code is read):

  align          intercept + 6 coefficients of the combined CorAl/CFEAR alignment classifier
                 (trained_alignment_classifier.txt, written by LogisticRegression::SaveCoefficients,
                 coral_alignment_quality/src/alignment_checker/alignmentinterface.cpp:255-269)
  loop           intercept + 3 coefficients of the loop verification classifier (trained_loop_classifier.txt)
  loop_rows      its 4390 training rows "y,odom-bounds,sc-sim,alignment_quality" (tbv_model_8.txt, SaveData :152-173)
  combined_head  the first 1300 rows (100 keyframe pairs x 13 perturbations) of combined.txt:
                 "y,joint,sep,overlap,cost,#residuals,mean #cells" -- real CorAl / GetCost outputs
  combined_aligned_quantiles   [1, 5, 25, 50, 75, 95, 99] % quantiles of {cost, #residuals, mean #cells} over ALL aligned
                 rows (label 1; 4 467 of the 58 071) of combined.txt, and combined_misaligned_quantiles of {cost, #residuals}
                 over the perturbed rows: the distribution gate of tests/test_oracle_pinning.py
  sha256_*       digest of each source file's bytes (of the first 1300 lines for combined.txt): the tests re-create
                 the text with the mirror's SaveData / SaveCoefficients and compare digests, which pins the two text
                 formats byte for byte without keeping the files

These are the only reference-generated numbers that touch this path: they pin the text formats, the
feature order and -- statistically -- the range of real CorAl / CFEAR quality values.
    python tests/golden/copy_reference_data.py        (needs /root/reference)"""
import hashlib
import os

import numpy as np

SRC = "/root/reference/tbv_slam/model_parameters"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_parameters.npz")


def _rows(text):
    return np.array([[float(t) for t in ln.split(",")] for ln in text.splitlines() if ln.strip()], np.float64)


if __name__ == "__main__":
    out = {}
    for key, name, head in (("align", "trained_alignment_classifier.txt", None), ("loop", "trained_loop_classifier.txt", None),
                            ("loop_rows", "tbv_model_8.txt", None), ("combined_head", "combined.txt", 1300)):
        text = open(os.path.join(SRC, name)).read()
        if head:
            text = "".join(text.splitlines(keepends=True)[:head])
        a = _rows(text)
        out[key] = a[0] if a.shape[0] == 1 else a
        out["sha256_" + key] = np.array(hashlib.sha256(text.encode()).hexdigest())
        # the numbers survive the reference's own text format: "%g" of every value reproduces the file
        again = "".join(",".join("%g" % v for v in row) + "\n" for row in a)
        assert again == text, name
    full = _rows(open(os.path.join(SRC, "combined.txt")).read())
    q = [1, 5, 25, 50, 75, 95, 99]
    al, mis = full[full[:, 0] == 1], full[full[:, 0] == 0]
    out["combined_aligned_quantiles"] = np.stack([np.percentile(al[:, c], q) for c in (4, 5, 6)])
    out["combined_misaligned_quantiles"] = np.stack([np.percentile(mis[:, c], q) for c in (4, 5)])
    out["combined_rows"] = np.array([full.shape[0], al.shape[0]])
    np.savez_compressed(DST, **out)
    print("wrote", DST, {k: getattr(v, "shape", None) for k, v in out.items()})
