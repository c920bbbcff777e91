#!/usr/bin/env python
"""Copies the reference's shipped classifier DATA files (no code) into tests/golden/model_parameters/:

  trained_alignment_classifier.txt  intercept + 6 coefficients of the combined CorAl/CFEAR alignment classifier
                                    (written by LogisticRegression::SaveCoefficients, alignmentinterface.cpp:255-269)
  trained_loop_classifier.txt       intercept + 3 coefficients of the loop verification classifier
  tbv_model_8.txt                   its training rows "y,odom-bounds,sc-sim,alignment_quality" (SaveData, :152-173)
  combined_head.txt                 the first 1300 rows (100 keyframe pairs x 13 perturbations) of combined.txt:
                                    "y,joint,sep,overlap,cost,#residuals,mean #cells" -- real CorAl / GetCost outputs

They are the only reference-generated vectors that touch this path: they pin the two text formats,
the feature order and -- statistically -- the range of real CorAl / CFEAR quality values.
    python tests/golden/copy_reference_data.py        (needs /root/reference)"""
import os
import shutil

SRC = "/root/reference/tbv_slam/model_parameters"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_parameters")

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for f in ("trained_alignment_classifier.txt", "trained_loop_classifier.txt", "tbv_model_8.txt"):
        shutil.copyfile(os.path.join(SRC, f), os.path.join(DST, f))
    with open(os.path.join(SRC, "combined.txt")) as fi, open(os.path.join(DST, "combined_head.txt"), "w") as fo:
        for i, ln in enumerate(fi):
            if i >= 1300:
                break
            fo.write(ln)
