#!/usr/bin/env python3
"""Drop-in for the reference's SAIS/scripts/process_inference_results.py (:11-14,201-261): same flag (-p/--path),
reads <path>/paths/Custom_Paths.csv and <path>/params/Fold_0/{reps_and_labels_Custom_inference, prototypes.zip},
writes <path>/results/Custom_inference_gestures.csv.  The logic lives in sais_amd/postprocess.py."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('-p', '--path', type=str)
    args = parser.parse_args()
    t0 = time.time()
    from sais_amd.postprocess import process
    dst, rows = process(args.path, folds=(0,), inference_set='Custom_inference')
    print('%i gesture intervals -> %s' % (len(rows), dst))
    print('Time taken (s): %.3f' % (time.time() - t0))


if __name__ == '__main__':
    main()
