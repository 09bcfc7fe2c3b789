#!/usr/bin/env python3
"""Drop-in for the reference's SAIS/scripts/generate_paths.py (:9-67): writes <path>/paths/Custom_Paths.csv (one row per
frame: path, category, label) and Custom_FlowPaths.csv (every 15th frame paired with the frame 15 later, plus the
flow-image path flows/<video>/flows_<n:08d>.jpg).  Extension: --synthetic_frames N lists N frame names without
touching <path>/images (used by `main.sh -s N`)."""
import argparse
import csv
import os
import time

JUMP_FRAMES = 15                                              # fps // 2  (:44)


def frame_files(root, video, synthetic):
    if synthetic:
        return ['frames_%08d.jpg' % i for i in range(synthetic)]
    return sorted(os.listdir(os.path.join(root, 'images', video)))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('-f', '--videoname', type=str)
    parser.add_argument('-p', '--path', type=str)
    parser.add_argument('--synthetic_frames', type=int, default=0)
    args = parser.parse_args()
    t0 = time.time()
    video = args.videoname
    os.makedirs(os.path.join(args.path, 'paths'), exist_ok=True)
    files = frame_files(args.path, video, args.synthetic_frames)
    with open(os.path.join(args.path, 'paths', 'Custom_Paths.csv'), 'w', newline='') as fh:
        w = csv.writer(fh, lineterminator='\n')
        w.writerow(['', 'path', 'category', 'label'])
        for i, name in enumerate(files):
            w.writerow([i, os.path.join('images', video, name), video, video])
    with open(os.path.join(args.path, 'paths', 'Custom_FlowPaths.csv'), 'w', newline='') as fh:
        w = csv.writer(fh, lineterminator='\n')
        w.writerow(['', 'path1', 'path2', 'category', 'label', 'flowpath'])
        for row, k in enumerate(range(0, len(files) - JUMP_FRAMES, JUMP_FRAMES)):
            frame = int(files[k].split('_')[-1].strip('.jpg'))                   # frames_00000123.jpg -> 123 (:52)
            w.writerow([row, os.path.join('images', video, files[k]),
                        os.path.join('images', video, 'frames_%08d.jpg' % (frame + JUMP_FRAMES)), video, video,
                        os.path.join('flows', video, 'flows_%08d.jpg' % (frame // JUMP_FRAMES))])
    print('Time taken (s): %.3f' % (time.time() - t0))


if __name__ == '__main__':
    main()
