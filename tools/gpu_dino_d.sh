#!/bin/bash
tag=${1:-dino_d}
bash tools/gpu_dino_a.sh $tag
bash tools/gpu_dino_b.sh $tag
