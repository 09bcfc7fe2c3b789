#!/bin/bash
# Same surface as the reference's SAIS/main.sh (:1-30): run from the repo root  ->  bash ./SAIS/main.sh -f <videoname>
# The ffmpeg frame dump (video_to_frames.sh) is out of this build's scope: frames are read from
# ./SAIS/images/<video>/.  Every other stage runs: the flow maps (RAFT, parity unpinned: sais_amd/raft.py), the two feature
# extractions and the inference on the MI355X kernels.
while getopts f:s: flag
do
    case "${flag}" in
        f) videoname=${OPTARG};;
        s) synthetic=${OPTARG};;     # extension: -s N = use N synthetic frames instead of ./SAIS/images/<video>/
    esac
done
SYN=""
if [ -n "$synthetic" ]; then SYN="--synthetic_frames $synthetic"; fi
# RAFT weights of the flow stage (ptlflow's "things" state dict; not shipped: no network here).  Without the file the stage
# only accepts flows the user has put under ./SAIS/flows/<video>/ and exits non-zero otherwise; a synthetic smoke run (-s)
# asks for seeded random weights explicitly — its folder is marked and regenerated once a checkpoint exists.
RAFT_CHECKPOINT=${RAFT_CHECKPOINT:-./SAIS/scripts/raft_things.pth}
if [ -f "$RAFT_CHECKPOINT" ]; then RAFT="--raft_checkpoint $RAFT_CHECKPOINT"
elif [ -n "$synthetic" ]; then RAFT="--raft_random_weights"
else RAFT=""; fi


# generate paths to frames and flows and save as csv files
python ./SAIS/scripts/generate_paths.py -f $videoname -p ./SAIS/ $SYN || exit 1

# generate flow maps (skipped for a video whose ./SAIS/flows/<video>/ is already complete, as in the reference)
python ./SAIS/scripts/extract_representations.py --arch vit_small --patch_size 16 --model_type ViT_SelfSupervised_ImageNet --batch_size_per_gpu 2 --data_path ./SAIS/ --data_list Custom --save_type h5 --optical_flow --video $videoname $SYN $RAFT || exit 1

# extract representations of rgb images
python ./SAIS/scripts/extract_representations.py --arch vit_small --patch_size 16 --model_type ViT_SelfSupervised_ImageNet --batch_size_per_gpu 1024 --data_path ./SAIS/ --data_list Custom --save_type h5 --video $videoname $SYN || exit 1

# extract representations of flow maps
python ./SAIS/scripts/extract_representations.py --arch vit_small --patch_size 16 --model_type ViT_SelfSupervised_ImageNet --batch_size_per_gpu 256 --data_path ./SAIS/ --data_list Custom --save_type h5 --optical_flow_to_reps --video $videoname $SYN || exit 1

# perform inference
python ./SAIS/scripts/run_experiments.py -p ./SAIS/ -data Custom_Gestures -d Custom -m ViT -enc ViT_SelfSupervised_ImageNet -t Prototypes -mod RGB-Flow -dim 384 -bs 2 -lr 1e-1 -nc 2 -bc -sa -domains in_vs_out -ph Custom_inference -dt reps -e 1 -f 1 --inference || exit 1

# process inference results to generate valid predictions
python ./SAIS/scripts/process_inference_results.py -p ./SAIS/ || exit 1
