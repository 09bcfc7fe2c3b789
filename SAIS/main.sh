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


# generate paths to frames and flows and save as csv files
python ./SAIS/scripts/generate_paths.py -f $videoname -p ./SAIS/ $SYN || exit 1

# generate flow maps (skipped for a video whose ./SAIS/flows/<video>/ already exists, as in the reference)
python ./SAIS/scripts/extract_representations.py --arch vit_small --patch_size 16 --model_type ViT_SelfSupervised_ImageNet --batch_size_per_gpu 2 --data_path ./SAIS/ --data_list Custom --save_type h5 --optical_flow --video $videoname $SYN || exit 1

# extract representations of rgb images
python ./SAIS/scripts/extract_representations.py --arch vit_small --patch_size 16 --model_type ViT_SelfSupervised_ImageNet --batch_size_per_gpu 1024 --data_path ./SAIS/ --data_list Custom --save_type h5 --video $videoname $SYN || exit 1

# extract representations of flow maps
python ./SAIS/scripts/extract_representations.py --arch vit_small --patch_size 16 --model_type ViT_SelfSupervised_ImageNet --batch_size_per_gpu 256 --data_path ./SAIS/ --data_list Custom --save_type h5 --optical_flow_to_reps --video $videoname $SYN || exit 1

# perform inference
python ./SAIS/scripts/run_experiments.py -p ./SAIS/ -data Custom_Gestures -d Custom -m ViT -enc ViT_SelfSupervised_ImageNet -t Prototypes -mod RGB-Flow -dim 384 -bs 2 -lr 1e-1 -nc 2 -bc -sa -domains in_vs_out -ph Custom_inference -dt reps -e 1 -f 1 --inference || exit 1

# process inference results to generate valid predictions
python ./SAIS/scripts/process_inference_results.py -p ./SAIS/ || exit 1
