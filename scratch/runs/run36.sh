#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/host_slack.py 2>&1 | tail -1
python scratch/host_slack.py configs/clip/vit-b-32_synthetic.yaml 128 2>&1 | tail -1
python scratch/host_slack.py configs/mae/mae_vit_b_synthetic.yaml 256 2>&1 | tail -1
python scratch/host_slack.py configs/simclr/simclr_r50_synthetic.yaml 64 2>&1 | tail -1
