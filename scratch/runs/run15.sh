#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/hbm_ceiling.py 2>&1 | grep -v amdgpu.ids
