#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 100 python -m pytest tests/test_moco_gpu.py -q -x -k "cfg1 or reproducible" -p no:cacheprovider 2>&1 | tail -1
