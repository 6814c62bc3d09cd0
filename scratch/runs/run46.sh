#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 70 python -m pytest tests/test_moco_gpu.py -q -x -k "not cfg1 and not reproducible" -p no:cacheprovider 2>&1 | tail -1
