#!/bin/bash
# flakiness check: the whole GPU suite twice more, plus 5 repeats of the stream-sensitive tests
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_moco_gpu.py -q -k "reproducible or cfg1_bf16 or small_bf16" -p no:cacheprovider 2>&1 | tail -1; done
