#!/bin/bash
# config-5 visit: the update tests, then the bench's replan cycle only (python scripts/c5_cycle.py)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "c5 or incremental or partner" 2>&1 | tail -3
timeout 300 python scripts/c5_cycle.py
