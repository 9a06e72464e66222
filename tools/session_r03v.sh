#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_hex.py -m gpu -q -k "curved or invariants" 2>&1 | tail -8
