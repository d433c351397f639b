#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_pipeline_gpu.py -q 2>&1 | tail -12
