#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_prnn.py -q -m gpu -x -k "reference_training or rccl or gradient_needs" -s 2>&1 | tail -15
