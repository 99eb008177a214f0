#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_reforder.py -x -q -m gpu --durations=3 2>&1 | tail -12
