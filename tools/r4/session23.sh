#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_hubrepr.py -x -q -m gpu -k "two_body" --durations=2 2>&1 | tail -6
