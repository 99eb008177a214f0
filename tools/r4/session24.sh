#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ragged.py -x -q -m gpu -k "communicator" 2>&1 | tail -15
