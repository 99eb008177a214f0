#!/bin/bash
R=/root/repo
O=$R/gpurun_out/r4s13
mkdir -p $O
cd $R
( time timeout 1700 python -m pytest tests -q -m gpu --durations=40 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -60 $O/pytest.log
