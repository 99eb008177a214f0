#!/bin/bash
mkdir -p gpurun_out/r4kronc
rm -f gpurun_out/r4kronc/quick.txt
QBH_KRONC_FAR_NT=0 bash tools/r4/kronc_quick.sh far_plain
QBH_KRONC_FAR_NT=1 bash tools/r4/kronc_quick.sh far_nt
