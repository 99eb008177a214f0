#!/bin/bash
# build a variant of libqbhip.so with extra compiler flags into tools/lab/variants/<name>.so (kernel experiments, A/B timing)
# usage: tools/lab/build_variant.sh <name> [-DFLAG ...]
set -e
R=/root/repo
NAME=$1; shift
D=$(mktemp -d)
mkdir -p $D/a $D/include $R/tools/lab/variants
cp -r $R/quantum_basis_amd/csrc $D/a/csrc
cp $R/include/qbhip.h $D/include/
(cd $D/a/csrc && rm -rf build && sed -i 's#^OUT .*#OUT = ../out.so#' Makefile && make -j4 EXTRA="$*" > $D/log 2>&1) || { tail -20 $D/log; exit 1; }
cp $D/a/out.so $R/tools/lab/variants/$NAME.so
rm -rf $D
echo built tools/lab/variants/$NAME.so "($*)"
