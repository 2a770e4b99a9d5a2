#!/bin/bash
# pk_repro.ll -> code objects under SelectionDAG and GlobalISel (build container; run check.py on a GPU box)
set -e
cd "$(dirname "$0")"
LLVM=/opt/rocm/lib/llvm/bin
for v in sdag gisel; do
  fl=""; [ $v = gisel ] && fl="-global-isel -global-isel-abort=2"
  $LLVM/llc -O3 -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 $fl -filetype=obj pk_repro.ll -o pk_$v.o
  $LLVM/ld.lld -shared pk_$v.o -o pk_$v.hsaco
done
