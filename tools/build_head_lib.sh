#!/bin/bash
# libisr_sr.so of the last commit (or of the revision given as $1) into tools/lib_head/ (git-ignored): the "before" side of tools/ab_train.sh / tools/ab_bench.sh
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/isr_head && mkdir -p /tmp/isr_head tools/lib_head
git archive ${1:-HEAD} isosurfacesuperresolution_amd/csrc include | tar -x -C /tmp/isr_head
make -C /tmp/isr_head/isosurfacesuperresolution_amd/csrc -j6 ../lib/libisr_sr.so > /tmp/isr_head/build.log 2>&1
cp /tmp/isr_head/isosurfacesuperresolution_amd/lib/libisr_sr.so tools/lib_head/libisr_sr.so
ls -la tools/lib_head/libisr_sr.so
