#!/bin/bash
# libisr_sr.so of the WORKING TREE with extra compiler defines into tools/lib_head/libisr_sr_<name>.so (git-ignored): side builds for
# tools/ab_bench.sh "ISR_SR_LIB=/root/repo/tools/lib_head/libisr_sr_<name>.so".  usage: bash tools/build_variant_lib.sh <name> -DFOO=1 ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
rm -rf /tmp/isr_var_$name && mkdir -p /tmp/isr_var_$name/isosurfacesuperresolution_amd tools/lib_head
cp -r isosurfacesuperresolution_amd/csrc /tmp/isr_var_$name/isosurfacesuperresolution_amd/ && cp -r include /tmp/isr_var_$name/
make -C /tmp/isr_var_$name/isosurfacesuperresolution_amd/csrc -j6 "COMMON=-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $*" ../lib/libisr_sr.so > /tmp/isr_var_$name/build.log 2>&1
cp /tmp/isr_var_$name/isosurfacesuperresolution_amd/lib/libisr_sr.so tools/lib_head/libisr_sr_$name.so
ls -la tools/lib_head/libisr_sr_$name.so
