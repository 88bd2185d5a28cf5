#!/bin/bash
# build ropebwt2_amd/lib/librb2hip_<tag>.so from the working tree with extra compiler flags (A/B experiments: tools/ab_long.sh <tag> ...)
#   usage: build_variant.sh <tag> [-DRB2_LQ=3 ...]
tag=$1; shift
cd $(dirname $0)/..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Iropebwt2_amd/csrc -Wno-unused-value "$@" -o ropebwt2_amd/lib/librb2hip_$tag.so ropebwt2_amd/csrc/rb2_engine.hip -ldl -lpthread
