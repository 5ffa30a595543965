#!/bin/bash
# usage: tools/build_variant.sh <tag> [-DFLAG=..]...   -> ctucopy_amd/_variants/lib_<tag>.so (an engine build for A/B runs: CTU_ENGINE_LIB)
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p ctucopy_amd/_variants
C=ctucopy_amd/csrc
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -fPIC -shared -pthread "$@" $C/engine.hip $C/opts.cc $C/design.cc $C/synth.cc -o ctucopy_amd/_variants/lib_$tag.so
echo "built ctucopy_amd/_variants/lib_$tag.so"
