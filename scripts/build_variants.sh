#!/bin/bash
# A/B builds of libnudf.so: recompile ONE source with extra -D flags and link with the other objects of the normal build.
#   scripts/build_variants.sh <tag> <source.hip> <flags...>   ->  neuraludf_amd/build/libnudf_<tag>.so   (use with NUDF_LIB=...)
set -e
cd "$(dirname "$0")/.."
python -m neuraludf_amd.build > /dev/null
tag=$1; src=$2; shift; shift
B=neuraludf_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c neuraludf_amd/csrc/$src -o $B/$src.$tag.o -I neuraludf_amd/csrc -I include -Wno-unused-result "$@"
objs=$(ls $B/*.hip.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libnudf_$tag.so $objs $B/$src.$tag.o
echo $B/libnudf_$tag.so
