#!/bin/bash
# A/B builds of libnudf.so: recompile ONE OR MORE sources (comma-separated) with extra -D flags and link with the other
# objects of the normal build.
#   scripts/build_variants.sh <tag> <a.hip[,b.hip]> <flags...>   ->  neuraludf_amd/build/libnudf_<tag>.so   (use with NUDF_LIB=...)
set -e
cd "$(dirname "$0")/.."
python -m neuraludf_amd.build > /dev/null
tag=$1; srcs=$2; shift; shift
B=neuraludf_amd/build
objs=$(ls $B/*.hip.o)
pids=""
for src in ${srcs//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c neuraludf_amd/csrc/$src -o $B/$src.$tag.o -I neuraludf_amd/csrc -I include -Wno-unused-result "$@" &
  pids="$pids $!"
  objs=$(echo "$objs" | grep -v "/$src.o")
  objs="$objs $B/$src.$tag.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libnudf_$tag.so $objs
echo $B/libnudf_$tag.so
