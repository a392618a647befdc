#!/bin/bash
# build libnudf.so here (hipcc cross-compiles), then run the given script on the MI355X box: never ship a stale .so
python -m neuraludf_amd.build > /dev/null || { echo "build failed"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-1500} -- "bash $1"
