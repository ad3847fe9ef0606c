#!/bin/bash
# Build a variant of libmantis_hip.so with one source recompiled under extra -D flags (timing probes; results may be wrong).
#   tools/build_probe_lib.sh <source-name> <out-name> -DFLAG ...      ->  tools/_bin/libmantis_<out-name>.so
set -e
cd "$(dirname "$0")/.."
src=$1; out=$2; shift 2
python -m mantis_amd.build >/dev/null 2>&1
mkdir -p tools/_bin
extra=""
[ "$src" = attn_fwd64 ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $extra "$@" -c mantis_amd/csrc/$src.hip -o tools/_bin/${src}_${out}.o
objs=$(ls mantis_amd/csrc/_obj/*.o | grep -v "/${src}\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libmantis_${out}.so $objs tools/_bin/${src}_${out}.o
echo tools/_bin/libmantis_${out}.so
