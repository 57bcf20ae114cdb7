#!/bin/bash
# Experiment builds that differ in the decoder only: tools/build_inflate_variant.sh <name> <extra hipcc flags...>  ->  tools/<name>.so
# (itx_inflate.hip compiled with the flags, linked with the product build's other objects; run with ITX_LIB=tools/<name>.so).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m iteres_amd.build > /dev/null
o=/tmp/itxvar_${name}_itx_inflate.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c iteres_amd/csrc/itx_inflate.hip -o "$o"
objs=()
for f in iteres_amd/csrc/*.o; do [ "$(basename $f)" = itx_inflate.o ] || objs+=("$f"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/${name}.so "$o" "${objs[@]}"
echo tools/${name}.so
