#!/bin/bash
# Timing-only experiment builds of the engine: tools/build_variant.sh <name> <extra hipcc flags...>  ->  tools/<name>.so
# (run with ITX_LIB=tools/<name>.so). Not part of the product build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
objs=()
for f in iteres_amd/csrc/*.hip; do
  o=/tmp/itxvar_${name}_$(basename "$f" .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$f" -o "$o"
  objs+=("$o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/${name}.so "${objs[@]}"
echo tools/${name}.so
