#!/bin/bash
# Experiment builds of libdd_hip.so: tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>
# recompiles ONE source with the extra flags and links it with the regular objects into tools/exp/libdd_<name>.so
# (selected at run time with DD_LIB=tools/exp/libdd_<name>.so; never the shipped library).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m deepdenoiser_amd.build > /dev/null
mkdir -p tools/exp
obj=tools/exp/${name}_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c deepdenoiser_amd/csrc/$src -o $obj
others=$(ls deepdenoiser_amd/csrc/*.o | grep -v "$(basename ${src%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libdd_${name}.so $obj $others
echo tools/exp/libdd_${name}.so
