#!/bin/bash
# tools/mkvariant.sh NAME TU.cpp [-Dflags...]: one translation unit recompiled with extra flags and linked with the other objects of build/ into build/libvar_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; tu=$2; shift 2
obj=build/var_${name}_$(basename ${tu%.cpp}).o
hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-signed-zeros -ffinite-math-only -DFV3LM_SPLIT_BUILD "$@" -c fv3_jedi_linearmodel_amd/csrc/$tu -o $obj
others=$(ls build/*.o | grep -v "/var_" | grep -v "/$(basename ${tu%.cpp}).o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/libvar_${name}.so $obj $others
echo build/libvar_${name}.so
