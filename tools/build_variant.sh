#!/bin/bash
# usage: tools/build_variant.sh <name> [hipcc flags ...]  ->  build_alt/libkg_<name>.so (an experimental build of the library,
# selected at run time with KG_LIB_PATH; build_alt/ is git-ignored but travels to the GPU box)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_alt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" -o build_alt/libkg_$name.so \
    kmergutsjava_amd/csrc/kmerguts_hip.hip -lz -lpthread
echo build_alt/libkg_$name.so
