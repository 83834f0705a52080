#!/bin/bash
# Builds barbell_amd/libbarbell_amd.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -o ../libbarbell_amd.so barbell_amd.hip "$@"
