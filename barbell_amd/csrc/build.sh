#!/bin/bash
# Builds barbell_amd/libbarbell_amd.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -o ../libbarbell_amd.so barbell_amd.hip bb_trim.hip bb_fastq.hip bb_format.hip "$@"
# host side above the C-ABI (C++ mirror of the reference's annotate interface) + CLI
mkdir -p ../bin
g++ -O2 -std=c++17 -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o ../bin/barbell-amd host/annotate_main.cpp host/bb_host.cpp host/bb_rccl.cpp -L.. -lbarbell_amd -L/opt/rocm/lib -lrccl -lamdhip64 -lz -lpthread -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib
