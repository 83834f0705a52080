#!/bin/bash
# Builds barbell_amd/libbarbell_amd.so for gfx950 (cross-compiles without a GPU) and the host CLI: see the Makefile next to this file.
set -e
cd "$(dirname "$0")"
exec make -j"$(nproc)" "$@"
