#!/bin/bash
# Development tool: build thesia_amd/libthesia_amd_<tag>.so with extra flags for kernels_stft.hip
# (A/B on one GPU box: THESIA_AMD_LIB=thesia_amd/libthesia_amd_<tag>.so python scripts/bench_stft.py).
# usage: scripts/build_variant.sh <tag> [flags...]     (run after __graft_entry__.build())
set -e
tag=$1; shift
cd "$(dirname "$0")/../thesia_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize "$@" -c kernels_stft.hip -o build/kernels_stft_$tag.o
objs=$(ls build/*.hip.o build/*.cpp.o | grep -v kernels_stft.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libthesia_amd_$tag.so $objs build/kernels_stft_$tag.o
echo built thesia_amd/libthesia_amd_$tag.so
