#!/bin/bash
# Development tool: build scripts/variants/libthesia_amd_<tag>.so (outside the package directory) with extra -D flags (A/B on one GPU box:
# THESIA_AMD_LIB=scripts/variants/libthesia_amd_<tag>.so python scripts/bench_stft.py).
# usage: scripts/build_variant.sh <tag> [flags...]     (run after __graft_entry__.build())
# By default only kernels_stft.hip is recompiled; VARIANT_SOURCES="kernels_image.hip api.hip" picks others.
set -e
tag=$1; shift
cd "$(dirname "$0")/../thesia_amd/csrc"
srcs=${VARIANT_SOURCES:-kernels_stft.hip kernels_stft_w1024.hip kernels_stft_w2048.hip kernels_stft_w4096.hip}   # (kernels_stft.hip is four translation units)
obj=../../build/obj   # __graft_entry__.build()'s object cache
objs=""
for f in api.hip track_manager.hip kernels_stft.hip kernels_stft_w1024.hip kernels_stft_w2048.hip kernels_stft_w4096.hip kernels_stft_long.hip kernels_mel.hip kernels_image.hip kernels_waveform.hip host_math.cpp tile_cache.cpp; do
  if [[ " $srcs " == *" $f "* ]]; then
    extra=""
    case $f in
      kernels_stft.hip|kernels_stft_w1024.hip|kernels_stft_w2048.hip|kernels_stft_w4096.hip|kernels_stft_long.hip) extra="-fno-slp-vectorize";;
      kernels_image.hip|kernels_waveform.hip) extra="-ffp-contract=off";;
    esac
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $extra "$@" -c $f -o $obj/${f}_$tag.o &
    objs="$objs $obj/${f}_$tag.o"
  else
    objs="$objs $obj/$f.o"
  fi
done
wait
out=${VARIANT_DIR:-../../scripts/variants}; mkdir -p $out && hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libthesia_amd_$tag.so $objs
echo built $out/libthesia_amd_$tag.so
