#!/bin/bash
# A variant build of the library for A/B runs: lastz_amd/liblzgpu_<tag>.so (picked up with LZGPU_LIB, tools/ab_lib.sh).
#   bash tools/build_variant.sh <tag> [tools/experiments/x.patch ...] [-- -DSOMETHING ...]
# The patches (measured-and-lost variants and timing-only what-ifs: they do not live in the product sources) are applied to a
# scratch copy of the sources under lastz_amd/build_<tag>/root/ (same layout: lastz_amd/csrc + include); flags behind "--" go to hipcc.
set -eu
cd "$(dirname "$0")/.."
TAG=$1; shift
R=lastz_amd/build_$TAG/root
rm -rf "$R"; mkdir -p "$R/lastz_amd/csrc" "$R/include"
cp lastz_amd/csrc/* "$R/lastz_amd/csrc/"; cp include/lzgpu.h "$R/include/"
FLAGS=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; FLAGS="$*"; break; fi
  patch -s -p1 -d "$R" < "$1"; shift
done
LZGPU_BUILD_TAG=$TAG LZGPU_CSRC=$PWD/$R/lastz_amd/csrc LZGPU_CXXFLAGS="$FLAGS" python -m lastz_amd.build | tail -1
