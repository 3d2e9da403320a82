#!/bin/bash
# ORACLE — builds oracle/_ref/hevc_hls_ref from the reference's own HEVC high-level-syntax parser sources, compiled
# where they lie under /root/reference (nothing is copied; outputs only under oracle/_ref/, which is git-ignored).
# The reference's build system is NOT run: these seven files compile directly with g++ and need no generated headers.
# Not buildable this way (need cmake-generated PCCConfig.h or libav*/x265): PccLibBitstreamCommon, PccLibTranscoder.
set -e
REF=${REF_ROOT:-/root/reference}/dependencies/PccLibHevcParser
HERE=$(cd "$(dirname "$0")" && pwd)
[ -d "$REF" ] || { echo "reference not present at $REF: skipping oracle/_ref build"; exit 0; }
mkdir -p "$HERE/_ref/obj"
for f in "$REF"/source/*.cpp; do
  o="$HERE/_ref/obj/$(basename "$f" .cpp).o"
  [ "$o" -nt "$f" ] || g++ -std=c++14 -O1 -w -I"$REF/include" -c "$f" -o "$o"
done
g++ -std=c++14 -O1 -w -I"$REF/include" -I"$REF/source" "$HERE/ref_harness.cpp" $(ls "$HERE"/_ref/obj/*.o | grep -v -e PccHevcDebug.o -e PccHevcTComRom.o) -o "$HERE/_ref/hevc_hls_ref"
echo "built $HERE/_ref/hevc_hls_ref"
