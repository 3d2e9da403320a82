#!/bin/bash
# Measurement builds (here, no GPU needed): the library with one phase of a CTB chain LEFT OUT (csrc/rbt_platform.h RBT_ABLATE), one build per mask in "$@",
# into rabbit-transcoding_amd/ablate/librbt_a<mask>.so (git-ignored; they travel to the GPU box). tools/ablate_run.sh runs a counter pass with each of them.
cd "$(dirname "$0")/../rabbit-transcoding_amd" || exit 1
mkdir -p ablate
HOST="host/rbt_hls.cpp host/rbt_decode.cpp host/rbt_transcode.cpp host/rbt_pcc.cpp host/rbt_api.cpp host/rbt_v3c.cpp"
[ -f csrc/rbt_kernels_parse.o ] || make librbt.so || exit 1
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Wall -Wno-unused-function -Os -DRBT_ABLATE=$1 -c csrc/rbt_kernels.hip -o ablate/k_$1.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Wall -Wno-unused-function -pthread -shared -o ablate/librbt_a$1.so ablate/k_$1.o csrc/rbt_kernels_parse.o $HOST && rm -f ablate/k_$1.o
}
n=0
for m in "$@"; do build $m & n=$((n + 1)); if [ $n -ge 4 ]; then wait; n=0; fi; done
wait
ls -la ablate
