#!/bin/bash
# developer tool (build container): compile the kernel sources of a COMMIT (default HEAD) into cloops_amd/libcloops_hip_devel.so, the
# "B" slot of tools/ab_bench.sh (CLOOPS_DEVEL_LIB=1), so that the working tree (A) and the commit (B) can be
# timed on one box.  usage: bash tools/build_head_as_b.sh [commit]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=${1:-HEAD}
D=$R/build/b_src
rm -rf $D && mkdir -p $D/cloops_amd/csrc $D/include $D/obj
for f in $(git -C $R ls-tree --name-only $C cloops_amd/csrc/ include/); do git -C $R show $C:$f > $D/$f; done
for s in $D/cloops_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -c $s -o $D/obj/$(basename $s .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/obj/*.o -o $R/cloops_amd/libcloops_hip_devel.so
echo "B = $(git -C $R rev-parse --short $C) -> cloops_amd/libcloops_hip_devel.so"
