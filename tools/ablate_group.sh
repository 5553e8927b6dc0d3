#!/bin/bash
# tools/ablate_group.sh — development: builds of the library whose group kernel leaves something OUT (GCN_ABLATE bits, see
# spmm_group.hip: 1 no partial-row stores, 2 no row-end handling, 4 no stream loads) into artifacts/ablate/ (git-ignored,
# travels to the GPU box), to be run as  GCN_AMD_LIB=artifacts/ablate/libgcnspmm_ablN.so python bench.py ...
# The numbers these builds return are WRONG on purpose; what they cost is exact.  ABLATE_DEFS="-DGCN_STORE_POLICY=0" adds defines
# (store policy of the partial rows: 0 plain, 1 sc1, 2 nt — right results) and ABLATE_TAG names the output.
set -e
cd "$(dirname "$0")/.."
python3 -m gcn_amd.build > /dev/null
mkdir -p artifacts/ablate
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fvisibility=default -Wall -Wno-unused-result -DGCN_ABLATE=$n $ABLATE_DEFS \
    -x hip -c gcn_amd/csrc/spmm_group.hip -o artifacts/ablate/spmm_group_abl$n$ABLATE_TAG.o
  objs=$(ls gcn_amd/lib/obj/*.o | grep -v spmm_group.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs artifacts/ablate/spmm_group_abl$n$ABLATE_TAG.o -o artifacts/ablate/libgcnspmm_abl$n$ABLATE_TAG.so
  echo artifacts/ablate/libgcnspmm_abl$n$ABLATE_TAG.so
done
