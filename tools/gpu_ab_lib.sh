#!/bin/bash
# variant libraries are built by hand into tools/exp/lib<name>.so (not kept in the tree)
# A/B: default library vs tools/exp/lib$1.so
cp genmap_amd/lib/libgenmap_amd.so /tmp/lib_default.so
echo "== default"; bash tools/gpu_abc.sh
for v in "$@"; do cp tools/exp/lib$v.so genmap_amd/lib/libgenmap_amd.so; echo "== $v"; bash tools/gpu_abc.sh; done
cp /tmp/lib_default.so genmap_amd/lib/libgenmap_amd.so
