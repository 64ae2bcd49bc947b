#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   -> torch-m3gnet_amd/lib/variants/<name>.so
# SRC=<dir with the csrc/*.hip,*.h to compile> (default: csrc) builds the variant from another copy of the sources, e.g. a
# `git worktree` of an earlier commit, for same-box A/B timing.
# (the MFMA edge kernels m3g_edge_mfma.hip and m3g_edge_rev_f32.hip rebuilt with the flags, everything else from build/)
set -e
cd "$(dirname "$0")/../torch-m3gnet_amd"
name=$1; shift
src=${SRC:-csrc}
mkdir -p lib/variants
for f in m3g_edge_mfma m3g_edge_rev_f32; do
  /opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -I../include "$@" -Rpass-analysis=kernel-resource-usage -c $src/$f.hip -o /tmp/${f}_$name.o 2>&1 | grep -A8 "k_edge_rev_fusedILi3ELb1\|k_edge_block_mfmaILi3ELb0ELb0\|k_edge_rev_f32ILi3ELb1" | grep -E "Name|VGPRs:|ScratchSize" | sed "s/.*remark: /  /" | cut -c1-100 &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/variants/$name.so /tmp/m3g_edge_mfma_$name.o /tmp/m3g_edge_rev_f32_$name.o $(ls build/*.o | grep -v "m3g_edge_mfma.o\|m3g_edge_rev_f32.o")
