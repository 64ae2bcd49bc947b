#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   -> torch-m3gnet_amd/lib/variants/<name>.so
# SRC=<dir with the csrc/*.hip,*.h to compile> (default: csrc) builds the variant from another copy of the sources, e.g. a
# `git worktree` of an earlier commit, for same-box A/B timing.
# (FILES="m3g_node ..." names the sources rebuilt with the flags -- default: the MFMA edge kernels m3g_edge_mfma.hip and
# m3g_edge_rev_f32.hip --, everything else comes from build/)
set -e
cd "$(dirname "$0")/../torch-m3gnet_amd"
name=$1; shift
src=${SRC:-csrc}
mkdir -p lib/variants
files=${FILES:-"m3g_edge_mfma m3g_edge_rev_f32"}
for f in $files; do
  /opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -I../include "$@" -Rpass-analysis=kernel-resource-usage -c $src/$f.hip -o /tmp/${f}_$name.o 2>&1 | grep -A8 "k_edge_rev_fusedILi3ELb1\|k_edge_block_mfmaILi3ELb0ELb0\|k_edge_rev_f32ILi3ELb1" | grep -E "Name|VGPRs:|ScratchSize" | sed "s/.*remark: /  /" | cut -c1-100 &
done
wait
objs=""; skip="NONE"
for f in $files; do objs="$objs /tmp/${f}_$name.o"; skip="$skip\|$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/variants/$name.so $objs $(ls build/*.o | grep -v "$skip")
