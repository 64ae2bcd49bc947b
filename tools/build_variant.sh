#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   -> torch-m3gnet_amd/lib/variants/<name>.so (edge kernels rebuilt with the flags)
set -e
cd "$(dirname "$0")/../torch-m3gnet_amd"
name=$1; shift
mkdir -p lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -I../include "$@" -Rpass-analysis=kernel-resource-usage -c csrc/m3g_edge_mfma.hip -o /tmp/edge_$name.o 2>&1 | grep -A8 "k_edge_rev_fusedILi3ELb1\|k_edge_block_mfmaILi3ELb0ELb0\|k_edge_rev_node_mlp\|k_edge_rev_edge_mlpILi3ELb0" | grep -E "Name|VGPRs:|ScratchSize" | sed "s/.*remark: /  /" | cut -c1-90
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/variants/$name.so /tmp/edge_$name.o $(ls build/*.o | grep -v m3g_edge_mfma.o)
