#!/bin/bash
# usage: tools/build_variant_tb.sh <rows>  -> torch-m3gnet_amd/lib/variants/tb<rows>.so
# three-body kernels + topology rebuilt with M3G_TB_ROWS=<rows> active-edge rows per workgroup (the staged LDS window
# holds min(rows + 64, 255) rows); everything else from build/
set -e
cd "$(dirname "$0")/../torch-m3gnet_amd"
rows=$1; shift
extra="$@"   # e.g. -DM3G_TB_CAP=255 -DM3G_TB_LIST=64
name=${NAME:-tb$rows}
mkdir -p lib/variants
for f in m3g_threebody m3g_topology; do
  /opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -I../include -DM3G_TB_ROWS=$rows $extra -c csrc/$f.hip -o /tmp/${f}_$name.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/variants/$name.so /tmp/m3g_threebody_$name.o /tmp/m3g_topology_$name.o $(ls build/*.o | grep -v "m3g_threebody.o\|m3g_topology.o")
