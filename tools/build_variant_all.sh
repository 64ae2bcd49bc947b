#!/bin/bash
# usage: tools/build_variant_all.sh <name> <extra hipcc flags...>  -> lib/variants/<name>.so, every csrc file rebuilt with the flags
# (the MFMA edge kernels take ~1.5 min; pass M3G_SKIP_EDGE=1 in the environment to reuse build/m3g_edge_mfma.o)
set -e
cd "$(dirname "$0")/../torch-m3gnet_amd"
name=$1; shift
mkdir -p lib/variants /tmp/var_$name
objs=""
for f in csrc/*.hip; do
  stem=$(basename $f .hip)
  if [ "$stem" = m3g_edge_mfma ] && [ -n "$M3G_SKIP_EDGE" ]; then objs="$objs build/$stem.o"; continue; fi
  /opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -I../include "$@" -c $f -o /tmp/var_$name/$stem.o &
  objs="$objs /tmp/var_$name/$stem.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/variants/$name.so $objs
