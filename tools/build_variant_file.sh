#!/bin/bash
# usage: tools/build_variant_file.sh <name> <source stem, e.g. m3g_node> <extra hipcc flags...>  -> torch-m3gnet_amd/lib/variants/<name>.so
# (one translation unit rebuilt with the flags, everything else from build/)
set -e
cd "$(dirname "$0")/../torch-m3gnet_amd"
name=$1; stem=$2; shift; shift
mkdir -p lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -fPIC -I../include "$@" -c csrc/$stem.hip -o /tmp/${stem}_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/variants/$name.so /tmp/${stem}_$name.o $(ls build/*.o | grep -v "/$stem.o")
