#!/bin/bash
# usage: tools/stamp_variants.sh name:waves ...   (stamp_report.py with lib/variants/<name>.so swapped in; "base:16" = the built library)
cd "$(dirname "$0")/.."
L=torch-m3gnet_amd/lib
cp $L/libm3gnet_hip.so /tmp/base.so
for spec in "$@"; do
  v=${spec%%:*}; w=${spec##*:}
  if [ "$v" = base ]; then cp /tmp/base.so $L/libm3gnet_hip.so; else cp $L/variants/$v.so $L/libm3gnet_hip.so; fi
  STAMP_WAVES=$w timeout -k 10 200 python tools/stamp_report.py ${PREC:-fp32} > gpurun_out/stamps_$v.txt 2>&1 || echo "$v FAILED"
done
cp /tmp/base.so $L/libm3gnet_hip.so
