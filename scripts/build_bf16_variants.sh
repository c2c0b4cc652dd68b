#!/bin/bash
# measurement builds of gemm_bf16.hip only (the other objects are the product library's): lib/big_abl<mask>/libgvqa_hip.so
set -e
cd "$(dirname "$0")/.."
L=graphvqa_amd/lib
# arguments: name=DEFINE[,DEFINE...]   e.g.  abl1=GVQA_BIG_ABL=1  skew=GVQA_BIG_SKEW=1,GVQA_BIG_SADDR=0
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  mkdir -p $L/big_$name
  D=""; for d in ${defs//,/ }; do D="$D -D$d"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude $D -c graphvqa_amd/csrc/gemm_bf16.hip -o $L/big_$name/gemm_bf16.o 2>/dev/null
  objs=$(ls $L/*.o | grep -v gemm_bf16.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $L/big_$name/gemm_bf16.o -ldl -Wl,--version-script=graphvqa_amd/csrc/exports.map -o $L/big_$name/libgvqa_hip.so
  echo built $L/big_$name/libgvqa_hip.so
done
