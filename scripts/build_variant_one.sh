#!/bin/bash
# measurement build of ONE translation unit (the other objects are the product library's): scripts/build_variant_one.sh <file.hip> <name> [DEFINE...]
# -> graphvqa_amd/lib/<name>/libgvqa_hip.so (loaded through GVQA_LIB by scripts/; never by the product path)
set -e
cd "$(dirname "$0")/.."
L=graphvqa_amd/lib; f=$1; name=$2; shift 2
mkdir -p $L/$name
D=""; for d in "$@"; do D="$D -D$d"; done
extra=""; [ "$f" = hopagg.hip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $extra -Iinclude $D -c graphvqa_amd/csrc/$f -o $L/$name/${f%.hip}.o 2>/dev/null
objs=$(ls $L/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $L/$name/${f%.hip}.o -ldl -Wl,--version-script=graphvqa_amd/csrc/exports.map -o $L/$name/libgvqa_hip.so
echo built $L/$name/libgvqa_hip.so
