#!/bin/bash
# A/B builds of the HIP library: tools/build_variant.sh <name> "<extra hipcc flags>" [file.hip ...]
# The listed sources (default: all) are compiled with the extra flags, the others reuse the product objects;
# result: deepfm_amd/lib/variants/lib_<name>.so (git-ignored; select with DFM_LIB_PATH=...).
set -e
cd "$(dirname "$0")/../deepfm_amd/csrc"
name=$1; flags=$2; shift 2
files=("$@")
make -j8 >/dev/null
out=../lib/variants; mkdir -p $out/obj_$name
objs=()
for src in *.hip; do
  o=../lib/obj/${src%.hip}.o
  for f in "${files[@]}"; do
    if [ "$f" == "$src" ]; then
      o=$out/obj_$name/${src%.hip}.o
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c $src -o $o &
    fi
  done
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so "${objs[@]}"
rm -f $out/lib_$name.so.*
echo "built $out/lib_$name.so"
