#!/bin/bash
# Build the C-ABI HIP library for gfx950 (MI355X) in-tree.  Usage: ./build.sh
set -e
cd "$(dirname "$0")/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
mkdir -p ../build
if [ -n "$FW_CLEAN" ]; then rm -f ../build/*.o ../libfwair_hip.so; fi       # __graft_entry__.build(): always prove a clean compile
pids=()
for f in fw_gemm fw_norm fw_attn fw_elem fw_heads fw_conv fw_vit fw_leff fw_gattn fw_data; do
  if [ ! -f ../build/$f.o ] || [ $f.hip -nt ../build/$f.o ] || [ fw_common.h -nt ../build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o ../build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC ../build/fw_gemm.o ../build/fw_norm.o ../build/fw_attn.o ../build/fw_elem.o ../build/fw_heads.o ../build/fw_conv.o ../build/fw_vit.o ../build/fw_leff.o ../build/fw_gattn.o ../build/fw_data.o -o ../libfwair_hip.so
# the package carries its own copy of the C-ABI header (fwair/lib.py builds the ctypes prototypes from it at import time)
if [ -f ../../include/fwair.h ]; then cp ../../include/fwair.h ../fwair/fwair.h; fi
echo "built $(cd ..; pwd)/libfwair_hip.so"
