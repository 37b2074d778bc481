#!/bin/bash
# Debug build: the product library's objects + the layer-persistent decode-block experiment (tools/debug/chain/chain.hip,
# mi_op_chain) -> mlx_parallm_amd/csrc/alt/libmi355_chain.so.  The experiment is NOT part of libmi355_decode.so (DESIGN 8a:
# built, measured 24 % slower than the four launches, not used by the engine).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/mlx_parallm_amd/csrc"
make -s
mkdir -p alt
OBJS=$(make -s -pn | sed -n 's/^OBJS = //p' | head -1)
[ -n "$OBJS" ] || OBJS=$(ls *.o)
for f in chain chain_api; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -c "$ROOT/tools/debug/chain/$f.hip" -o alt/$f.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o) alt/chain.o alt/chain_api.o -o alt/libmi355_chain.so
echo built alt/libmi355_chain.so
