#!/bin/bash
# Debug build of the library with per-workgroup time stamps in gemm_skinny (read by tools/debug/skinny_trace.py).
set -e
cd "$(dirname "$0")/../../mlx_parallm_amd/csrc"
make -s
mkdir -p alt
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMI_SK_TRACE -c gemm_skinny.hip -o alt/gemm_skinny_trace.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 engine.o gemv_v1.o gemv_mfma.o gemm_prefill.o alt/gemm_skinny_trace.o attn.o attn_decode.o attn_prefill.o misc.o repack.o ops_api.o -o alt/libmi355_trace.so
echo built alt/libmi355_trace.so
