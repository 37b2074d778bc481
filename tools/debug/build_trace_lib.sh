#!/bin/bash
# Debug build of the library with per-workgroup time stamps in gemm_skinny (read by tools/debug/skinny_trace.py).
set -e
cd "$(dirname "$0")/../../mlx_parallm_amd/csrc"
make -s
mkdir -p alt
# (MfmaParams / SkinnyParams grow a trace pointer: every unit that sees them is rebuilt with the macro)
for f in gemm_skinny gemv_mfma attn_decode; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMI_SK_TRACE $TRACE_FLAGS -c $f.hip -o alt/${f}_trace.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v -e '^gemv_mfma\.o$' -e '^gemm_skinny\.o$' -e '^attn_decode\.o$') alt/gemv_mfma_trace.o alt/gemm_skinny_trace.o alt/attn_decode_trace.o -o alt/libmi355_trace.so
echo built alt/libmi355_trace.so
