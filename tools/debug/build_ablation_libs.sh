#!/bin/bash
# Timing-only ablation builds of gemm_skinny.hip (the guide's diagnostic loop, step 2: "ablate"): each build drops one
# part of the int4 loop -- results are WRONG on purpose, only the launch time of the kernel is read.
#   tools/debug/build_ablation_libs.sh        (in the build container; the .so files travel with the snapshot)
set -e
cd "$(dirname "$0")/../../mlx_parallm_amd/csrc"
make -s
mkdir -p alt
for A in NOFMA NOUNPACK NOSTAGE NOSX "NOFMA -DMI_ABL_NOUNPACK -DMI_ABL_NOSTAGE -DMI_ABL_NOSX"; do
  tag=$(echo "$A" | tr -d ' ' | sed 's/-DMI_ABL_/_/g' | tr 'A-Z' 'a-z')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMI_ABL_$A -c gemm_skinny.hip -o alt/gemm_skinny_abl_$tag.o &
done
wait
for f in alt/gemm_skinny_abl_*.o; do
  tag=$(basename $f .o | sed 's/gemm_skinny_abl_//')
  # every product object except gemm_skinny.o (the list follows the Makefile's sources, whatever they are this round)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v '^gemm_skinny\.o$') $f -o alt/libmi355_abl_$tag.so
done
ls -la alt/*.so
