#!/bin/bash
# Build variants of ONE kernel file into separate copies of the library (kernel A/B experiments on the GPU box):
#   scripts/variants.sh conv_wino64 "name1:-DX=1" "name2:-DX=2 -DY=3" ...
# -> icsg3d_amd/variants/libicsg3d_hip_<name>.so; select with ICSG3D_LIB_PATH=...; prints register use / spills.
set -e
cd "$(dirname "$0")/../icsg3d_amd/csrc"
src=$1; shift
make -j4 >/dev/null
mkdir -p ../variants build/var
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include $flags \
      -Rpass-analysis=kernel-resource-usage -c $src.hip -o build/var/${src}_$name.o 2> build/var/${src}_$name.txt
  objs=""
  for o in build/*.o; do b=$(basename $o .o); if [ "$b" == "$src" ]; then objs="$objs build/var/${src}_$name.o"; else objs="$objs $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../variants/libicsg3d_hip_$name.so -L/opt/rocm/lib -lrccl -lroctx64 -Wl,-rpath,/opt/rocm/lib
  echo "== $name ($flags)"; python3 ../../scripts/kernel_resources.py build/var/${src}_$name.txt | sed 's/ \+/ /g' | sed "s/'AGPRs': 0, //; s/'Occupancy': 2, //"
done
