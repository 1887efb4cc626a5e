for v in base a16 a32 a64 a4 a2 a1 a7; do
  echo "== $v"
  ICSG3D_LIB_PATH=$GRAFT_REPO_ROOT/icsg3d_amd/variants/libicsg3d_hip_$v.so WINO_MODES=2 python scripts/wino_bench.py 2>&1 | grep -E "c18|c16|sum" | cut -c1-110
done
