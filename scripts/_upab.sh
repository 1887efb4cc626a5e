for v in big256 big768 big2048 big9999; do
  echo "== $v"
  ICSG3D_LIB_PATH=$GRAFT_REPO_ROOT/icsg3d_amd/variants/libicsg3d_hip_$v.so python scripts/quick_bench.py 32 32 10 2>&1 | grep -E "^ms/step|conv_up3" | cut -c1-130
  ICSG3D_LIB_PATH=$GRAFT_REPO_ROOT/icsg3d_amd/variants/libicsg3d_hip_$v.so python scripts/quick_bench_vae.py 2>&1 | grep -E "^ms/step|^DFC|conv_up3" | cut -c1-130
done
