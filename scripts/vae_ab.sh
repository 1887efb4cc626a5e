cd $GRAFT_REPO_ROOT
for k in none 30 28 26 24 20 16; do
  if [ $k = none ]; then e=""; else e="ICSG3D_PM_SIDE_CUS=$k"; fi
  echo "== $e"
  env $e python scripts/quick_bench_vae.py 32 32 30 2>&1 | sed -n 2,2p
done
