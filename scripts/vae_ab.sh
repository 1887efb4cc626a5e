cd $GRAFT_REPO_ROOT
for pr in none "0,1" "-1,0" "-1,1" "0,-1" "1,-1"; do
  if [ "$pr" = none ]; then e=""; else e="ICSG3D_ST_PRIO=$pr"; fi
  echo "== $e"
  env $e python scripts/quick_bench_vae.py 32 32 30 > /tmp/o.txt 2>&1; grep "ms/step" /tmp/o.txt | head -1
done
