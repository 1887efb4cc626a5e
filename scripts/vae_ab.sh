cd $GRAFT_REPO_ROOT
for env in "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 ICSG3D_NO_PM_SIDE=1 ICSG3D_NO_VAE_SIDE_WGRAD=1" "GPU_MAX_HW_QUEUES=4 ICSG3D_NO_PM_SIDE=1 ICSG3D_NO_VAE_SIDE_WGRAD=1" "GPU_MAX_HW_QUEUES=16"; do
  echo "== env: $env"
  env $env python bench.py --workload joint --no-cpu-baseline 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('joint', o['ms_per_step'], o['value'])"
done
