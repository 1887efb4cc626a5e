#!/bin/bash
# bisect the e0 / e1 gradient error of the DFC-VAE step at B = 32, d = 32 over the engine's fallback switches
mkdir -p gpurun_out
for sw in NONE ICSG3D_NO_FAST_BNBWD ICSG3D_NO_VAE_SIDE_WGRAD ICSG3D_NO_FWD_SPLITK ICSG3D_NO_PM_SIDE ICSG3D_NO_THIN_N ICSG3D_NO_COND_FOLD ICSG3D_NO_WINOG; do
  echo "== $sw"
  env $sw=1 python -m pytest "tests/test_gpu_fullsize_oracle.py::test_vae_step_at_stated_batch_matches_pinned_fp64_oracle[32-32]" -q -s 2>&1 \
    | grep -E "per-tensor|passed|failed" | sed -E "s/'(d[0-9]|dout|dec_dense|z_|enc_dense|e4|e3)[^,]*, //g" | cut -c1-700
done > gpurun_out/r6_bisect_e1.txt 2>&1
cat gpurun_out/r6_bisect_e1.txt
timeout 600 python scripts/prof_refine.py 32 > gpurun_out/r6_prof_refine2.txt 2>&1
head -30 gpurun_out/r6_prof_refine2.txt; tail -3 gpurun_out/r6_prof_refine2.txt
