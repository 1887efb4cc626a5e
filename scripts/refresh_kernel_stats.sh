#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics of the contract bench's two training workloads only (the short form of
# refresh_profiles.sh, for a final tree whose kernels changed little): gpurun_out/kstats_<tag>/{unet,vae}_kernel_stats.csv
TAG=${1:-final}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_unet -o prof -- python3 $ROOT/bench.py --steps 8 --warmup 2 --soak-seconds 0 --no-cpu-baseline --no-secondary --no-inference > $OUT/prof_unet.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_vae -o prof -- python3 $ROOT/bench.py --steps 8 --warmup 2 --workload vae > $OUT/prof_vae.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cp $(find $OUT/prof_unet -name "*kernel_stats.csv" | head -1) $OUT/unet_kernel_stats.csv
cp $(find $OUT/prof_vae -name "*kernel_stats.csv" | head -1) $OUT/vae_kernel_stats.csv
head -4 $OUT/unet_kernel_stats.csv | cut -c1-160; grep -h "thin_c_fwd" $OUT/unet_kernel_stats.csv $OUT/vae_kernel_stats.csv | cut -c1-60,200-400
