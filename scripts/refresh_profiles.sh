#!/bin/bash
# On the GPU box, from the repo root: full GPU test suite, the contract bench line, the rocprofv3 kernel statistics
# of the same workloads (U-Net step; DFC-VAE step), and the PMC passes.  Outputs land in gpurun_out/refresh_<tag>/;
# copy what is to be judged into profiles/ (scripts/collect_profiles.py).
TAG=${1:-r3}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/refresh_$TAG
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log; fi
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json
python bench.py --workload joint --no-cpu-baseline > $OUT/bench_joint.json 2> $OUT/bench_joint.err
python bench.py --workload joint --d 64 --batch 8 --no-cpu-baseline > $OUT/bench_joint_d64.json 2> $OUT/bench_joint_d64.err
# round 5: the inference configurations and the d = 64 grid, each engine on its own (a joint run shares the chip between two
# streams: its per-kernel brackets time contention)
python bench.py --workload predict --no-cpu-baseline > $OUT/bench_predict.json 2> $OUT/bench_predict.err
python bench.py --workload generate --no-cpu-baseline > $OUT/bench_generate.json 2> $OUT/bench_generate.err
python bench.py --workload unet --d 64 --batch 8 --no-cpu-baseline --no-secondary --no-inference > $OUT/bench_unet_d64.json 2> $OUT/bench_unet_d64.err
python bench.py --workload vae --d 64 --batch 8 --no-cpu-baseline > $OUT/bench_vae_d64.json 2> $OUT/bench_vae_d64.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_unet -o prof -- python3 $ROOT/bench.py --steps 8 --warmup 2 --soak-seconds 0 --no-cpu-baseline --no-secondary --no-inference > $OUT/prof_unet.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_vae -o prof -- python3 $ROOT/bench.py --steps 8 --warmup 2 --workload vae > $OUT/prof_vae.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_predict -o prof -- python3 $ROOT/bench.py --steps 8 --warmup 2 --workload predict --no-cpu-baseline > $OUT/prof_predict.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_generate -o prof -- python3 $ROOT/bench.py --steps 8 --warmup 2 --workload generate --no-cpu-baseline > $OUT/prof_generate.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_unet_d64 -o prof -- python3 $ROOT/bench.py --steps 4 --warmup 2 --soak-seconds 0 --d 64 --batch 8 --no-cpu-baseline --no-secondary --no-inference > $OUT/prof_unet_d64.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_vae_d64 -o prof -- python3 $ROOT/bench.py --steps 4 --warmup 2 --d 64 --batch 8 --workload vae > $OUT/prof_vae_d64.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cd $ROOT && KSTATS=$OUT/prof_unet/prof_kernel_stats.csv bash scripts/run_pmc.sh $TAG
# one PMC pass set on the d = 64 U-Net step's dominant kernels (the same counter groups; its own output directory)
cd $ROOT && KSTATS=$OUT/prof_unet_d64/prof_kernel_stats.csv bash scripts/run_pmc.sh ${TAG}_d64 --d 64 --batch 8
bash scripts/step_trace.sh vae ${TAG}_step_trace_vae > /dev/null 2>&1
bash scripts/step_trace.sh unet ${TAG}_step_trace_unet > /dev/null 2>&1
