"""Copy the judged artefacts of a scripts/refresh_profiles.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
usage: python scripts/collect_profiles.py r2"""
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "refresh_" + tag)
pmc = os.path.join(root, "gpurun_out", "pmc_" + tag)
dst = os.path.join(root, "profiles")
pairs = [
    (os.path.join(src, "bench.json"), "%s_bench.json" % tag),
    (os.path.join(src, "bench_joint.json"), "%s_bench_joint.json" % tag),
    (os.path.join(src, "bench_joint_d64.json"), "%s_bench_joint_d64.json" % tag),
    (os.path.join(src, "prof_unet", "prof_kernel_stats.csv"), "%s_unet_b32_kernel_stats.csv" % tag),
    (os.path.join(src, "prof_vae", "prof_kernel_stats.csv"), "%s_vae_b32_kernel_stats.csv" % tag),
    (os.path.join(pmc, "summary.txt"), "%s_pmc_summary.txt" % tag),
    (os.path.join(pmc, "traffic.json"), "%s_pmc_traffic.json" % tag),
    # round 5: inference configurations and the d = 64 grid, each engine on its own
    (os.path.join(src, "bench_predict.json"), "%s_bench_predict.json" % tag),
    (os.path.join(src, "bench_generate.json"), "%s_bench_generate.json" % tag),
    (os.path.join(src, "bench_unet_d64.json"), "%s_d64_bench_unet.json" % tag),
    (os.path.join(src, "bench_vae_d64.json"), "%s_d64_bench_vae.json" % tag),
    (os.path.join(src, "prof_predict", "prof_kernel_stats.csv"), "%s_predict_kernel_stats.csv" % tag),
    (os.path.join(src, "prof_generate", "prof_kernel_stats.csv"), "%s_generate_kernel_stats.csv" % tag),
    (os.path.join(src, "prof_unet_d64", "prof_kernel_stats.csv"), "%s_d64_unet_b8_kernel_stats.csv" % tag),
    (os.path.join(src, "prof_vae_d64", "prof_kernel_stats.csv"), "%s_d64_vae_b8_kernel_stats.csv" % tag),
    (os.path.join(root, "gpurun_out", "pmc_%s_d64" % tag, "summary.txt"), "%s_d64_pmc_summary.txt" % tag),
    (os.path.join(root, "gpurun_out", "pmc_%s_d64" % tag, "traffic.json"), "%s_d64_pmc_traffic.json" % tag),
    (os.path.join(root, "gpurun_out", "%s_step_trace_vae.txt" % tag), "%s_step_trace_vae.txt" % tag),
    (os.path.join(root, "gpurun_out", "%s_step_trace_unet.txt" % tag), "%s_step_trace_unet.txt" % tag),
]
for a, b in pairs:
    if os.path.exists(a):
        shutil.copy(a, os.path.join(dst, b))
        print("copied", b)
    else:
        print("missing", a)
