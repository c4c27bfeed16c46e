#!/bin/bash
# round 3: the driver's exact command on a fresh process, with and without the disclosed pre-warm, per-step log kept
root=${GRAFT_REPO_ROOT:-.}
cd $root
AABR_BENCH_STEP_LOG=gpurun_out/ramp_noprewarm.json timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-prewarm --min-timed-s 0 --no-extras --no-cpu-baseline > gpurun_out/ramp_a.log 2> gpurun_out/ramp_a.err; echo "a rc=$?"
AABR_BENCH_STEP_LOG=gpurun_out/ramp_long.json timeout -k 10 300 python3 bench.py --gpus 1 --steps 300 --warmup 0 --no-prewarm --min-timed-s 0 --no-extras --no-cpu-baseline > gpurun_out/ramp_b.log 2> gpurun_out/ramp_b.err; echo "b rc=$?"
AABR_BENCH_STEP_LOG=gpurun_out/ramp_default.json timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/ramp_c.log 2> gpurun_out/ramp_c.err; echo "c rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 200 --warmup 50 --no-extras --no-cpu-baseline > gpurun_out/ramp_d.log 2> gpurun_out/ramp_d.err; echo "d rc=$?"
cat gpurun_out/ramp_a.log gpurun_out/ramp_c.log gpurun_out/ramp_d.log
