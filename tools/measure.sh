#!/bin/bash
# Run on the GPU box (gpurun): the round's measurements as NAMED STEPS, each under its own timeout, each leaving only its few KB of
# summary in gpurun_out/ before the next starts (rocprofv3 databases are deleted on the box: gpurun copies back <= 64 MiB, and round 4's
# closing call lost everything to that cap and to one 40-minute limit).
#   usage: tools/measure.sh TAG step [step ...]        -> gpurun_out/TAG_*; progress in gpurun_out/TAG_progress.txt
#   steps: lib=<path to a build of the library, or "default"> | mb:<tools/microbench/NAME> | tie | parity | gputests | bench | bench_driver | bench_2rank | bench_2rank_ns (the default workload on 2 ranks: prints north_star_workload) | bench_configs | ab:<lib1>,<lib2>[:<bench flags>] | prof:<workload> | sections:<task>
#   prof workloads: walk4096 walk8192 (north_star's shard) walk262144 arm4096 mixedarm2048 mixedarm16384 gallop8192 gallop65536 turnhf4096 turnhf32768 poses4096
#   other steps: policy_cost (tools/policy_cost.py: the fused actor against the open-loop kernels on the same trajectory) | ppo_standup (train_segments, round 1's standup run) | bench_walk8192
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=gpurun_out/${TAG}_progress.txt
stamp() { echo "$(date +%s) $*" >> $P; }
declare -A WL=( [walk4096]="" [walk262144]="--envs-per-gpu 262144" [arm4096]="--mark arm" [mixedarm2048]="--mixed --mark arm --envs-per-gpu 2048"
                [gallop8192]="--task gallop --signal ol --envs-per-gpu 8192" [turnhf4096]="--task turn --terrain random" [poses4096]="--task poses"
                [walk8192]="--envs-per-gpu 8192" [gallop65536]="--task gallop --signal ol --envs-per-gpu 65536" [turnhf32768]="--task turn --terrain random --envs-per-gpu 32768"
                [mixedarm16384]="--mixed --mark arm --envs-per-gpu 16384" )
for step in "$@"; do
  stamp "start $step"
  case $step in
    tie)          timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "trace_kernels_are_bit_identical" > gpurun_out/${TAG}_tie_tests_$(basename ${REX_LIB_PATH:-default} .so).txt 2>&1 ;;
    parity)       rm -f gpurun_out/parity_windows.jsonl
                  timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "every_kernel_variant or walk_ik_trajectory_rmse" > gpurun_out/${TAG}_parity_tests.txt 2>&1 ;;
    gputests)     timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gpu_tests.txt 2>&1 ;;
    bench)        timeout 300 bash -c "python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err" ;;
    bench_driver) timeout 180 bash -c "python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2>> gpurun_out/${TAG}_bench.err" ;;
    bench_2rank)  timeout 300 bash -c "python bench.py --gpus 2 --steps 200 --warmup 50 --config 3 --backend gloo --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_gloo_1gpu.json 2>> gpurun_out/${TAG}_bench.err" ;;
    bench_2rank_ns) timeout 300 bash -c "python bench.py --gpus 2 --steps 50 --warmup 10 --backend gloo --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_north_star_gloo_1gpu.json 2>> gpurun_out/${TAG}_bench.err" ;;
    bench_configs) timeout 2400 bash -c "bash tools/bench_configs.sh ${TAG} 600 > gpurun_out/${TAG}_bench_configs.txt 2>&1" ;;
    ab:*)         IFS=: read -r _ libs flags <<< "$step"
                  OUT=gpurun_out/${TAG}_ab.txt; echo "== bench.py --steps 600 --warmup 100 $flags" >> $OUT
                  for lib in ${libs//,/ }; do
                    REX_LIB_PATH=$PWD/$lib timeout 200 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-walking-workload $flags 2>> gpurun_out/${TAG}_ab.err | tail -1 | \
                      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %-28s %8.2f M env-steps/s  %.4f ms/step  kernel %.4f ms (min %.4f)' % ('$lib'.split('/')[-1], d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min']))" >> $OUT
                  done ;;
    prof:*)       w=${step#prof:}
                  PT=150; [ "$w" = walk262144 ] && PT=420       # (the 262 144-env bench run is 1 950 launches of 1.8 ms + its pre-roll under the profiler)
                  REX_PROFILE_TIMEOUT=$PT timeout $((4 * PT + 100)) bash tools/profile_round.sh ${TAG}_$w ${WL[$w]} > gpurun_out/${TAG}_profile_$w.txt 2>&1
                  rm -rf gpurun_out/${TAG}_${w}_prof gpurun_out/${TAG}_${w}_pmc_* ;;
    policy_cost)  : > gpurun_out/${TAG}_policy_cost.jsonl
                  for args in "--envs 4096" "--envs 8192" "--envs 16384" "--envs 4096 --mark arm" "--envs 8192 --task gallop --signal ol"; do
                    timeout 200 python tools/policy_cost.py $args >> gpurun_out/${TAG}_policy_cost.jsonl 2>> gpurun_out/${TAG}_bench.err
                  done ;;
    ppo_standup)  timeout 600 python -m rex_gym_amd.agents.ppo --task standup --signal ol --envs 512 --iterations 12 --max-length 400 > gpurun_out/${TAG}_ppo_standup.txt 2>&1
                  timeout 600 python -m rex_gym_amd.agents.ppo --task standup --signal ol --envs 512 --iterations 12 --max-length 400 --loop steps > gpurun_out/${TAG}_ppo_standup_per_step_loop.txt 2>&1 ;;
    bench_walk8192) timeout 400 bash -c "python bench.py --envs-per-gpu 8192 --no-cpu-baseline > gpurun_out/${TAG}_bench_walk8192.json 2>> gpurun_out/${TAG}_bench.err" ;;
    prof_policy:*) IFS=: read -r _ pn pt <<< "$step"; pn=${pn:-4096}; pt=${pt:-25}; export TMPDIR=/tmp
                  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_polprof -- python $GRAFT_REPO_ROOT/tools/closed_loop_run.py --envs $pn --segment $pt --launches 40 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_policy_${pn}_T${pt}.txt 2>/dev/null )
                  python tools/rocprof_stats.py $(find gpurun_out/${TAG}_polprof -name "*_results.db" | head -1) --window 30:70 --csv gpurun_out/${TAG}_kernel_stats_policy_${pn}_T${pt}.csv | head -5 >> gpurun_out/${TAG}_prof_policy_${pn}_T${pt}.txt
                  rm -rf gpurun_out/${TAG}_polprof ;;
    sections:*)   t=${step#sections:}
                  timeout 400 python tools/prof_sections.py --task=$t > gpurun_out/${TAG}_sections_$t.txt 2>&1 ;;
    lib=*)        export REX_LIB_PATH=$PWD/${step#lib=}; [ "${step#lib=}" = default ] && unset REX_LIB_PATH ;;      # the steps after it load this build of the library
    mb:*)         m=${step#mb:}; export TMPDIR=/tmp
                  MBFLAGS=""; [ "$m" = policy_mb ] && MBFLAGS="-std=c++17 -ffp-contract=on -I rex_gym_amd/csrc"     # (includes the kernels' headers)
                  timeout 300 bash -c "hipcc --offload-arch=gfx950 -O3 $MBFLAGS -o /tmp/$m tools/microbench/$m.hip 2>/dev/null && /tmp/$m" > gpurun_out/${TAG}_mb_$m.txt 2>&1 ;;
    *)            echo "unknown step $step" >> $P ;;
  esac
  stamp "rc=$? $step"
  find gpurun_out -name "*.db" -size +1M -delete 2>/dev/null      # nothing heavy may stay for the pull
done
du -sh gpurun_out >> $P
cat $P
