# Round-2 profiles (run on the GPU box through gpurun):  bash tools/prof_round2.sh
#   bench.py (C2 headline): rocprofv3 --kernel-trace --stats, then separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   C3 / C4 phase 1 / C4 phase 2 (bf16 + hipGraph) / C5: rocprofv3 --kernel-trace --stats of the per-config bench tools
# Every pass is bounded; the program itself follows `--` (no env / shell hop under the profiler).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r2
rm -rf $O && mkdir -p $O
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 200 --warmup 5 --no-extra --no-cpu-baseline > $O/bench.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/write.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 tools/bench_c3.py > $O/c3.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4p1 -- python3 tools/bench_finetune.py --steps 20 > $O/c4p1.log 2>&1
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4p2 -- python3 tools/bench_finetune.py --unfreeze --gemm-bf16 --graph --steps 20 > $O/c4p2.log 2>&1
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 tools/bench_c5.py > $O/c5.log 2>&1
python3 tools/prof_round2_summary.py $O
