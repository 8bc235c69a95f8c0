# Per-round profiles (run on the GPU box through gpurun):  bash tools/prof_round.sh <round number>
#   bench.py (C2 headline): rocprofv3 --kernel-trace --stats, then separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   C3 / C4 phase 1 / C4 phase 2 (bf16 + hipGraph) / C4 phase 2 fp32 (the f32 matrix-core GEMM) / C5 / wav2vec2 fine-tuning (f32 and mixed precision, one process each): rocprofv3 --kernel-trace --stats of tools/bench_extra.py, one configuration per process
# TS_PROF_MARK=1 brackets every timed loop with a marker kernel: the summary reports the TIMED REGION ONLY (per-step shares).
# Every pass is bounded; the program itself follows `--` (no env / shell hop under the profiler).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TS_PROF_MARK=1
R=${1:-5}
O=gpurun_out/prof_r$R
rm -rf $O && mkdir -p $O
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline --no-trained-check --no-predict-api > $O/bench.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-trained-check --no-predict-api > $O/fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-trained-check --no-predict-api > $O/write.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 tools/bench_extra.py c3 --no-check > $O/c3.log 2>&1
export TS_C4_ONLY=c4_phase1
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4p1 -- python3 tools/bench_extra.py c4 > $O/c4p1.log 2>&1
export TS_C4_ONLY=c4_phase2
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4p2 -- python3 tools/bench_extra.py c4 > $O/c4p2.log 2>&1
export TS_C4_ONLY=c4_phase2_fp32
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4p2f -- python3 tools/bench_extra.py c4 > $O/c4p2f.log 2>&1
unset TS_C4_ONLY
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 tools/bench_extra.py c5 --no-check > $O/c5.log 2>&1
export TS_C5FT_ONLY=fp32
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5ft -- python3 tools/bench_extra.py c5_finetune > $O/c5ft.log 2>&1
export TS_C5FT_ONLY=bf16
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5ftb -- python3 tools/bench_extra.py c5_finetune > $O/c5ftb.log 2>&1
unset TS_C5FT_ONLY
python3 tools/prof_round_summary.py $O $R
