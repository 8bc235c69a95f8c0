# round-4 profile session: bash tools/gpu_r4_prof.sh   (through gpurun)
bash tools/prof_round.sh 4 > gpurun_out/prof_r4.log 2>&1
tail -5 gpurun_out/prof_r4.log
