#!/bin/bash
# round 5, session 1: new kernels' tests + the trained-weights transcript study
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s1
timeout 900 python -m pytest tests/test_gpu_r5.py tests/test_gpu_w2v_train.py tests/test_gpu_citrinet.py tests/test_gpu_r2.py -x -q > gpurun_out/r5s1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5s1/pytest.log
tail -5 gpurun_out/r5s1/pytest.log
for steps in 600 1500; do
  timeout 900 python tools/train_margin_model.py --steps $steps --out gpurun_out/r5s1/qn_tones_$steps.pt > gpurun_out/r5s1/train_$steps.log 2>&1
  echo "train $steps rc=$?"
  tail -c 3000 gpurun_out/r5s1/train_$steps.log
done
rm -f gpurun_out/r5s1/*.pt
