#!/bin/bash
# SQ counter passes over the C2 headline step (rocprofv3 --pmc only: no trace domain beside it), for the TCS launches:
#   bash tools/prof_pmc_tcs.sh <round>
# Three passes of <= 8 SQ counters (+ GRBM_GUI_ACTIVE, its own block); tools/prof_pmc_summary.py condenses them per kernel instantiation.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-5}
O=gpurun_out/pmc_r$R
rm -rf $O && mkdir -p $O
ARGS="bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-trained-check --no-predict-api"
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $ARGS > $O/p1.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $ARGS > $O/p2.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_VALU --output-format csv -d $O/p3 -- python3 $ARGS > $O/p3.log 2>&1
python3 tools/prof_pmc_summary.py $O $R
