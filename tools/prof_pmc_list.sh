#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
timeout 120 rocprofv3 -L > gpurun_out/pmc/counters.txt 2>&1
grep -o -E "\b(SQ|GRBM|TCC|TCP|TA|TD|SPI)_[A-Z0-9_]+" gpurun_out/pmc/counters.txt | sort -u > gpurun_out/pmc/names.txt
wc -l gpurun_out/pmc/names.txt
grep -E "MFMA|LDS|SQ_BUSY|SQ_WAVE_CYCLES|SQ_WAIT|SQ_ACTIVE|GUI_ACTIVE|SQ_WAVES$|SQ_INSTS_VALU$|SQ_INSTS_VMEM|SQ_INSTS_SALU" gpurun_out/pmc/names.txt | tr '\n' ' '
