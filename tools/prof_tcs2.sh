# PMC passes for the memory pipeline of one TCS layer: bash tools/prof_tcs2.sh CIN COUT K RES
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE"
P2="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
P3="TD_TD_BUSY_sum TD_TC_STALL_sum"
P4="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
P5="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  # a counter set the hardware cannot collect makes rocprofv3 abort and then hang in its finaliser: bound every pass
  timeout -k 5 150 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmcb$i -- python3 tools/bench_one.py $1 $2 $3 $4 3 > gpurun_out/pmcb$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for i in (1,2,3,4,5):
    files = glob.glob(f"gpurun_out/pmcb{i}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])] += 1
    for k, d in agg.items():
        if "tcs_" not in k: continue
        print(k)
        for c, v in d.items():
            print(f"   {c:40s} {v / cnt[(k,c)]:16.1f}  (per dispatch, {cnt[(k,c)]} dispatches)")
PY
