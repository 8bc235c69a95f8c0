cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT

P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU"
P3="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P4="SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc$i -- python3 tools/bench_one.py $1 $2 $3 $4 3 > gpurun_out/pmc$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for i in (1,2,3,4):
    files = glob.glob(f"gpurun_out/pmc{i}/**/*counter_collection.csv", recursive=True)
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
            print(f"   {c:36s} {v / cnt[(k,c)]:16.1f}  (per dispatch, {cnt[(k,c)]} dispatches)")
PY
