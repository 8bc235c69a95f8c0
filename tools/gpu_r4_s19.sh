cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA --output-format csv -d gpurun_out/s19_pmc -- python3 tools/diag/pw_tile_bench.py > gpurun_out/s19.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/s19_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:50], r["Grid_Size"] if "Grid_Size" in r else "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "tcs_kernel" not in k[0]: continue
    n = cnt[(k, "SQ_WAVE_CYCLES")]
    wc = d["SQ_WAVE_CYCLES"] / n
    print(k, "dispatches", n, " ".join(f"{c}={v / n / wc:.3f}" for c, v in d.items() if c != "SQ_WAVE_CYCLES"), f"wave_cycles={wc:.3g}", f"mfma={d['SQ_INSTS_MFMA']/n:.3g}")
PY
