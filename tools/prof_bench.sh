# Profiles of the default bench workload (run on the GPU box through gpurun):  bash tools/prof_bench.sh
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 2`
#   2./3. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (the TCC block cannot hold both at once)
# Every pass is bounded: a counter set the hardware cannot collect makes rocprofv3 abort and hang in its finaliser.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_bench
rm -rf $O && mkdir -p $O
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
tail -1 $O/trace.log
python3 - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/prof_bench"
out = {}
for name in ("fetch", "write"):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            agg[k] += float(row["Counter_Value"]); cnt[k] += 1
    out[name] = {k: {"dispatches": cnt[k], "sum": agg[k]} for k in agg}
json.dump(out, open(f"{O}/pmc_raw.json", "w"), indent=1)
for name in out:
    tot = sum(v["sum"] for k, v in out[name].items() if "tcs_" in k); n = sum(v["dispatches"] for k, v in out[name].items() if "tcs_" in k)
    print(name, "tcs kernels: dispatches", n, "counter sum", tot, "per dispatch", tot / max(n, 1))
PY
ls $O/trace/*/ | head
