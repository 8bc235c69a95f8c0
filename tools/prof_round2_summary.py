"""Condense the rocprofv3 outputs of tools/prof_round2.sh into the files that go under profiles/ (round2_*)."""
import collections, csv, glob, json, os, shutil, sys

O = sys.argv[1]
out = os.path.join(O, "profiles")
os.makedirs(out, exist_ok=True)


def stats(name):
    fs = sorted(glob.glob(f"{O}/{name}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    if not fs:
        return None
    shutil.copy(fs[-1], os.path.join(out, f"round2_{name}_kernel_stats.csv"))
    return list(csv.DictReader(open(fs[-1])))


def last_line(name, prefix=None):
    try:
        lines = [l.strip() for l in open(f"{O}/{name}.log") if l.strip() and "amdgpu" not in l]
    except OSError:
        return ""
    if prefix:
        lines = [l for l in lines if l.startswith(prefix)] or lines
    return lines[-1] if lines else ""


md = ["# Round 2 -- rocprofv3 summaries (tools/prof_round2.sh on one MI355X)", ""]


def table(rows, top=16, per=None):
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    t = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        t.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {100*int(r['TotalDurationNs'])/max(tot,1):.1f} |")
    return t, tot


rows = stats("bench")
if rows:
    line = last_line("bench", "{")
    try:
        bj = json.loads(line)
        json.dump(bj, open(os.path.join(out, "round2_bench.json"), "w"), indent=1)
        md += [f"## C2 headline: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 5 --no-extra --no-cpu-baseline`", "",
               f"bench line of the profiled run: value {bj['value']:.0f} {bj['unit']}, {bj['ms_per_step']:.3f} ms/step, encoder {bj['roofline']['encoder_ms']:.3f} ms, "
               f"avg_launch_us {bj['roofline']['avg_launch_us']:.1f} (HIP events), roofline.frac {bj['roofline']['frac']:.3f}", ""]
    except Exception:
        md += ["## C2 headline", "", "bench line: " + line[:300], ""]
    t, tot = table(rows)
    md += t
    tcs = [r for r in rows if "tcs_" in r["Name"]]
    n = sum(int(r["Calls"]) for r in tcs); ns = sum(int(r["TotalDurationNs"]) for r in tcs)
    md += ["", f"All `ts::tcs_*` kernels: {n} calls, {ns/1e6:.2f} ms, average {ns/max(n,1)/1e3:.1f} us per launch.", ""]

pmc = {}
for name in ("fetch", "write"):
    agg, cnt = 0.0, 0
    for f in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "tcs_" in row["Kernel_Name"]:
                agg += float(row["Counter_Value"]); cnt += 1
    pmc[name] = (agg, cnt)
if pmc["fetch"][1] and pmc["write"][1]:
    fkb = pmc["fetch"][0] / pmc["fetch"][1]; wkb = pmc["write"][0] / pmc["write"][1]
    traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline`, all ts::tcs_* dispatches",
               "dispatches": pmc["fetch"][1], "fetch_size_kb_per_dispatch": fkb, "write_size_kb_per_dispatch": wkb,
               "gfx950_correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B); WRITE_SIZE exact",
               "traffic_bytes_per_launch": fkb * 1024 * 2 + wkb * 1024}
    json.dump(traffic, open(os.path.join(out, "round2_traffic.json"), "w"), indent=1)
    md += [f"HBM-side traffic of the TCS launches (PMC, per launch): FETCH_SIZE {fkb:.0f} KB (x2 on gfx950 = {fkb*2048/1e6:.1f} MB), WRITE_SIZE {wkb:.0f} KB "
           f"({wkb*1024/1e6:.1f} MB) -> {traffic['traffic_bytes_per_launch']/1e6:.1f} MB per launch (`round2_traffic.json`).", ""]

for name, title, prefix in (("c3", "C3 Citrinet-1024 inference 32 x 20 s (`tools/bench_c3.py`)", "C3"),
                            ("c4p1", "C4 phase 1: fine-tune, encoder frozen (`tools/bench_finetune.py`)", "C4"),
                            ("c4p2", "C4 phase 2: fine-tune, everything trainable, bf16 activations, fwd+bwd from one hipGraph (`tools/bench_finetune.py --unfreeze --gemm-bf16 --graph`)", "C4"),
                            ("c5", "C5 wav2vec2-large inference 16 x 20 s (`tools/bench_c5.py`)", "C5")):
    rows = stats(name)
    if not rows:
        md += [f"## {title}", "", "(no trace)", ""]
        continue
    md += [f"## {title}", "", "bench line of the profiled run: " + last_line(name, prefix)[:400], ""]
    t, tot = table(rows, top=22)
    md += t + ["", f"total kernel time in the trace {tot/1e6:.1f} ms, {sum(int(r['Calls']) for r in rows)} launches (warm-up, capture and timed steps together).", ""]
open(os.path.join(out, "round2_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md[:40]))
