"""Condense the rocprofv3 outputs of tools/prof_round.sh into the files that go under profiles/ (round<N>_*).  Per-kernel tables cover the
TIMED REGION only: the dispatches between the two marker pairs (tools/prof_mark.py) that bracket each timed loop -- warm-up passes,
graph capture and one-time weight packing are dropped, so the percentages are per-step shares."""
import collections, csv, glob, json, os, sys

O = sys.argv[1]
R = sys.argv[2] if len(sys.argv) > 2 else "4"
out = os.path.join(O, "profiles")
os.makedirs(out, exist_ok=True)
MARK = "counter_add_kernel"


def trace(name, anchor=None):
    fs = sorted(glob.glob(f"{O}/{name}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    if not fs:
        return None, 0, 0
    rows = list(csv.DictReader(open(fs[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i in range(len(rows) - 1) if MARK in rows[i]["Kernel_Name"] and MARK in rows[i + 1]["Kernel_Name"]]   # a marker is a PAIR
    if len(marks) >= 2:
        rows = [r for r in rows[marks[0] + 2: marks[-1]] if MARK not in r["Kernel_Name"]]
    agg = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c, t = agg.get(k, (0, 0))
        agg[k] = (c + 1, t + d)
    table = sorted(({"Name": k, "Calls": c, "TotalDurationNs": t, "AverageNs": t / c} for k, (c, t) in agg.items()), key=lambda r: -r["TotalDurationNs"])
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) if rows else 0
    with open(os.path.join(out, f"round{R}_{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        tot = sum(r["TotalDurationNs"] for r in table)
        for r in table:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], f"{r['AverageNs']:.1f}", f"{100 * r['TotalDurationNs'] / max(tot, 1):.2f}"])
    steps = next((r["Calls"] for r in table if anchor and anchor in r["Name"]), 0)
    return table, span, steps


def last_json(name):
    try:
        lines = [l.strip() for l in open(f"{O}/{name}.log") if l.strip().startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except (OSError, ValueError):
        return None


def md_table(rows, top=18):
    tot = sum(r["TotalDurationNs"] for r in rows)
    t = ["| kernel | calls | total ms | avg us | % of kernel time |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        t.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {r['TotalDurationNs'] / 1e6:.2f} | {r['AverageNs'] / 1e3:.1f} | {100 * r['TotalDurationNs'] / max(tot, 1):.1f} |")
    return t, tot


md = [f"# Round {R} -- rocprofv3 summaries (tools/prof_round.sh {R} on one MI355X)", "",
      "Tables cover the timed region of each run only (between the two marker pairs of tools/prof_mark.py).", ""]
rows, span, _ = trace("bench")
if rows:
    bj = last_json("bench")
    if bj:
        json.dump(bj, open(os.path.join(out, f"round{R}_bench.json"), "w"), indent=1)
        md += ["## C2 headline: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 5 --no-extra --no-cpu-baseline --no-trained-check`", "",
               f"bench line of the profiled run: value {bj['value']:.0f} {bj['unit']}, {bj['ms_per_step']:.3f} ms/step, encoder {bj['roofline']['encoder_ms']:.3f} ms, "
               f"avg_launch_us {bj['roofline']['avg_launch_us']:.1f} (HIP events), roofline.frac {bj['roofline']['frac']:.3f}", ""]
    t, tot = md_table(rows)
    md += t
    tcs = [r for r in rows if "tcs_" in r["Name"]]
    n = sum(r["Calls"] for r in tcs); ns = sum(r["TotalDurationNs"] for r in tcs)
    steps = bj["steps"] if bj else 100
    md += ["", f"All `ts::tcs_*` kernels in the timed region: {n} calls ({n / steps:.0f} per step), {ns / 1e6:.2f} ms, average {ns / max(n, 1) / 1e3:.1f} us per launch; "
               f"kernel time {tot / 1e6 / steps:.3f} ms per step of {span / 1e6 / steps:.3f} ms wall per step in the trace.", ""]

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
    traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-trained-check`, all ts::tcs_* dispatches",
               "dispatches": pmc["fetch"][1], "fetch_size_kb_per_dispatch": fkb, "write_size_kb_per_dispatch": wkb,
               "gfx950_correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B); WRITE_SIZE exact",
               "traffic_bytes_per_launch": fkb * 1024 * 2 + wkb * 1024}
    json.dump(traffic, open(os.path.join(out, f"round{R}_traffic.json"), "w"), indent=1)
    md += [f"HBM-side traffic of the TCS launches (PMC, per launch): FETCH_SIZE {fkb:.0f} KB (x2 on gfx950 = {fkb * 2048 / 1e6:.1f} MB), WRITE_SIZE {wkb:.0f} KB "
           f"({wkb * 1024 / 1e6:.1f} MB) -> {traffic['traffic_bytes_per_launch'] / 1e6:.1f} MB per launch (`round{R}_traffic.json`).", ""]

for name, title, key, anchor in (("c3", "C3 Citrinet-1024 inference 32 x 20 s (`tools/bench_extra.py c3`)", "c3", "stft_mel_kernel"),
                         ("c4p1", "C4 phase 1 as the reference schedule leaves the model: convolutions frozen, BatchNorm + decoder trainable, train-mode encoder "
                                  "(`TS_C4_ONLY=c4_phase1 tools/bench_extra.py c4`)", "c4_phase1", "ctc_kernel"),
                         ("c4p2", "C4 phase 2: everything trainable, bf16 activations, fwd + bwd from one hipGraph (`TS_C4_ONLY=c4_phase2 tools/bench_extra.py c4`)", "c4_phase2", "ctc_kernel"),
                         ("c4p2f", "C4 phase 2 in fp32 (every GEMM on the f32 matrix-core kernel of csrc/gemm_f32.hip; no vendor library is linked) "
                                   "(`TS_C4_ONLY=c4_phase2_fp32 tools/bench_extra.py c4`)", "c4_phase2_fp32", "ctc_kernel"),
                         ("c5", "C5 wav2vec2-large inference 16 x 20 s, own GEMM (`tools/bench_extra.py c5`)", "c5", "w2v_posconv_mfma_kernel"),
                         ("c5ft", "wav2vec2-large fine-tuning step 8 x 10 s, f32, eager launches (`TS_C5FT_ONLY=fp32 tools/bench_extra.py c5_finetune`)", "c5_finetune",
                          "w2v_conv0_finalize_kernel"),
                         ("c5ftb", "wav2vec2-large fine-tuning step 8 x 10 s, mixed precision (bf16 operands / f32 accumulation), eager launches "
                                   "(`TS_C5FT_ONLY=bf16 tools/bench_extra.py c5_finetune`)", "c5_finetune_bf16", "w2v_conv0_finalize_kernel")):
    rows, span, traced = trace(name, anchor)
    if not rows:
        md += [f"## {title}", "", "(no trace)", ""]
        continue
    bj = (last_json(name) or {}).get(key, {})
    steps = traced or bj.get("steps", 1)
    md += [f"## {title}", "", f"bench line of the profiled run: {bj.get('ms_per_step', float('nan')):.2f} ms/step, {bj.get('value', float('nan')):.1f} {bj.get('unit', '')}", ""]
    t, tot = md_table(rows, top=22)
    md += t + ["", f"timed region: {sum(r['Calls'] for r in rows)} launches ({sum(r['Calls'] for r in rows) / steps:.0f} per step), kernel time {tot / 1e6 / steps:.2f} ms per step, "
                   f"{span / 1e6 / steps:.2f} ms wall per step in the trace.", ""]
open(os.path.join(out, f"round{R}_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md[:45]))
