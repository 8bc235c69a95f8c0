"""Condense the rocprofv3 --pmc passes of tools/prof_pmc_tcs.sh into profiles/round<N>_tcs_sq_counters.md: per TCS kernel instantiation the mean
of every counter per dispatch, and the ratios the DESIGN quotes (matrix-core busy share, LDS active / conflict share, wave-cycle split)."""
import collections, csv, glob, os, sys

O, R = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "5")
vals = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [per dispatch]
dur = collections.defaultdict(list)                                         # kernel -> [dispatch duration, ns] (under the profiler)
for f in glob.glob(f"{O}/p*/**/*counter_collection.csv", recursive=True):
    per, seen = collections.defaultdict(float), set()
    for row in csv.DictReader(open(f)):
        if "tcs_" not in row["Kernel_Name"]:
            continue
        per[(row["Dispatch_Id"], row["Kernel_Name"], row["Counter_Name"])] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"])
            dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for (_, k, c), v in per.items():
        vals[k][c].append(v)
if not vals:
    print("no counters collected"); print(open(f"{O}/p1.log").read()[-2000:]); sys.exit(1)
N_CU, SIMD = 256, 4
out = [f"# Round {R} -- SQ counters of the fused TCS launches (C2 headline step, 64 x 15 s)", "",
       "`bash tools/prof_pmc_tcs.sh` on one MI355X: three `rocprofv3 --pmc` passes (no trace domain beside them) over "
       "`python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-trained-check`; every number is the mean per dispatch of one kernel "
       "instantiation (template arguments: NPASS = depthwise tap passes of 12 taps, XJ, MT, WM, DIL, NT, CHAIN, SE).", "",
       "Units (MI355X_MICROARCH.md): `SQ_WAVE_CYCLES`, `SQ_WAIT_*`, `SQ_ACTIVE_INST_*` count quad-cycles summed over waves; `SQ_VALU_MFMA_BUSY_CYCLES` counts "
       "cycles summed over SIMDs.  The kernels are persistent (every wave lives for the whole dispatch), so the dispatch lasts L = 4 x SQ_WAVE_CYCLES / SQ_WAVES "
       "shader cycles; derived: matrix core busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x L); LDS active = SQ_LDS_IDX_ACTIVE / (256 CUs x L); conflict share = "
       "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; the wave-cycle split is WAIT_ANY : WAIT_INST_ANY : ACTIVE_INST_ANY (disjoint, ~ WAVE_CYCLES).  "
       "`GRBM_GUI_ACTIVE` is listed but not used: under counter collection it spans the profiler's own bracket, not the kernel.", ""]
mean = lambda xs: sum(xs) / len(xs) if xs else float("nan")
rows = []
for k in sorted(vals, key=lambda k: -mean(vals[k].get("SQ_WAVE_CYCLES", [0])) * len(vals[k].get("SQ_WAVE_CYCLES", []))):
    c = {n: mean(v) for n, v in vals[k].items()}
    n_disp = max(len(v) for v in vals[k].values())
    L = 4.0 * c.get("SQ_WAVE_CYCLES", float("nan")) / max(c.get("SQ_WAVES", 1), 1)
    us = mean(dur[k]) / 1e3
    out += [f"## `{k[:140]}` ({n_disp // 3 if n_disp >= 3 else n_disp} dispatches per pass)", "", "| counter | mean per dispatch |", "|---|---|"]
    out += [f"| {n} | {v:,.0f} |" for n, v in sorted(c.items())]
    d = ["", f"* dispatch: {us:.1f} us under the profiler; wave lifetime L = {L:,.0f} cycles ({L / us / 1e3:.2f} GHz implied)"]
    mf = 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (N_CU * SIMD * L)
    d.append(f"* matrix core busy: {mf:.1f} % of SIMD-cycles (co-issued with vector instructions: "
             f"{100 * c.get('SQ_VALU_MFMA_COEXEC_CYCLES', float('nan')) / max(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 1), 1):.1f} % of the busy cycles); "
             f"{c.get('SQ_INSTS_MFMA', float('nan')):,.0f} MFMA instructions")
    la = 100 * c.get("SQ_LDS_IDX_ACTIVE", float("nan")) / (N_CU * L)
    cf = 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    d.append(f"* LDS active: {la:.1f} % of CU-cycles; bank-conflict cycles {cf:.1f} % of the active cycles; address conflicts {c.get('SQ_LDS_ADDR_CONFLICT', 0):,.0f}, "
             f"unaligned stalls {c.get('SQ_LDS_UNALIGNED_STALL', 0):,.0f}")
    tot = c.get("SQ_WAIT_ANY", 0) + c.get("SQ_WAIT_INST_ANY", 0) + c.get("SQ_ACTIVE_INST_ANY", 0)
    if tot:
        d.append(f"* wave-cycle split: parked (s_waitcnt / barrier) {100 * c['SQ_WAIT_ANY'] / tot:.1f} %, issue-stalled {100 * c.get('SQ_WAIT_INST_ANY', 0) / tot:.1f} % "
                 f"(LDS issue {100 * c.get('SQ_WAIT_INST_LDS', 0) / tot:.1f}), issuing {100 * c['SQ_ACTIVE_INST_ANY'] / tot:.1f} % (vector ALU incl. MFMA "
                 f"{100 * c.get('SQ_ACTIVE_INST_VALU', 0) / tot:.1f}, LDS {100 * c.get('SQ_ACTIVE_INST_LDS', 0) / tot:.1f}, vector memory {100 * c.get('SQ_ACTIVE_INST_VMEM', 0) / tot:.1f}, "
                 f"scalar {100 * c.get('SQ_ACTIVE_INST_SCA', 0) / tot:.1f})")
    out += d + [""]
    rows.append((k, n_disp, us, L, mf, la, cf, 100 * c.get("SQ_WAIT_ANY", 0) / tot if tot else float("nan")))
tab = ["## Summary", "", "| instantiation | us (profiled) | matrix core busy % | LDS active % | conflict % of LDS cycles | waves parked % |", "|---|---|---|---|---|---|"]
for k, n, us, L, mf, la, cf, pk in rows:
    name = k[k.index("<"):k.index(">") + 1] if "<" in k else k
    tab.append(f"| `{k.split('::')[1].split('<')[0]}{name}` | {us:.1f} | {mf:.1f} | {la:.1f} | {cf:.1f} | {pk:.1f} |")
out = out[:6] + tab + [""] + out[6:]
os.makedirs(os.path.join(O, "profiles"), exist_ok=True)
path = os.path.join(O, "profiles", f"round{R}_tcs_sq_counters.md")
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(tab))
