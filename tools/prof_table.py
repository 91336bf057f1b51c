"""Per-kernel table of the TIMED steps of `python bench.py`, from a rocprofv3 kernel trace of that same command.

    python tools/prof_table.py <..._kernel_trace.csv> <bench JSON line file> [out.json]

bench.py launches an empty `db1_marker_kernel` right before its first timed step and right after its last one; this script keeps the
dispatches between the two markers (warm-up, model construction, the mixture / decode legs and the CPU baseline fall outside), groups
them by kernel, and divides the work per step the bench line states for every timed family (`work_per_step`: FLOPs for the MFMA-bound
families, bytes for the HBM-bound ones -- the same numbers the line's HIP-event fractions use) by the profiler's own kernel durations.
So `families.gemm.frac` here and `roofline.frac` in the line are the same quantity measured two ways on the same run.
"""
import csv
import json
import re
import sys

MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBPS = 8000.0

# kernel name (regex on the demangled name) -> timed family of bdm_db1_amd/ops.py.  "gemm" is the roofline family of the bench line: every
# tile GEMM incl. the head sweep's products, its split-K reduces and the sweep's loss kernels (they sit behind one C call with the GEMMs).
FAMILIES = [
    ("gemm", r"gemm_bf16_\w+_kernel|splitk_reduce_kernel|ce_fwd_bwd_kernel|ce_sum_kernel", "mfma"),
    ("flash_fwd", r"relattn_flash_fwd", "mfma"),
    ("flash_bwd", r"relattn_flash_bwd", "mfma"),
    ("relattn_dqr", r"relattn_dqr", "hbm"),
    ("layernorm_fwd", r"ln_fwd", "hbm"),
    ("layernorm_bwd", r"ln_bwd|ln_param_reduce", "hbm"),
    ("ffn_act_fwd", r"act_fwd_kernel", "hbm"),
    ("ffn_act_bwd", r"act_bwd", "hbm"),
    ("adam", r"adam_kernel", "hbm"),
    ("sumsq", r"sumsq", "hbm"),
]


def short(name: str) -> str:
    return re.sub(r"^void ", "", name).split("(")[0][:100]


def main():
    trace, line_file = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    line = None
    for ln in open(line_file):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    if line is None:
        raise SystemExit(f"no bench JSON line in {line_file}")
    steps = int(line["steps"])
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "db1_marker_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit("the trace holds no pair of db1_marker_kernel dispatches: was it taken from bench.py?")
    lo, hi = marks[0], marks[1]
    sel = rows[lo + 1:hi]
    t_region = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["End_Timestamp"])) / 1e6   # ms between the markers
    per = {}
    for r in sel:
        k = short(r["Kernel_Name"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3   # us
        e = per.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += d
    busy_ms = sum(v[1] for v in per.values()) / 1e3
    work = line.get("work_per_step", {})
    if "lmhead_ce" in work and "gemm" in work:
        work = dict(work, gemm=work["gemm"] + work["lmhead_ce"])
    fam_rows = {}
    kernels = []
    for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        fam = next((f for f, rx, _ in FAMILIES if re.search(rx, k)), None)
        kernels.append({"kernel": k, "family": fam, "calls_per_step": round(n / steps, 2), "avg_us": round(us / n, 1),
                        "ms_per_step": round(us / 1e3 / steps, 3), "share_of_kernel_time": round(us / 1e3 / busy_ms, 4)})
        if fam:
            fr = fam_rows.setdefault(fam, [0, 0.0])
            fr[0] += n
            fr[1] += us
    families = {}
    for fam, rx, bound in FAMILIES:
        if fam not in fam_rows:
            continue
        n, us = fam_rows[fam]
        ms_step = us / 1e3 / steps
        rec = {"bound": bound, "kernels_per_step": round(n / steps, 1), "ms_per_step": round(ms_step, 3), "share_of_kernel_time": round(us / 1e3 / busy_ms, 4)}
        if fam in work:
            rate = work[fam] / (ms_step * 1e-3) / (1e12 if bound == "mfma" else 1e9)
            rec.update({"work_per_step": work[fam], "achieved": round(rate, 1), "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                        "frac": round(rate / (MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBPS), 4)})
        families[fam] = rec
    named = sum(v["ms_per_step"] for v in families.values())
    other = busy_ms / steps - sum(families[f]["ms_per_step"] for f in families if f in ("gemm", "flash_fwd", "flash_bwd", "relattn_dqr"))
    out = {"command": "rocprofv3 --kernel-trace -- python bench.py (tools/prof_step.sh)", "steps": steps,
           "region_ms_per_step": round(t_region / steps, 3), "kernel_busy_ms_per_step": round(busy_ms / steps, 3),
           "bench_line": {k: line.get(k) for k in ("value", "ms_per_step", "pct_mfma_peak_step")},
           "bench_line_roofline": {k: line.get("roofline", {}).get(k) for k in ("frac", "achieved", "ms_per_step", "flop_per_step")},
           "non_gemm_non_attention_share": round(other / (busy_ms / steps), 4),
           "families": families, "kernels": kernels[:60]}
    if "gemm" in families and line.get("roofline"):
        out["gemm_frac_hip_events_vs_rocprof"] = {"hip_events": line["roofline"]["frac"], "rocprof": families["gemm"].get("frac"),
                                                  "ratio": round(families["gemm"].get("frac", 0) / max(line["roofline"]["frac"], 1e-9), 4)}
    txt = json.dumps(out, indent=1)
    if out_path:
        open(out_path, "w").write(txt + "\n")
    print(f"region {out['region_ms_per_step']} ms/step, kernels busy {out['kernel_busy_ms_per_step']} ms/step, named families {named:.1f} ms/step")
    print(f"{'kernel':100s} {'calls/step':>10s} {'avg us':>9s} {'ms/step':>9s} {'share':>7s}")
    for k in kernels[:40]:
        print(f"{k['kernel']:100s} {k['calls_per_step']:10.2f} {k['avg_us']:9.1f} {k['ms_per_step']:9.3f} {k['share_of_kernel_time']:7.4f}")
    for f, v in families.items():
        print(f"family {f:14s} {v['ms_per_step']:9.3f} ms/step  {v.get('achieved', '')} {v.get('unit', '')}  frac {v.get('frac', '')}")
    if "gemm_frac_hip_events_vs_rocprof" in out:
        print("GEMM family frac: HIP events", out["gemm_frac_hip_events_vs_rocprof"])


if __name__ == "__main__":
    main()
