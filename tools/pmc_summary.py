"""Summarise the rocprofv3 --pmc passes of the bench command (tools/gpu.sh pmc): HBM-side bytes per decoder pass and per encoder FC1 launch.
bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes); the x2 is the gfx950 FETCH_SIZE correction for 16-B/lane streams (MI355X_MICROARCH.md, HBM).
usage: python tools/pmc_summary.py <dir with pmc_FETCH_SIZE_results.db, pmc_WRITE_SIZE_results.db> <tag> [out.json] [out.md] [dtype]"""
import json
import sqlite3
import sys

d, tag = sys.argv[1], sys.argv[2]
dtype = sys.argv[5] if len(sys.argv) > 5 else "f16"
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(f"{d}/pmc_{c}_results.db")
    for name, n, tot, mx in db.execute("select kernel_name, count(*), sum(value), max(value) from counters_collection where counter_name = ? group by kernel_name", (c,)):
        per.setdefault(name, {})[c] = (n, tot, mx)
rows = []
n_pass = 0
dec_bytes = 0.0
fc1 = None
for name, v in per.items():
    n = v.get("FETCH_SIZE", v.get("WRITE_SIZE"))[0]
    f = v.get("FETCH_SIZE", (0, 0.0, 0.0))
    w = v.get("WRITE_SIZE", (0, 0.0, 0.0))
    b = (2.0 * f[1] + w[1]) * 1024.0
    rows.append((b, name, n, f[1] / max(1, n), w[1] / max(1, n)))
    if "logits_rules_pick" in name:
        n_pass = n
    if any(k in name for k in ("dec_", "logits_rules", "logits_probs", "sample_draw", "skinny", "embed_kernel")) or ("layernorm" in name and n > 400):
        dec_bytes += b
    is_fc1 = (("gemm_f8" in name and "Li1E" in name) if dtype == "fp8" else ("gemm256" in name and "Li1E" in name)) and "selftest" not in name     # gemm256_kernel / gemm256k64_kernel (round 4 on) / gemm_f8k128_kernel
    if is_fc1:      # EPI_GELU_T / F8_GELU_F8: FC1 (+ conv1 in the f16 engines): the launch with the most traffic is an FC1
        fc1 = (2.0 * f[2] + w[2]) * 1024.0
rows.sort(reverse=True)
lines = ["| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | total (2F+W) MB |", "|---|---|---|---|---|"]
for b, name, n, fa, wa in rows[:24]:
    lines.append(f"| `{name[:90]}` | {n} | {fa:.1f} | {wa:.1f} | {b / 1e6:.1f} |")
lines.append(f"\ndecoder passes in this run: {n_pass}; HBM-side bytes of all decoder kernels per pass: {dec_bytes / max(1, n_pass) / 1e6:.1f} MB")
print("\n".join(lines))
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write("\n".join(lines) + "\n")
if len(sys.argv) > 3:
    try:
        j = json.load(open(sys.argv[3]))
    except Exception:
        j = {}
    rows_pl = alg_b = None
    try:     # the bench line of the PMC run itself: rows per pass and algorithmic bytes of THAT run (bench.py scales the counters to other row counts)
        bl = json.loads([l for l in open(f"{d}/bench_FETCH_SIZE.log") if l.startswith("{")][-1])
        rows_pl, alg_b = bl["roofline"]["rows_per_launch"], bl["roofline"]["algorithmic_bytes"]
    except Exception:
        pass
    src = f"profiles/{tag}_pmc.md (tools/gpu.sh pmc: rocprofv3 --kernel-trace --pmc, one counter per pass of the bench command)"
    if n_pass:
        j[f"large-v3/batch8/{dtype}/decoder_pass"] = {"bytes_per_launch": dec_bytes / n_pass, "passes": n_pass, "rows_per_launch": rows_pl, "algorithmic_bytes": alg_b, "source": src,
                                                 "note": "sum over every decoder-side kernel of (2 x FETCH_SIZE + WRITE_SIZE) / decoder passes; the pass carried the rows the default bench configuration merges (see rows_per_launch)"}
    if fc1:
        j[f"large-v3/batch8/{dtype}/fc1"] = {"bytes_per_launch": fc1, "source": src, "note": "max over the launches of the FC1 GEMM kernel (gemm256_kernel<EPI_GELU_T> / gemm_f8_kernel<F8_GELU_F8>)"}
    json.dump(j, open(sys.argv[3], "w"), indent=1)
