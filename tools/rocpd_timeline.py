"""Timeline view of a rocprofv3 kernel trace (rocpd sqlite): is the chip ever idle, how many kernels run side by side, and what does each queue
(= HIP stream = engine lane) look like -- busy fraction and the gaps between its consecutive kernels.
usage: python tools/rocpd_timeline.py <results.db> [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
ev = db.execute(f"select start, end, {qcol or '0'}, {name_col} from kernels order by start").fetchall()
# steady state only: the middle of the trace (drop the warm-up / fill at the head and the drain at the tail)
t_first, t_last = ev[0][0], max(e[1] for e in ev)
lo, hi = t_first + 0.45 * (t_last - t_first), t_first + 0.85 * (t_last - t_first)
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
t0, t1 = ev[0][0], max(e[1] for e in ev)
span = t1 - t0
pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
busy = 0
lvl = 0
hist = {}
prev = pts[0][0]
for t, d in pts:
    if t > prev:
        hist[lvl] = hist.get(lvl, 0) + (t - prev)
        if lvl > 0:
            busy += t - prev
    lvl += d
    prev = t
tot = sum(e[1] - e[0] for e in ev)
lines = [f"columns: {cols}", f"window {span / 1e6:.1f} ms, {len(ev)} kernels; some kernel running {100 * busy / span:.1f} % of the time; sum of kernel durations / window = {tot / span:.2f}",
         "time with k kernels in flight: " + ", ".join(f"{k}: {100 * v / span:.1f} %" for k, v in sorted(hist.items()) if v / span > 0.002)]
byq = {}
for s, e, q, n in ev:
    byq.setdefault(q, []).append((s, e, n))
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    if len(lst) < 50:
        continue
    gaps = sorted(lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1))
    qb = sum(e - s for s, e, _ in lst)
    qspan = lst[-1][1] - lst[0][0]
    big = sum(g for g in gaps if g > 50e3)
    enc = sum(e - s for s, e, n in lst if "gemm" in n or "enc_attn" in n or "layernorm" in n or "mel_" in n)
    lines.append(f"queue {q}: encoder-side kernels {100 * enc / qspan:.1f} % of its span; {len(lst)} kernels, busy {100 * qb / qspan:.1f} % of its span, gaps between consecutive kernels: median {gaps[len(gaps) // 2] / 1e3:.2f} us, "
                 f"p90 {gaps[int(0.9 * len(gaps))] / 1e3:.2f} us, gaps > 50 us add up to {100 * big / qspan:.1f} % of the span")
# where the long gaps of a queue sit: (kernel before -> kernel after), summed
import re
def short(n):
    n = re.sub(r"\(.*", "", n)
    m = re.search(r"ss(\d+)(\w+?)I", n)
    return (n[6:44] if n.startswith("_ZN2ss") else n[:38])
agg = {}
for q, lst in byq.items():
    for i in range(len(lst) - 1):
        g = lst[i + 1][0] - lst[i][1]
        if g > 50e3:
            k = (short(lst[i][2]), short(lst[i + 1][2]))
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += g; a[2] = max(a[2], g)
lines.append("long gaps (> 50 us) by (kernel before -> kernel after): count, total ms, longest ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    lines.append(f"   {k[0]} -> {k[1]}: {a[0]}, {a[1] / 1e6:.1f}, {a[2] / 1e6:.2f}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
