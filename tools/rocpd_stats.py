"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: python tools/rocpd_stats.py gpurun_out/prof/r01_results.db [out.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n if len(n) < 110 else n[:107] + "..."


lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, t, a, mn, mx in rows[:40]:
    lines.append(f"| `{short(n)}` | {c} | {t / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * t / tot:.1f} |")
lines.append(f"\ntotal kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
# timeline of the decode steps: where the time between two logits_rules launches goes (kernels vs gaps)
try:
    ev = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, e in enumerate(ev) if "logits_rules_pick" in e[0] or "rules_pick" in e[0]]
    if len(marks) > 8:
        spans, busy, biggest = [], [], []
        for a, b in zip(marks[4:-1], marks[5:]):
            seg = ev[a + 1:b + 1]
            spans.append(seg[-1][2] - ev[a][2])
            busy.append(sum(e[2] - e[1] for e in seg))
            gaps = [seg[0][1] - ev[a][2]] + [seg[i + 1][1] - seg[i][2] for i in range(len(seg) - 1)]
            biggest.append(max(gaps))
        n = len(spans)
        lines.append(f"decode steps (between consecutive rules_pick ends, n={n}): span {sum(spans) / n / 1e3:.1f} us, kernel time {sum(busy) / n / 1e3:.1f} us, "
                     f"idle {(sum(spans) - sum(busy)) / n / 1e3:.1f} us, largest single gap {sum(biggest) / n / 1e3:.1f} us")
except Exception as e:  # pragma: no cover
    lines.append(f"(timeline analysis skipped: {e})")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
