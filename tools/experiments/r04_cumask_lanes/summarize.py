"""One line per point of ab.sh's JSON-lines file."""
import json
import sys

for ln in open(sys.argv[1]):
    r = json.loads(ln)
    if r.get("failed"):
        print(f"{r['label']:42s} FAILED")
        continue
    j = r["bench"]
    rf = j["roofline"]
    print(f"{r['label']:42s} {j['value']:8.1f} xRT  p50 {j.get('p50_chunk_latency_ms', 0):7.1f} ms  pass {rf.get('avg_launch_ms', 0):6.3f} ms x "
          f"{rf.get('rows_per_launch', 0):5.1f} rows  roofline {rf['frac']:.3f}")
