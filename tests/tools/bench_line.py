"""Reads bench.py's JSON line on stdin, prints the few numbers worth comparing."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print("%s value %.4e rows/s  step %.3f ms  %s %.0f GB/s (%.1f%%)  kernel %.3f ms" %
              (tag, d["value"], d["ms_per_step"], r["kernel"], r["achieved"], 100 * r["frac"], r["avg_kernel_ms"]))
