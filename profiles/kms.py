# prints the kernel span (ms) and ms_per_step of one bench.py JSON line read from stdin
import json, sys
for line in sys.stdin:
    if line.startswith("{"):
        j = json.loads(line)
        print("kernel_avg_ms %.3f  ms_per_step %.3f  needles %s" % (j["roofline"]["kernel_avg_ms"], j["ms_per_step"], j.get("needles_found")))
