# prints "batch ms" pairs of one sweep_batch.py JSON line read from stdin
import json, sys
j = json.loads(sys.stdin.readline())
print(" ".join("%d:%.3f" % (p["batch"], p["ms_per_search"]) for p in j["points"]))
