#!/usr/bin/env python3
"""Prints the few numbers of a bench.py JSON line read from stdin (tools only)."""
import json
import sys

# (a file name, or stdin when there is none: never block on a terminal-less box)
src = open(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].endswith(".json") else sys.stdin
o = json.loads([l for l in src.read().strip().splitlines() if l.startswith("{")][-1])
c = o["config"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "threads", c["host_threads"], "passes",
      c["passes_per_step"], "ms/pass", round(c["ms_per_pass"], 4), "cand/s", "%.3e" % o["value"],
      "timed_s", round(c["timed_region_s"], 3), "nodes", c.get("nodes_expanded_per_step"),
      "found", c.get("found"), "frac", round(o["roofline"]["frac"], 4))
