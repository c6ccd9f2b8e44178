import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1], d["ms_per_step"], "%.4g" % d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_timed"], "%.3f" % d["roofline"]["frac"], d.get("graph_replay", {}).get("ms_per_step"))
