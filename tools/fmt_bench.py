import sys, json
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(d["config"]["batches_in_flight_per_gpu"], "Matoms/s", round(d["value"] / 1e6, 2), "ms/step", round(d["ms_per_step"], 4),
          r["kernel"][:30], "k_ms", round(r["kernel_ms_avg"], 4), "in_flight", round(r.get("launches_in_flight", 0), 2),
          "TF", round(r["achieved"], 1), "frac", round(r["frac"], 3))
