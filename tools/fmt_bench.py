import sys,json
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    r=d["roofline"]
    print(d["config"]["batches_in_flight_per_gpu"], "Matoms/s", round(d["value"]/1e6,2), "ms/step", round(d["ms_per_step"],4), r["kernel"][:30], "k_ms", round(r["kernel_ms_avg"],4), "fused_ms", round(r["fused_stage_ms_avg"],4), "wholeTF", round(r["whole_forward_tflops"],1))
