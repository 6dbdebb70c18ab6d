import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["gpu_telemetry"]["mean_socket_w"], d["gpu_telemetry"]["mean_sclk_mhz"])
for c in d["in_step_kernels"]["calls"]:
    if c["k"]==60: print(c["op"], round(c["ms"]*1000,1), "us min", round(c["ms_min"]*1000,1))
