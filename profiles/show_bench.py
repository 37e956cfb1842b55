import json
d=json.loads(open("gpurun_out/r03/bench_sec.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in d["secondary_kernels"]: print({a:k.get(a) for a in ("overrides","kernel","avg_kernel_ms","frac","error")})
for k in d["step_per_launch"]: print(k)
print(d.get("sac_first_capture"))
print(d.get("extras_timed_out"))
