import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in list(d)[:16]})
print(d.get("entry_point"))
print("unet peak_mem", d["secondary"]["detector"]["unet4_forward"]["peak_mem_gb"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"])
