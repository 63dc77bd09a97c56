import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
print("ms/step", round(d['ms_per_step'],3), "edges/s %.3g" % d['value'], "roof", d['roofline'] and round(d['roofline']['frac'],3), d['roofline'] and round(d['roofline']['fwd_kernel']['frac'],3))
tot=0
for k,v in d['kernels'].items():
    lps=v.get('launches_per_step', v.get('launches',0)/d['steps'])
    per=lps*v['avg_ms']
    tot+=per
    print(f"{k:28s} {lps:5.1f} x {v['avg_ms']*1e3:8.1f} us = {per:6.3f} ms")
print("sum", round(tot,3))
