import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
def short(n):
    n = re.sub(r"^void ", "", n); n = n.replace("siggan::", ""); n = re.sub(r"\(.*$", "", n)
    return n[:58]
tot = 0.0; mf = 0.0; nl = 0
out = []
for r in rows:
    n = int(r["Calls"]); us = float(r["TotalDurationNs"]) / 1e3
    if "elementwise_kernel" in r["Name"] or "fillBuffer" in r["Name"]: continue
    tot += us; nl += n
    if any(k in r["Name"] for k in ("k_gconv", "k_wgrad<", "k_fc_")) and "reduce" not in r["Name"]: mf += us
    out.append((us, n, short(r["Name"]), float(r["AverageNs"]) / 1e3))
out.sort(reverse=True)
print(f"per step: {tot/steps:.1f} us serialized, MFMA family {mf/steps:.1f}, everything else {(tot-mf)/steps:.1f}; launches/step {nl/steps:.1f}")
for us, n, name, avg in out:
    print(f"{name:60s} {n/steps:6.2f}/step  avg {avg:7.2f} us  {us/steps:7.1f} us/step")
