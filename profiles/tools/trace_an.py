import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); n = n.replace("siggan::", "")
    return n[:46]
# find step boundaries: k_adam launches; D apply then G apply. A step = from after G adam to next G adam.
adam = [i for i, r in enumerate(rows) if ("k_adam(" in r["Kernel_Name"] or "k_adam<" in r["Kernel_Name"] or "k_adam_pack" in r["Kernel_Name"])]
print("adam launches", len(adam), file=sys.stderr)
# choose step near the end
k = len(adam) - 7
i0, i1 = adam[k] + 1, adam[k + 2] + 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = int(rows[adam[k]]["End_Timestamp"])
print(f"step span {(int(rows[i1-1]['End_Timestamp'])-prev_end)/1e3:.1f} us, {i1-i0} kernels")
busy_end = prev_end
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - busy_end
    busy_end = max(busy_end, e)
    print(f"{(s-prev_end)/1e3:8.1f} {(e-s)/1e3:7.1f} q{r['Queue_Id']:>2} {'GAP%.1f'%(gap/1e3) if gap>1500 else '':8} {short(r['Kernel_Name'])} g{r['Grid_Size_X']}")
