"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: python tools/prof_summary.py <dir> [steps]"""
import csv, glob, re, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total GPU ms {tot/1e6:.1f}  per step {tot/1e6/steps:.1f}')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Name'])
    name = re.sub(r'\(.*', '', name)
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  calls/step {float(r['Calls'])/steps:6.1f}  avg {float(r['AverageNs'])/1e3:9.1f} us  ms/step {float(r['TotalDurationNs'])/1e6/steps:7.2f}  {name[:90]}")
