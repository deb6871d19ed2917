"""Per-kernel SQ counter ratios from a rocprofv3 --pmc run (counter_collection.csv):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d <dir> -- python ...
    python tools/pmc_sq.py <dir> [name filter]
mfma = MFMA-busy cycles / (duration x 1024 SIMDs x 2.4 GHz); the others are fractions of the resident wave-cycles."""
import csv, glob, re, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']); k = re.sub(r'\(.*', '', k)
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[k] += 1
print('  avg_us  mfma  wait_any wait_inst  valu   lds  any  waves/simd  kernel')
for k in sorted(dur, key=lambda k: -dur[k]):
    if flt not in k: continue
    v = acc[k]; wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    simd_cycles = dur[k] * 2.4 * 1024
    g = lambda n: v.get(n, 0) * 4 / wc
    print(f"{dur[k]/cnt[k]/1e3:8.1f} {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/simd_cycles:5.2f} {g('SQ_WAIT_ANY'):8.2f} {g('SQ_WAIT_INST_ANY'):8.2f} "
          f"{g('SQ_ACTIVE_INST_VALU'):6.2f} {g('SQ_ACTIVE_INST_LDS'):5.2f} {g('SQ_ACTIVE_INST_ANY'):5.2f} {wc*4/simd_cycles:8.2f}   {k[:70]}")
