"""Summarise the clock / instruction-mix PMC pass of tools/final_profiles.sh (GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE): effective clock, VALU instructions per MFMA instruction, LDS conflict share."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
files = sorted(glob.glob(d + '/*/*counter_collection.csv'))
rows = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
cnt = collections.defaultdict(int)
seen = set()
for r in csv.DictReader(open(files[-1])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    k = re.sub(r'^void ', '', k).split('(')[0]
    rows[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'])
    if key not in seen:
        seen.add(key)
        dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        cnt[k] += 1
print('Second PMC pass of the headline step (rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA '
      'SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE):')
print('effective clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back); VALU per MFMA '
      'instruction; LDS conflict share')
print('  avg_us  clock_GHz  valu/mfma  lds_conflict  kernel')
for k in sorted(dur, key=lambda k: -dur[k])[:top]:
    c = rows[k]
    ghz = c['GRBM_GUI_ACTIVE'] / 8 / dur[k] if dur[k] else 0
    vm = '%9.1f' % (c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']) if c['SQ_INSTS_MFMA'] else '        -'
    lc = c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] if c['SQ_LDS_IDX_ACTIVE'] else 0
    print('%8.1f  %9.2f  %s  %12.3f  %s' % (dur[k] / cnt[k] / 1e3, ghz, vm, lc, k[:60]))
