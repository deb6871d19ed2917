"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_roofline.py into per-kernel HBM bytes.
Units/corrections per MI355X_MICROARCH.md (HBM section): counters are in KB; on gfx950 FETCH_SIZE reports exactly 1/2
of a coalesced streaming read -- confirmed here on two calibration kernels with known byte counts (bn_act_fwd: 16 B/lane,
2x245.76 MB read -> 245.8 reported; bn_bwd_reduce: 4 B/lane, 3x245.76 MB read -> 370.0 reported) -- so reads are doubled.
usage: python tools/pmc_parse.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, re, sys

def load(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
        name = re.sub(r'\(.*', '', name)
        out[(name, r['Grid_Size'])].append(float(r['Counter_Value']))
    return out

fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
res = {}
for key in sorted(set(fetch) | set(write)):
    name, grid = key
    f = fetch.get(key, [0.0])[-1] * 1024 * 2.0        # KB -> bytes, x2 gfx950 correction
    w = write.get(key, [0.0])[-1] * 1024
    res[f'{name} grid={grid}'] = {'read_bytes': f, 'write_bytes': w, 'hbm_bytes': f + w}
import os, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_fingerprint
try:
    head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
except OSError:
    head = None
res['_meta'] = {'csrc_sha': csrc_fingerprint(), 'git_head': head or os.environ.get('AGCN_GIT_HEAD'),
                'collected': time.strftime('%Y-%m-%d %H:%M:%S'),
                'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); reads x2 (gfx950), KB -> bytes'}
json.dump(res, open(sys.argv[3], 'w'), indent=1)
for k, v in res.items():
    if k != '_meta' and v['hbm_bytes'] > 1e6:
        print(f"{v['hbm_bytes']/1e6:9.1f} MB  (R {v['read_bytes']/1e6:8.1f}  W {v['write_bytes']/1e6:8.1f})  {k[:100]}")
