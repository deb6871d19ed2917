"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py conv_gemm [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, '2s-agcn_amd', 'csrc', sys.argv[1] + '.hip')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-c', src, '-o', '/dev/null',
                    '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
name = None
row = {}
for line in r.stderr.splitlines():
    m = re.search(r'remark:\s+(.*?): (.*?) \[-Rpass', line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        name = subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
        row = {}
    else:
        row[k] = v
        if k.startswith('LDS Size') and flt in name:
            print(f"{name:70s} VGPR {row.get('VGPRs'):>4} AGPR {row.get('AGPRs'):>3} scratch {row.get('ScratchSize [bytes/lane]'):>4} "
                  f"occ {row.get('Occupancy [waves/SIMD]')} vspill {row.get('VGPRs Spill')}")
if r.returncode:
    print(r.stderr[-2000:])
