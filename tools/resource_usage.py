"""Per-kernel register / LDS / occupancy table from `make -C gymgo_amd/csrc resource-usage` (cross-compiles, no GPU).
  python tools/resource_usage.py [substring ...]"""
import re
import subprocess
import sys

ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))
out = subprocess.run(['make', '-C', ROOT + '/gymgo_amd/csrc', 'resource-usage'], capture_output=True, text=True)
rec, cur = {}, None
for l in (out.stdout + out.stderr).splitlines():
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1)
        rec[cur] = {}
        continue
    for k, short in (('VGPRs', 'vgpr'), ('VGPR Spill', 'spill'), ('ScratchSize [bytes/lane]', 'scratch'),
                     ('LDS Size [bytes/block]', 'lds'), ('Occupancy [waves/SIMD]', 'occ'), ('SGPRs', 'sgpr')):
        m = re.search(re.escape(k) + r': (\d+)', l)
        if m and cur:
            rec[cur][short] = int(m.group(1))
names = subprocess.run(['c++filt'], input='\n'.join(rec), capture_output=True, text=True).stdout.splitlines()
for mangled, d in zip(rec, names):
    d = d.split('(')[0].replace('void gg::', '')
    if not sys.argv[1:] or any(x in d for x in sys.argv[1:]):
        v = rec[mangled]
        print('%-58s vgpr %3d spill %3d scratch %4d lds %6d occ %d' % (d[:58], v.get('vgpr', -1), v.get('spill', -1),
                                                                   v.get('scratch', -1), v.get('lds', -1), v.get('occ', -1)))
