#!/usr/bin/env python3
"""usage: tools/kernel_resources.py <file.hip> [regex] [extra hipcc flags...] -> VGPR/SGPR/scratch/occupancy per kernel"""
import re
import subprocess
import sys

f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else '.'
extra = sys.argv[3:]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-gpu-rdc', '--cuda-device-only', '-c', f,
       '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + extra
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
keys = [('vgpr', r'VGPRs'), ('agpr', r'AGPRs'), ('sgpr', r'SGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
        ('occ', r'Occupancy \[waves/SIMD\]'), ('lds', r'LDS Size \[bytes/block\]')]
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split()[0]
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(.*', '', dem).replace('void pnp::', '')
    if not re.search(pat, dem):
        continue
    vals = []
    for k, rx in keys:
        m = re.search(rx + r': (\d+)', b)
        vals.append(f'{k}={m.group(1) if m else "?":>4s}')
    print(f'{dem:52s} ' + ' '.join(vals))
