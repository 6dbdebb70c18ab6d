#!/usr/bin/env python
"""Compile one csrc/*.hip file for gfx950 and print a compact per-kernel resource table
(VGPRs, spills, scratch, SGPRs, LDS) from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
    python tools/kres.py qk_hgemm_bf16mfma.hip [filter-substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
CSRC = os.path.join(ROOT, 'quaternion-convolutional-neural-networks-for-end-to-end-automatic-speech-recognition_amd', 'csrc')
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ["hipcc"] + os.environ.get("KRES_FLAGS", "").split() + ['--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fPIC', '-Wno-unused-value',
       '-c', os.path.join(CSRC, src), '-o', '/tmp/kres.o', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'remark:\s+Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    elif ' error' in line:
        print(line)
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace('void qk::(anonymous namespace)::', '').replace('qk::', '')
    n = re.sub(r'\(.*', '', n)
    if flt and flt not in n:
        continue
    print('%-72s vgpr %3d spill %3d scratch %4d sgpr %3d lds %6d occ %d' % (
        n[:72], r.get('VGPRs', -1), r.get('VGPRs Spill', -1), r.get('ScratchSize', -1),
        r.get('TotalSGPRs', -1), r.get('LDS Size', -1), r.get('Occupancy', -1)))
