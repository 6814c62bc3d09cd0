"""Per-kernel resource table of the gfx950 build: registers, spills, LDS, occupancy, as reported by the compiler
(`-Rpass-analysis=kernel-resource-usage`, same flags as passl_amd/csrc/build.py).  Runs without a GPU.
    python tools/kernel_resources.py > profiles/<round>_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from passl_amd.csrc import build as B          # noqa: E402


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<.*>)?)\(', name)
    return (m.group(1) if m else name)[:86]


def main():
    cc = B.hipcc()
    print('# %s %s -Rpass-analysis=kernel-resource-usage (one row per kernel; spill = scratch bytes per lane + spilled VGPRs + spilled SGPRs; LDS = static bytes per workgroup; Occ = waves per SIMD the register / LDS budget allows)' % (
        os.path.basename(cc), ' '.join(B.FLAGS)))
    print('%-18s %-86s %5s %5s %5s %7s %7s %4s' % ('file', 'kernel', 'VGPR', 'AGPR', 'SGPR', 'spill', 'LDS B', 'Occ'))
    for src in B.SOURCES:
        cmd = [cc] + B.FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(B.HERE, src),
                                '-o', '/dev/null']
        r = subprocess.run(cmd, capture_output=True, text=True)
        cur = None
        rows = {}
        for line in r.stderr.splitlines():
            m = re.search(r'remark: Function Name: (\S+)', line)
            if m:
                dem = subprocess.run(['c++filt', m.group(1)], capture_output=True,
                                     text=True).stdout.strip()
                cur = rows.setdefault(short(dem), {})
                continue
            m = re.search(r'remark:\s+([\w ]+?)(?: \[[^\]]+\])?: (\d+)', line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
        for k, v in rows.items():
            if 'VGPRs' not in v:
                continue
            print('%-18s %-86s %5d %5d %5d %7d %7d %4d' % (
                src, k, v.get('VGPRs', 0), v.get('AGPRs', 0), v.get('TotalSGPRs', 0),
                v.get('ScratchSize', 0) + v.get('VGPRs Spill', 0) + v.get('SGPRs Spill', 0),
                v.get('LDS Size', 0), v.get('Occupancy', 0)))


if __name__ == '__main__':
    main()
