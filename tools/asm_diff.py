"""Is the device code of the kernels unchanged?  Compiles every HIP source of passl_amd/csrc at a given commit and in
the working tree to gfx950 assembly (same flags as the build) and compares the instruction streams kernel by kernel
(comments, debug lines and basic-block label numbers ignored).  Runs without a GPU.

    python tools/asm_diff.py <commit>        # e.g. the last commit whose build passed the GPU suite

Used to show that an opt-in kernel added next to the product kernels (a new template parameter with a default, a new
kernel in the same file) left the product kernels' code bit-identical."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from passl_amd.csrc import build as B          # noqa: E402


def kernels(path):
    out, cur, buf = {}, None, []
    for line in open(path).read().split('\n'):
        m = re.match(r'^(_Z[\w$.]+):', line)
        if m and cur is None:
            cur, buf = m.group(1), []
            continue
        if cur is not None:
            if re.match(r'^\.Lfunc_end\d+:', line):
                out[cur] = buf
                cur = None
                continue
            s = re.sub(r'\s*;.*$', '', line).rstrip()
            if s.strip() and not s.strip().startswith(('.loc', '.file', '.cfi')):
                buf.append(re.sub(r'\.LBB\d+_', '.LBB_', s))
    return out


def assemble(src, dst):
    cmd = [B.hipcc()] + B.FLAGS + ['--cuda-device-only', '-S', src, '-o', dst]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])


def main():
    commit = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, 'old')
        os.makedirs(old)
        tar = subprocess.run(['git', '-C', ROOT, 'archive', commit, 'passl_amd/csrc', 'include'], capture_output=True, check=True)
        subprocess.run(['tar', '-x', '-C', old], input=tar.stdout, check=True)
        print('# device code of passl_amd/csrc at %s against the working tree (%s %s)' % (commit, os.path.basename(B.hipcc()), ' '.join(B.FLAGS)))
        changed = 0
        for s in B.SOURCES:
            po = os.path.join(old, 'passl_amd', 'csrc', s)
            if not os.path.exists(po):
                print('%-24s new file' % s)
                continue
            assemble(po, os.path.join(tmp, 'a.s'))
            assemble(os.path.join(B.HERE, s), os.path.join(tmp, 'b.s'))
            a, b = kernels(os.path.join(tmp, 'a.s')), kernels(os.path.join(tmp, 'b.s'))
            diff = [n for n in a if n in b and a[n] != b[n]]
            gone = [n for n in a if n not in b]
            new = [n for n in b if n not in a]
            changed += len(diff) + len(gone)
            print('%-24s %3d kernels: %3d identical, %d different, %d removed, %d new' % (s, len(a), len(a) - len(diff) - len(gone), len(diff), len(gone), len(new)))
            for n in diff + gone:
                print('    ' + subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()[:140])
        print('# %s' % ('every kernel of the old build is unchanged' if changed == 0 else '%d kernels changed' % changed))


if __name__ == '__main__':
    main()
