"""Build libpassl_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/passl_hip.h).

    python -m passl_amd.csrc.build          # incremental
    python -m passl_amd.csrc.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), 'lib')
LIB = os.path.join(OUT_DIR, 'libpassl_hip.so')
SOURCES = ['runtime.hip', 'plan.hip', 'flat.hip', 'layout_pool.hip', 'stem_pool.hip', 'bn.hip', 'head.hip', 'ntxent.hip', 'vit.hip', 'attention.hip', 'attention_bf16.hip', 'clip.hip', 'clas.hip',
           'conv_igemm.hip', 'conv3x3_wave.hip', 'conv_igemm_ring.hip', 'conv_igemm_8p.hip', 'conv_stem.hip', 'conv_wgrad.hip']
HEADERS = ['common.h', 'plan.h', 'prof.h', 'igemm_epi.h', 'igemm_dma.h', 'halo_geom.h', 'wgrad_halo_geom.h', 'conv_wgrad_halo.inc', os.path.join('..', '..', 'include', 'passl_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-Wno-unused-result']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    obj_dir = os.path.join(OUT_DIR, 'obj')
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    cc = hipcc()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(obj_dir, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([cc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr[-4000:]))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
