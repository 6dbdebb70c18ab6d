"""Build libqk_hip.so (the C-ABI library of include/qk.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.realpath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libqk_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fPIC',
         '-Wno-unused-value']


def _newest_header():
    deps = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc'))
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'qk.h'))
    return max(os.path.getmtime(p) for p in deps)


def _compile(src, obj):
    subprocess.check_call(['hipcc'] + FLAGS + ['-c', src, '-o', obj])
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdr = _newest_header()
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            todo.append((s, o))
    if todo:
        if verbose:
            print('hipcc: compiling %d file(s) for gfx950' % len(todo))
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda so: _compile(*so), todo))
    if todo or not os.path.exists(LIB):
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB])
    return LIB


if __name__ == '__main__':
    print(build(verbose=True))
