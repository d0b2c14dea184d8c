"""
Build libcsrk.so (hand-written HIP kernels + the C ABI of include/csrk.h) for gfx950.

    python csr_amd/build.py [--force]      (run as a script: importing the package needs the library)

hipcc cross-compiles without a GPU.  The library is built IN-TREE (csr_amd/libcsrk.so) so
it travels with the repo snapshot; it is git-ignored.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libcsrk.so')
ARCH = 'gfx950'


def sources():
    return sorted(glob.glob(os.path.join(SRC_DIR, '*.hip')))


def _deps():
    return sources() + glob.glob(os.path.join(SRC_DIR, '*.h')) + \
        [os.path.join(os.path.dirname(HERE), 'include', 'csrk.h')]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    objs = []
    procs = []
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(src), *[os.path.getmtime(h) for h in _deps() if h.endswith('.h')]):
            continue
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
               '-Wall', '-Wno-unused-function', '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {src}')
        if verbose and out:
            sys.stderr.write(out.decode())
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
