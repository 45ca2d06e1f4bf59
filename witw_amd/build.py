"""Build libwitw_hip.so (the C-ABI shared library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo
snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libwitw_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile and link; serialised across processes with a file lock (N ranks may import at once)."""
    if not force and not stale():
        return LIB
    import fcntl
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    with open(os.path.join(HERE, 'build', '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():      # another process built it while we waited
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, 'build', os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    tmp = LIB + '.tmp.%d' % os.getpid()
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)                       # atomic: a concurrent dlopen never sees a half-written file
    return LIB


HOST_SRC = os.path.join(HERE, 'csrc_host', 'jpeg_coef.cpp')
HOST_LIB = os.path.join(HERE, 'libwitw_jpeg.so')


def build_host(force=False, verbose=True):
    """libwitw_jpeg.so: the host-only part of the data path (JPEG entropy decoding, csrc_host/jpeg_coef.cpp), plain g++ --
    DataLoader workers load it without ever mapping the HIP runtime."""
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= os.path.getmtime(HOST_SRC):
        return HOST_LIB
    import fcntl
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    with open(os.path.join(HERE, 'build', '.lock_host'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or not os.path.exists(HOST_LIB) or os.path.getmtime(HOST_LIB) < os.path.getmtime(HOST_SRC):
                tmp = HOST_LIB + '.tmp.%d' % os.getpid()
                cmd = [os.environ.get('CXX', 'g++'), '-O3', '-fPIC', '-shared', '-std=c++17', '-Wall', '-o', tmp, HOST_SRC]
                if verbose:
                    print(' '.join(cmd), flush=True)
                subprocess.check_call(cmd)
                os.replace(tmp, HOST_LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return HOST_LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    build_host(force='--force' in sys.argv)
    print(LIB)
    print(HOST_LIB)
