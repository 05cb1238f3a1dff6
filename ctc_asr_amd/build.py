"""Builds ``libctcasr.so`` (the C-ABI library of ``include/ctcasr.h``) for gfx950 with hipcc.

In-tree, explicit ``hipcc -shared -fPIC``: the built library travels with the repository
snapshot to the GPU box; no JIT cache is involved.  hipcc cross-compiles without a GPU.
"""

import glob
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libctcasr.so')
HOST_LIB_PATH = os.path.join(PKG_DIR, 'libctcasr_host.so')     # plain C helpers, no device code
HOST_SOURCES = [os.path.join(PKG_DIR, 'host', 'crc32c.c')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# per-file additions (dgrad16.hip: see the note at its `fold1`)
FILE_FLAGS = {'dgrad16.hip': ['-fno-slp-vectorize']}


def sources():
    return sorted(glob.glob(os.path.join(CSRC_DIR, '*.hip')))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC_DIR, '*.h')) + \
        [os.path.join(PKG_DIR, '..', 'include', 'ctcasr.h'), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > built for d in deps)


def build_host(force=False, verbose=True):
    """gcc over ``host/*.c`` -> ``libctcasr_host.so`` (``include/ctcasr_host.h``)."""
    deps = HOST_SOURCES + [os.path.join(PKG_DIR, '..', 'include', 'ctcasr_host.h')]
    if not force and os.path.exists(HOST_LIB_PATH) and \
            all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB_PATH) for d in deps):
        return HOST_LIB_PATH
    cmd = [os.environ.get('CC', 'gcc'), '-O2', '-std=c99', '-fPIC', '-shared', '-Wall', '-o',
           HOST_LIB_PATH] + HOST_SOURCES
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HOST_LIB_PATH


def build(force=False, verbose=True):
    """Compile every ``csrc/*.hip`` into one shared object (and the host helper library);
    returns the device library's path."""
    build_host(force, verbose)
    if not force and not _stale():
        return LIB_PATH
    obj_dir = os.path.join(PKG_DIR, 'csrc', '_obj')
    os.makedirs(obj_dir, exist_ok=True)
    objects = []
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + '.o')
        objects.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(
                os.path.getmtime(src),
                max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC_DIR, '*.h'))),
                os.path.getmtime(os.path.join(PKG_DIR, '..', 'include', 'ctcasr.h'))):
            continue
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in procs:
        if proc.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objects
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
