"""Build libcarta1_hip.so (and the N-API addon when node headers are present) in-tree.

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container as well as on the
GPU box.  The built library lives at carta1_amd/lib/libcarta1_hip.so (git-ignored; it travels to the
GPU box with the gpurun snapshot).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'lib', 'libcarta1_hip.so')


def build_library(force=False):
    cmd = ['make', '-s', '-C', os.path.join(HERE, 'csrc')]
    if force:
        cmd.append('-B')
    subprocess.check_call(cmd)
    if not os.path.exists(LIB):
        raise RuntimeError('build did not produce ' + LIB)
    return LIB


def build_addon(force=False):
    d = os.path.join(HERE, 'js', 'addon')
    if not os.path.exists(os.path.join(d, 'Makefile')):
        return None
    cmd = ['make', '-s', '-C', d]
    if force:
        cmd.append('-B')
    subprocess.check_call(cmd)
    return os.path.join(HERE, 'js', 'addon', 'carta1_napi.node')


if __name__ == '__main__':
    print(build_library())
    print(build_addon())
