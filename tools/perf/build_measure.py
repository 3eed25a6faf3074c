"""A MEASUREMENT copy of the package under build/measure/: the same sources compiled with
-DMEMB_HIP_MEASURE, which is what makes the `debug` option / MEMB_HIP_DEBUG switches exist (skip the
decode, skip the output, store policies ...: hip_trained_kernels.h). The shipped library in memb_amd/
is never built that way; tools/perf/ab3.py and friends pick the copy up through MEMB_PACKAGE_ROOT:

    python tools/perf/build_measure.py [extra hipcc flags]
    MEMB_PACKAGE_ROOT=build/measure python tools/perf/ab3.py ...

build/ is git-ignored and travels to the GPU box with the snapshot.
"""
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def build(extra_flags=(), root=None, plain=False):
    import build_native
    root = root or os.path.join(REPO, 'build', 'measure')
    package = os.path.join(root, 'memb_amd')
    if os.path.isdir(package):
        shutil.rmtree(package)
    shutil.copytree(os.path.join(REPO, 'memb_amd'), package,
                    ignore=shutil.ignore_patterns('*.so', '__pycache__', '*.pyc'))
    include = os.path.join(root, 'include')
    if os.path.isdir(include):
        shutil.rmtree(include)
    shutil.copytree(os.path.join(REPO, 'include'), include)
    # point build_native at the copy
    build_native.PACKAGE_DIR = package
    build_native.CSRC = os.path.join(package, 'csrc')
    build_native.INCLUDE = include
    build_native.HIP_LIBRARY = os.path.join(package, os.path.basename(build_native.HIP_LIBRARY))
    build_native.EXTENSION = os.path.join(package, os.path.basename(build_native.EXTENSION))
    build_native.build_hip_library(force=True, extra_flags=[*([] if plain else ['-DMEMB_HIP_MEASURE']), *extra_flags])
    build_native.build_extension(force=False)
    return root


if __name__ == '__main__':
    # --plain=DIR: a copy WITHOUT the measurement switches (two shipped builds side by side, e.g. -DMEMB_HIP_SGPRS=0)
    arguments = sys.argv[1:]
    plain_root = next((a.split('=', 1)[1] for a in arguments if a.startswith('--plain=')), None)
    arguments = [a for a in arguments if not a.startswith('--plain=')]
    measure_root = next((a.split('=', 1)[1] for a in arguments if a.startswith('--root=')), None)
    arguments = [a for a in arguments if not a.startswith('--root=')]
    if plain_root:
        print('plain package:', build(arguments, root=os.path.abspath(plain_root), plain=True))
    else:
        print('measurement package:', build(arguments, root=os.path.abspath(measure_root) if measure_root else None))
