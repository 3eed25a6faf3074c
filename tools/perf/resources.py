#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in libmemb_hip.so, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed).

    python tools/perf/resources.py [extra hipcc flags...]    e.g. -DMEMB_HIP_BOUNDS_WAVES=5
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SOURCE = os.path.join(REPO, 'memb_amd', 'csrc', 'memb_hip.hip')
FIELDS = ('VGPRs', 'AGPRs', 'TotalSGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs Spill',
          'VGPRs Spill', 'LDS Size [bytes/block]')


def main():
    command = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
               '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wno-unused-value',
               '-Wno-align-mismatch', '-Wno-pass-failed', '-Rpass-analysis=kernel-resource-usage',
               '-o', '/dev/null', SOURCE] + sys.argv[1:]
    output = subprocess.run(command, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    kernels = []
    current = None
    for line in output.splitlines():
        name = re.search(r'Function Name: (\S+)', line)
        if name:
            current = {'name': name.group(1)}
            kernels.append(current)
            continue
        for field in FIELDS:
            match = re.search(re.escape(field) + r': (\d+)', line)
            if match and current is not None and line.split('remark: ')[-1].strip().startswith(field):
                current[field] = int(match.group(1))
    demangled = subprocess.run(['c++filt'] + [k['name'] for k in kernels], stdout=subprocess.PIPE, text=True).stdout.splitlines()
    print('%-86s %5s %5s %7s %5s %6s' % ('kernel', 'VGPR', 'SGPR', 'scratch', 'occ', 'spills'))
    for kernel, pretty in zip(kernels, demangled):
        pretty = pretty.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        print('%-86s %5d %5d %7d %5d %6d' % (
            pretty[:86], kernel.get('VGPRs', -1), kernel.get('TotalSGPRs', -1), kernel.get('ScratchSize [bytes/lane]', -1),
            kernel.get('Occupancy [waves/SIMD]', -1), kernel.get('SGPRs Spill', 0) + kernel.get('VGPRs Spill', 0)))


if __name__ == '__main__':
    main()
