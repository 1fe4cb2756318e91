#!/usr/bin/env python3
"""Register / LDS / scratch report of every kernel in the library (hipcc -Rpass-analysis=kernel-resource-usage, gfx950):
    python scripts/kernel_resources.py > profiles/rNN_kernel_resources.md
VGPRs counts the unified file (architectural + accumulation registers; rocprofv3's `arch_vgpr_count` shows the
architectural half only, e.g. 128 for a 256-register kernel)."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'thesis_clip_nerf_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt',
         '-Wno-unused-function', '-Rpass-analysis=kernel-resource-usage', '-c', '-o', '/dev/null']


def main():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    print('| file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/block | waves/SIMD |')
    print('|---|---|---|---|---|---|---|---|')
    for src in srcs:
        out = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [src], cwd=CSRC, capture_output=True, text=True).stderr
        cur = None
        rows = {}
        for ln in out.splitlines():
            m = re.search(r'remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/\w+\])?(?: \[waves/SIMD\])?: (\S+) \[-Rpass', ln)
            if not m:
                continue
            key, val = m.group(1).strip(), m.group(2)
            if key == 'Function Name':
                # llvm-cxxfilt knows the __bf16 mangling (DF16b); kernels inside an anonymous namespace demangle to
                # "mvnerf::(anonymous namespace)::name<...>(args)": drop that qualifier BEFORE cutting the argument list at the first "("
                filt = '/opt/rocm/lib/llvm/bin/llvm-cxxfilt' if os.path.exists('/opt/rocm/lib/llvm/bin/llvm-cxxfilt') else 'c++filt'
                cur = subprocess.run([filt, val], capture_output=True, text=True).stdout.strip()
                cur = cur.replace('(anonymous namespace)::', '').replace('mvnerf::', '').replace('void ', '')
                cur = re.sub(r'\(.*$', '', cur) or val
                rows[cur] = {}
            elif cur:
                rows[cur][key] = val
        for k, r in rows.items():
            print(f"| {src} | `{k}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('ScratchSize')} | {r.get('LDS Size')} | {r.get('Occupancy')} |")


if __name__ == '__main__':
    sys.exit(main())
