"""Repository rules: the product never touches the oracle; tests never read /root/reference."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(os.path.join(ROOT, d)):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                yield os.path.join(base, f)


def test_product_does_not_import_oracle():
    for path in _py_files('thesis_clip_nerf_amd'):
        text = open(path).read()
        if path.endswith('.py'):
            assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), path
            assert 'mvnerf_oracle' not in text and 'import_module' not in text, path
        else:                                   # C/HIP sources may cite the oracle in comments only
            assert not re.search(r'#include\s*[<"][^>"]*oracle', text), path


def test_bench_uses_oracle_only_in_cpu_baseline():
    text = open(os.path.join(ROOT, 'bench.py')).read()
    hits = [m.start() for m in re.finditer(r'from oracle|import oracle', text)]
    start = text.index('def cpu_baseline')
    end = text.index("if __name__ == '__main__'")
    assert hits and all(start < h < end for h in hits)       # the NumPy oracle and its torch twin, inside cpu_baseline() only


def test_runtime_code_does_not_read_reference():
    for d in ('thesis_clip_nerf_amd', 'oracle'):
        for path in _py_files(d):
            assert '/root/reference' not in open(path).read().replace('/root/reference/src/lib)', ''), path
    for f in ('bench.py', '__graft_entry__.py'):
        assert '/root/reference' not in open(os.path.join(ROOT, f)).read()
