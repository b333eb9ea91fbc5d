"""The JavaScript host (carta1_amd/js) + N-API addon: host-side checks on CPU, and on the GPU the
reference's API -- encode()/decode() closures, encodeAeaPcm/decodeAeaPcm, AudioProcessor streams --
against the reference's golden vectors.  Skipped when node is not installed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, 'carta1_amd', 'js')

node = shutil.which('node')
pytestmark = pytest.mark.skipif(node is None, reason='node is not installed')


def _run(args):
    from carta1_amd import build
    build.build_library()
    build.build_addon()
    p = subprocess.run([node, 'selftest.mjs'] + args, cwd=JS, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=600)
    return p.returncode, p.stdout


def test_js_host_side_on_cpu():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip('CPU-side variant (expects no device)')
    except ImportError:
        pass
    rc, out = _run([])
    assert rc == 0 and 'ALL OK' in out, out


@pytest.mark.gpu
def test_js_host_api_against_reference_on_gpu():
    rc, out = _run(['--gpu'])
    assert rc == 0 and 'ALL OK' in out, out
