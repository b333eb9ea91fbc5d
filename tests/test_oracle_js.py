"""CPU: the JavaScript restatement of the encode path (oracle/js/atrac1_oracle.mjs, the Node CPU baseline of bench.py) reproduces
the reference's golden vectors byte for byte: the config-1 known answer, 64-frame stereo runs with fixed modes [0,0,0] and
[2,2,3] and with transient detection (white and pink + bursts), and a frame range encoded from its 2-frame history."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
node = shutil.which('node')
pytestmark = pytest.mark.skipif(node is None, reason='node is not installed')


def test_js_restatement_reproduces_the_golden_vectors():
    r = subprocess.run([node, os.path.join(ROOT, 'oracle', 'js', 'cpu_baseline.mjs'), '--check'], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and out['parity'] is True and all(out['checks'].values()), out
    assert set(out['checks']) >= {'config1', 'white_m000_b1', 'white_m223_b1', 'white_detect', 'pinkT_detect', 'range_from_history'}
