"""The JavaScript host side (N-API addon, ColumnGrid shim, flow driver) under Node."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "fem_amd", "js", "colgrid_napi.node")

needs_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is not available")


def _run(script):
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", script)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout


@needs_node
def test_js_cpu():
    assert "js cpu tests ok" in _run("test_cpu.js")


@needs_node
@pytest.mark.gpu
def test_js_gpu():
    assert "js gpu tests ok" in _run("test_gpu.js")
