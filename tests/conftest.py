import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # gpu tests are skipped automatically where no GPU is visible (the CPU container)
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this environment')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """measured parity margins of this session -> gpurun_out/parity_report.json (merged with what earlier sessions of the same
    GPU call left there; tests/common.py PARITY)"""
    try:
        from tests import common as C
    except Exception:
        return
    if not C.PARITY:
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, 'parity_report.json')
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except (OSError, ValueError):
            data = {}
    for k, v in C.PARITY.items():
        data.setdefault(k, {}).update(v)
    with open(path, 'w') as f:
        json.dump(data, f, indent=1, sort_keys=True)
