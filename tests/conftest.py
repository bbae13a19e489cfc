import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a checkout without build artefacts (they are git-ignored): build once, the way __graft_entry__.build() does
    need = [os.path.join(ROOT, "bramble_amd", "libbramble_amd.so"), os.path.join(ROOT, "bramble_amd", "libbramble_synth.so"),
            os.path.join(ROOT, "bramble_amd", "bin", "bramble"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)
