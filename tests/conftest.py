import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graspnet-baseline_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def model_oracle(oracle):
    from oracle import model_oracle as MO
    return MO


@pytest.fixture(scope="session")
def state_dict():
    from graspnet_amd import synthetic as S
    return S.make_state_dict(0)


def pytest_sessionfinish(session, exitstatus):
    """Model-level parity cases record what they measured (tests/parity_util.REPORT); on the GPU box the record goes to
    gpurun_out/parity_tests.json, from where it is copied into profiles/parity_r02.json."""
    try:
        import json
        import parity_util as PU
        if PU.REPORT:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_tests.json"), "w") as f:
                json.dump({"rule": "|hip - oracle| <= 1e-4 * max(1, ||oracle key||_inf); raw head outputs <= 1e-4 absolute",
                           "cases": PU.REPORT}, f, indent=1)
    except Exception as e:      # never turn a finished run red over the report
        print("parity report not written:", e)
