import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc cross-compiles
    # without a GPU) and the CPU oracle once, exactly as the driver's build check does
    lib = os.path.join(ROOT, "multi-robot-fabrics_amd", "csrc", "libmrf_hip.so")
    ora = os.path.join(ROOT, "oracle", "libmrf_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


def pytest_collection_modifyitems(config, items):
    """The default build carries neither the float32 kernels (-DMRF_WITH_F32) nor the wave-pair rollout kernel
    (-DMRF_WITH_WP): test cases parametrized with scalar = abi.F32 / dtype = "f32" / kernel = 3 are skipped then; tests that
    loop over the variants ask abi.has_f32() / abi.has_wp() themselves, inside the test."""
    from multi_robot_fabrics_amd import abi
    try:
        have_f32, have_wp = abi.has_f32(), abi.has_wp()
    except Exception:       # noqa: BLE001 -- no library at collection time: let the tests report that
        return
    skip_f32 = pytest.mark.skip(reason="library built without float32 kernels (MRF_WITH_F32=1 python __graft_entry__.py)")
    skip_wp = pytest.mark.skip(reason="library built without the wave-pair kernel (MRF_WITH_WP=1 python __graft_entry__.py)")
    for item in items:
        params = getattr(getattr(item, "callspec", None), "params", {})
        if not have_f32 and (params.get("scalar") == abi.F32 or params.get("dtype") == "f32"):
            item.add_marker(skip_f32)
        if not have_wp and params.get("kernel") == 3:
            item.add_marker(skip_wp)
