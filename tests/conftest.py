import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """A fresh checkout has no libmdx.so (built files are kept out of the history): build it
    once where hipcc is at hand, so that the symbol and host-logic tests have the library
    they check.  An existing library is left alone, whatever its age."""
    import shutil

    from lammps_analysis_amd import build
    if not build.LIB_PATH.exists() and (shutil.which("hipcc") or
                                        pathlib.Path("/opt/rocm/bin/hipcc").exists()):
        build.build_library()


def _gpu_available() -> bool:
    try:
        from lammps_analysis_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """Fails (not skips) when a GPU test runs without libmdx or without a device."""
    from lammps_analysis_amd import _lib
    _lib.load()
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the GPU box")
    from lammps_analysis_amd import engine
    engine.set_device(0)
    return engine
