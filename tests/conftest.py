import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs through libvgen_hip.so)")
    config.addinivalue_line("markers", "reference: needs the read-only reference tree (/root/reference)")
    # CPU suite: a bounded thread count.  The emulator's many small torch ops fork / join on every one of the box's threads
    # by default; with a second pytest (or any other load) on an 8-core box that oversubscription made one session test take
    # 9 minutes instead of 9 seconds.  VGEN_TEST_THREADS overrides; GPU boxes keep torch's default
    # (the calibration pass there sizes its own LAPACK threads).
    if not torch.cuda.is_available() or os.environ.get("VGEN_TEST_THREADS"):
        torch.set_num_threads(max(1, int(os.environ.get("VGEN_TEST_THREADS", min(os.cpu_count() or 1, 4)))))


def pytest_collection_modifyitems(config, items):
    from oracle import ref_import
    have_ref = ref_import.available()
    have_gpu = torch.cuda.is_available()
    for it in items:
        if "reference" in it.keywords and not have_ref:
            it.add_marker(pytest.mark.skip(reason="reference tree not present on this box"))
        if "gpu" in it.keywords and not have_gpu:
            it.add_marker(pytest.mark.skip(reason="no GPU"))


def gold(name):
    return torch.load(os.path.join(GOLD, name), map_location="cpu", weights_only=False)


@pytest.fixture
def emu_backend():
    """Installs the CPU emulator of the C ABI for host-logic tests (test double, never product)."""
    from oracle.abi_emulator import EmuBackend
    from vgen_amd import ops
    prev = ops.set_backend(EmuBackend())
    # ONE thread for everything that runs on the emulator: many of these tests compare two evaluations with different batch
    # shapes bit for bit (see `one_thread`), the models are tiny (a threaded fork / join per op costs more than it saves), and
    # a second pytest on the box can no longer oversubscribe the cores
    n = torch.get_num_threads()
    torch.set_num_threads(int(os.environ.get("VGEN_EMU_THREADS", "1")))
    yield
    torch.set_num_threads(n)
    ops.set_backend(prev)


@pytest.fixture
def one_thread():
    """Bit-level comparisons of two emulator evaluations with DIFFERENT batch shapes (a batch of units vs one forward per
    unit): a threaded sgemm partitions its K loop by the matrix shape, the last bits of the fp32 sums then differ and a few
    16-bit roundings downstream flip (1e-3 on the tiny model with 4 threads, 0 with 1, 2 or 8).  One thread: one order."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


@pytest.fixture
def hip_backend():
    from vgen_amd import ops
    prev = ops.set_backend(None)          # force the real HipBackend
    be = ops.backend()
    assert be.name == "hip"
    yield be
    ops.set_backend(prev)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
