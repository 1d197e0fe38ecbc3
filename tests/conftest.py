"""pytest configuration: registers the ``gpu`` marker and shared helpers."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# BASELINE.json's configurations, by the tests that hold each to the oracle at the size the benchmark is quoted on.  They run
# FIRST and each in a child process of its own (tests/test_gpu_00_configs.py, which says why); the main session therefore takes
# them out of their home modules, a child (D2D_CONFIG_CHILD set) keeps them.
CONFIG_TESTS = {
    "cfg1": ["tests/test_gpu_forward.py::test_cfg1_square_scene_64"],
    "cfg2": ["tests/test_gpu_fullmap.py::test_cfg2_full_map_against_committed_oracle_fixture",
             "tests/test_gpu_fullmap.py::test_cfg2_sigmoid_full_map_properties"],
    "cfg3": ["tests/test_gpu_grad.py::test_cfg3_full_size_value_and_grad",
             "tests/test_gpu_grad.py::test_cfg3_full_map_nan_positions_equal_the_exhaustive_kernels",
             "tests/test_gpu_grad.py::test_cfg3_rows_against_the_c_gradient_oracle",
             "tests/test_gpu_grad.py::test_nan_scan_with_a_full_queue_and_a_full_list"],
    "cfg4": ["tests/test_gpu_forward.py::test_cfg4_full_size_against_sampled_oracle_cells",
             "tests/test_gpu_forward.py::test_cfg4_full_size_against_contiguous_oracle_blocks"],
    "cfg5": ["tests/test_gpu_opt.py::test_cfg5_full_size_value_and_gradient_on_sampled_cells",
             "tests/test_gpu_opt.py::test_cfg5_full_map_against_the_c_oracle"],
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("D2D_CONFIG_CHILD"):
        return
    moved = {n for nodes in CONFIG_TESTS.values() for n in nodes}
    keep, gone = [], []
    for it in items:
        base = it.nodeid.split("[")[0]
        (gone if base in moved else keep).append(it)
    if gone:
        items[:] = keep
        config.hook.pytest_deselected(items=gone)


def pytest_sessionstart(session):
    """libd2d.so must bind the SYSTEM libamdhip64: torch (imported by the autodiff oracle) bundles its own copy under the
    same SONAME, and a process that loads torch's first reports "no ROCm-capable device" to libd2d afterwards.  Load the
    library and initialise the HIP runtime before any test gets to import torch, whatever the test order or -k selection."""
    try:
        from differt2d_amd import _lib

        if os.path.exists(_lib.LIB_PATH):
            _lib.device_count()
    except Exception:  # noqa: BLE001 -- CPU box without the library built yet: the tests that need it say so themselves
        pass


def random_scene(n_walls: int, seed: int = 1234):
    """Layout of Scene.random_uniform_scene (reference scene.py:718-733) with a NumPy PRNG
    (jax.random is unavailable): pts[0] = tx, wall i = pts[1+2i : 3+2i]."""
    pts = np.random.default_rng(seed).random((1 + 2 * n_walls + 1, 2), dtype=np.float32)
    tx = pts[0].copy()
    walls = pts[1 : 1 + 2 * n_walls].reshape(n_walls, 2, 2).copy()
    return tx, walls


def unit_grid(n: int, m: int | None = None):
    m = n if m is None else m
    x = np.linspace(0.0, 1.0, n).astype(np.float32)
    y = np.linspace(0.0, 1.0, m).astype(np.float32)
    return np.meshgrid(x, y)


@pytest.fixture(scope="session")
def seed() -> int:
    return 1234
