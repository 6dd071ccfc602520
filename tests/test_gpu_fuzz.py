"""Bounded runs of the randomised parity sweep (tools/fuzz_parity.py) with fixed seeds: a few seconds per mode, so every
GPU test run also covers shapes nobody wrote down.  Longer sweeps: `python tools/fuzz_parity.py <seconds> <seed> <mode>`."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(fuzz, mode, seconds, seed, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["fuzz_parity.py", str(seconds), str(seed), mode])
    return fuzz.main()


@pytest.mark.parametrize("mode,seconds", [("calib", 10), ("shard", 8), ("flow", 8), ("roi", 5), ("api", 6), ("run", 8)])
def test_randomised_parity(fuzz, mode, seconds, monkeypatch):
    assert _run(fuzz, mode, seconds, 2026, monkeypatch) == 0
