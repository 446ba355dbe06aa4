"""GPU tier: short runs of the soak programs (tests/soak/*_soak.py) -- the long runs of round 5 are in profiles/r05j ... r05o_*.log.  Each tool
compares the product with itself across configurations (kernel variants, hand-over forms, host arrangements) or with the oracle on
random cases; the variant sweep is the one that found the zero-fill race of late allocations (docs/LABBOOK.md, round 5)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name, *args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak", name)] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


def test_kernel_variants_agree_on_random_sizes():
    assert "mismatches 0" in _tool("variant_sweep_soak.py", 60, 1300)


def test_every_hand_over_form_gives_the_same_bits():
    assert "mismatches 0" in _tool("surface_sweep_soak.py", 100)


def test_cli_host_arrangements_print_the_same_bytes():
    assert "mismatches 0" in _tool("cli_sweep_soak.py", 5)


def test_abi_survives_random_arguments():
    assert "no crash" in _tool("abi_fuzz_soak.py", 4000)


def test_deferred_api_under_repetition():
    assert "deferred soak ok" in _tool("deferred_soak.py", 3000)


def test_one_engine_through_random_state_changes():
    assert "mismatches 0" in _tool("engine_state_soak.py", 1500, 333, 203)


def test_ssim_kernels_against_the_float64_twin_on_random_sizes():
    assert "mismatches 0" in _tool("ssim_twin_sweep_soak.py", 16, 500)


def test_rgb_frames_with_random_pitches():
    assert "mismatches 0" in _tool("rgb_pitch_sweep_soak.py", 120)
