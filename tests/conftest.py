import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# ---- arithmetic modes of the MFMA gather-GEMMs ---------------------------------------------------------------------
# Every `-m gpu` test that touches a GEMM runs once per mode, so the mode bench.py reports is a mode the suite checked:
#   f32     exact fp32 products (v_mfma_f32_32x32x2_f32): the north_star's 1e-4 RMS budget and the tighter per-op bounds
#           written at each assertion apply as they are;
#   bf16x3  fp32 operands split hi + lo, products carry ~2^-17 relative rounding instead of 2^-24: same bounds unless an
#           assertion names a `bf16x3=` one (whole-network GRADIENT comparisons, where train-mode batch statistics
#           amplify the product rounding, are the only ones that do);
#   bf16    operands rounded to bf16 (2^-9), the arithmetic of trainer.precision=bf16-mixed.  The fp32 oracle is then a
#           reference, not a parity target: bounds come from `bf16=` or the default below, and the parity statement is
#           tests/test_gpu_bf16_mixed.py (error against the fp32 oracle no larger than the CPU autocast oracle's own).
MODES = [m for m in os.environ.get("RFX_TEST_MODES", "f32,bf16x3,bf16").split(",") if m]
MODE_INDEPENDENT = {"test_gpu_norm", "test_gpu_stft", "test_gpu_lstm_kernel"}      # no GEMM mode inside: run once
CURRENT = {"mode": None, "test": ""}
_LOG = os.environ.get("RFX_TOL_LOG")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "one_mode: GPU test that does not depend on the GEMM arithmetic mode")


def pytest_generate_tests(metafunc):
    if "gemm_mode" not in metafunc.fixturenames or metafunc.definition.get_closest_marker("gpu") is None:
        return
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if mod in MODE_INDEPENDENT or metafunc.definition.get_closest_marker("one_mode") is not None:
        return
    metafunc.parametrize("gemm_mode", MODES, indirect=True)


@pytest.fixture(autouse=True)
def gemm_mode(request):
    mode = getattr(request, "param", None)
    CURRENT["test"] = request.node.nodeid
    if mode is None:
        CURRENT["mode"] = None
        yield None
        return
    from remfx_amd import ops
    prev = ops.gemm_precision()
    ops.set_gemm_precision(mode)
    CURRENT["mode"] = mode
    yield mode
    ops.set_gemm_precision(prev)
    CURRENT["mode"] = None


def mode():
    return CURRENT["mode"] or "f32"


def tol(f32, bf16x3=None, bf16=None):
    """Bound for the current mode.  bf16 default: 2e-2 relative for per-op / forward comparisons (operand rounding 2^-9,
    roughly 2e-3 of the output scale per layer, a few dozen layers deep), never tighter than 100 x the fp32 bound."""
    m = mode()
    if m == "f32":
        return f32
    if m == "bf16x3":
        return f32 if bf16x3 is None else bf16x3
    return bf16 if bf16 is not None else min(0.25, max(2e-2, 100.0 * f32))


def check(err, f32, scale=1.0, *, bf16x3=None, bf16=None, what=""):
    """assert err < tol(mode) * scale; with RFX_TOL_LOG=<file> every comparison is also appended there (mode, test, what,
    err / scale, bound) -- that log is where the per-mode bounds written in the tests come from."""
    t = tol(f32, bf16x3, bf16)
    err, scale = float(err), float(scale)
    if _LOG:
        with open(_LOG, "a") as f:
            f.write(json.dumps({"mode": mode(), "test": CURRENT["test"], "what": str(what), "err_rel": err / scale if scale else err,
                                "bound": t}) + "\n")
    assert err < t * scale, (what, mode(), err, t * scale)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
