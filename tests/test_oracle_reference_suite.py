"""
Pins the ORACLE against the reference's own test-suite: every test of tests/test_reference_suite.py (the re-statement of
/root/reference/tests/test_linear_exprs.py that the GPU suite runs against the HIP library) is run here with the CPU oracle
answering behind the package's Python layer (tests/mock_device, swapped in for this module only).  What passes here is the
checker -- same frames, same scikit-learn / NumPy references, same tolerances as the reference's tests -- and the host-side
marshalling of polars_ds_extension_amd/lstsq.py; it says nothing about the kernels (`-m gpu` does that).
"""
import inspect
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import test_linear_models as M  # noqa: E402
import test_reference_suite as R  # noqa: E402

TESTS = [n for n, f in vars(R).items() if n.startswith("test_") and callable(f)]
MODEL_TESTS = ["test_lr", "test_online_lr", "test_elastic_net", "test_glm_family", "test_glm_convergence_failure"]  # the reference's tests/test_linear_models.py, host (NumPy) data


@pytest.fixture(scope="module")
def oracle_backed_package(orc):
    """The package with the mock device library in place of libpds_lstsq_hip.so -- for this module only."""
    import threading

    from mock_device import device

    import polars_ds_extension_amd as m
    from polars_ds_extension_amd import _lib, lstsq

    saved_lib, saved_tls = _lib._lib, lstsq._tls if hasattr(lstsq, "_tls") else None
    _lib._lib = device.load_for_package()
    # the mock answers HOST pointers only: while it stands in, the package must see no device (on a GPU box its model classes would
    # otherwise hand it device buffers) -- the test double adapts the environment, the product does not know about the double
    import torch

    saved_avail = torch.cuda.is_available
    torch.cuda.is_available = lambda: False
    if saved_tls is not None:
        lstsq._tls = threading.local()
    try:
        yield m
    finally:
        torch.cuda.is_available = saved_avail
        _lib._lib = saved_lib
        if saved_tls is not None:
            lstsq._tls = saved_tls
        m.config.LIN_REG_EXPR_F64 = True


def _params_of(fn):
    mark = [mk for mk in getattr(fn, "pytestmark", []) if mk.name == "parametrize"]
    return mark


def _cases():
    out = []
    for name in TESTS + MODEL_TESTS:
        fn = getattr(R, name, None) or getattr(M, name)
        sig = list(inspect.signature(fn).parameters)
        grids = [[{}]]
        for mk in _params_of(fn):
            keys = [k.strip() for k in mk.args[0].split(",")]
            vals = [v if isinstance(v, (tuple, list)) and len(keys) > 1 else (v,) for v in mk.args[1]]
            grids.append([dict(zip(keys, v)) for v in vals])
        if "lin_reg_dtype" in sig:
            grids.append([{"lin_reg_dtype": "f64"}, {"lin_reg_dtype": "f32"}])
        import itertools

        for combo in itertools.product(*grids):
            kw = {}
            for d in combo:
                kw.update(d)
            out.append(pytest.param(name, kw, id=name + ("[" + "-".join(str(v) for v in kw.values()) + "]" if kw else "")))
    return out


@pytest.mark.parametrize("name,kwargs", _cases())
def test_reference_suite_on_the_oracle(name, kwargs, oracle_backed_package):
    m = oracle_backed_package
    fn = getattr(R, name, None) or getattr(M, name)
    kw = dict(kwargs)
    if "lin_reg_dtype" in kw:
        m.config.LIN_REG_EXPR_F64 = kw["lin_reg_dtype"] == "f64"
    try:
        args = {}
        for p in inspect.signature(fn).parameters:
            args[p] = m if p == "pds" else kw[p]
        fn(**args)
    finally:
        m.config.LIN_REG_EXPR_F64 = True
