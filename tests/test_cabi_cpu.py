"""
CPU-side checks of the drop-in boundary (no GPU, no compute calls): the shared library loads, exports every
symbol include/pds_lstsq.h declares, fails loudly without a device, and its host-side special functions
are bit-identical to the oracle's restatement of src/stats_utils.
"""
import ctypes as C
import re
import sys
from pathlib import Path

import numpy as np
import pytest
from scipy import stats

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, str(ROOT))
    from polars_ds_extension_amd import _build, _lib

    if not _lib.LIB_PATH.exists():
        _build.build()
    return _lib


def test_header_symbols_are_exported(lib):
    header = (ROOT / "include" / "pds_lstsq.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pds_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 28
    so = lib.load()
    missing = [s for s in declared if not hasattr(so, s)]
    assert not missing, f"declared in include/pds_lstsq.h but not exported: {missing}"
    assert sorted(lib.EXPORTS) == declared
    assert lib.missing_exports() == []
    assert b"gfx950" in so.pds_version()


def test_no_cpu_fallback_without_device(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from polars_ds_extension_amd import lstsq

    with pytest.raises(lib.PdsError) as e:
        lstsq.Context(0)
    assert e.value.code == -4  # PDS_ERR_HIP: the product path never computes on the CPU
    with pytest.raises(lib.PdsError):
        lstsq.lin_reg(np.ones(10), target=np.ones(10))


def test_host_argument_validation_mirrors_reference(lib):
    from polars_ds_extension_amd import lstsq

    with pytest.raises(ValueError, match="max_iter"):
        lstsq.lin_reg(np.ones(5), target=np.ones(5), max_iter=0)  # expr_linear.py:231-232
    with pytest.raises(ValueError, match="window_size"):
        lstsq.rolling_lin_reg(np.ones(5), target=np.ones(5), window_size=1)  # :524-526
    with pytest.raises(ValueError, match="features"):
        lstsq.rolling_lin_reg(np.ones(5), np.ones(5), np.ones(5), target=np.ones(5), window_size=2)  # :527-531
    with pytest.raises(ValueError, match="features"):
        lstsq.recursive_lin_reg(np.ones(5), np.ones(5), target=np.ones(5), start_with=1)


def test_lr_params_struct_layout(lib):
    # pds_lr_params: int, double x3, int x3, double -- natural alignment on x86-64
    assert C.sizeof(lib.LRParams) == 56
    assert lib.LRParams.l1_reg.offset == 8 and lib.LRParams.solver.offset == 32 and lib.LRParams.singular_x_tol.offset == 48


def test_special_functions_bit_identical_to_oracle(lib, orc):
    so = lib.load()
    for dof in (3.0, 17.0, 4995.0, 119_991.0, 9_999_984.0):
        for t in (0.0, 0.37, 1.0, 1.96, 3.3, 25.0):
            assert so.pds_student_t_sf(t, dof) == orc.student_t_sf(t, dof)
        assert so.pds_student_t_ppf(0.975, dof) == orc.student_t_ppf(0.975, dof)
    # deliberate deviation: where the reference's AS109 loop never terminates (dof > ~1.42e7) the library
    # switches to the Cornish-Fisher expansion instead of hanging
    for dof in (2e7, 1e8 - 16, 1e10):
        v = so.pds_student_t_ppf(0.975, dof)
        assert abs(v / stats.t.ppf(0.975, dof) - 1) < 1e-12
