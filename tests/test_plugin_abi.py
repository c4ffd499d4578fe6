"""
The Polars plugin C ABI layer (csrc/plugin.cpp): pyarrow + ctypes play the engine (tests/plugin_harness.py).
CPU tests: symbol table, version, pickled-kwargs parser against real `pickle.dumps(..., protocol=5)` of the
reference's kwargs dicts, output-field functions.  GPU tests: Series in -> Series out through `_polars_plugin_*`,
compared with the oracle and with the reference's literal frames.
"""
import ctypes as C
import pickle
import sys
from pathlib import Path

import numpy as np
import pyarrow as pa
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import plugin_harness as ph  # noqa: E402

SYMBOLS = ["pl_lr", "pl_lr_pred", "pl_lin_reg_report", "pl_wls_report", "pl_rolling_lr", "pl_recursive_lr", "pl_lr_by", "pl_lr_by_pred", "pl_lr_multi",
           "pl_lr_multi_pred"]
# every `#[polars_expr] fn` of /root/reference/src/num_ext/linear_regression.rs (:419,517,587,651,704,822,982,1121,1206) and of
# linear_regression_f32.rs (:289,386,456,515,568,687,846,986,1072) -- the names `polars_ds.exprs.expr_linear` registers
REFERENCE_EXPRS = ["pl_lr", "pl_lr_multi", "pl_lr_multi_pred", "pl_lr_w_rcond", "pl_lr_pred", "pl_lin_reg_report", "pl_wls_report",
                   "pl_recursive_lr", "pl_rolling_lr"]


@pytest.fixture(scope="module")
def so():
    sys.path.insert(0, str(ROOT))
    from polars_ds_extension_amd import _build, _lib

    if not _lib.LIB_PATH.exists():
        _build.build()
    return _lib.load()


def test_plugin_symbols_and_version(so):
    for s in SYMBOLS:
        for suffix in ("", "_f32"):
            assert hasattr(so, f"_polars_plugin_{s}{suffix}") and hasattr(so, f"_polars_plugin_field_{s}{suffix}")
    so._polars_plugin_get_version.restype = C.c_uint32
    assert so._polars_plugin_get_version() == 1  # (major 0 << 16) | minor 1
    so._polars_plugin_get_last_error_message.restype = C.c_char_p
    assert isinstance(so._polars_plugin_get_last_error_message(), bytes)


def test_every_reference_expression_symbol_is_exported(so):
    """All 18 `#[polars_expr]` names of the reference's two lstsq files (9 + their _f32 twins), entry point and field function."""
    names = [n + sfx for n in REFERENCE_EXPRS for sfx in ("", "_f32")]
    assert len(names) == 18
    missing = [n for n in names if not (hasattr(so, f"_polars_plugin_{n}") and hasattr(so, f"_polars_plugin_field_{n}"))]
    assert not missing, f"reference expressions without a plugin symbol: {missing}"
    ref = Path("/root/reference/src/num_ext")
    if ref.exists():  # (this container only: the list above is what the reference declares)
        import re

        declared = []
        for f in ("linear_regression.rs", "linear_regression_f32.rs"):
            declared += re.findall(r"#\[polars_expr\([^\]]*\)\][^\n]*\n\s*fn\s+(\w+)", (ref / f).read_text())
        assert sorted(declared) == sorted(names)


def test_coalescing_queue_fails_cleanly_without_a_device(so):
    """No GPU here: every queued pl_lr call must come back with the library's error (nobody left waiting on a batch)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    kw = pickle.dumps({"bias": False, "null_policy": "raise", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5,
                       "max_iter": 200, "weighted": False, "positive": False, "singular_x_tol": 1e-12}, protocol=5)
    buf = (C.c_uint8 * len(kw)).from_buffer_copy(kw)
    sec, dev = C.c_double(), C.c_double()
    fails = so.pds_plugin_debug_concurrent_lr(8, 5, 50, 3, buf, len(kw), C.byref(sec), C.byref(dev))
    assert fails == 40
    so._polars_plugin_get_last_error_message.restype = C.c_char_p


def _debug_import(so, arr, kind=0):
    """Series import of csrc/plugin.cpp without a device: (values, valid, null_count, borrowed)."""
    se, keep = ph._export_series("c", arr)
    n = len(arr)
    dt = {0: np.float64, 1: np.float32, 2: np.int64}[kind]
    vals = np.full(n, -777, dtype=dt)
    valid = np.full(n, 9, dtype=np.uint8)
    nulls, borrowed = C.c_longlong(-1), C.c_int(-1)
    so.pds_plugin_debug_import.restype = C.c_longlong
    so.pds_plugin_debug_import.argtypes = [C.POINTER(ph.SeriesExport), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong),
                                           C.POINTER(C.c_int)]
    try:
        got = so.pds_plugin_debug_import(C.byref(se), kind, vals.ctypes.data, valid.ctypes.data, C.byref(nulls), C.byref(borrowed))
    finally:
        ph._release_inputs([keep])
    if got < 0:
        so._polars_plugin_get_last_error_message.restype = C.c_char_p
        raise ph.PluginFailure(so._polars_plugin_get_last_error_message().decode())
    assert got == n
    return vals, valid.astype(bool), nulls.value, bool(borrowed.value)


def test_series_import_borrows_single_chunks_and_reads_bitmaps(so):
    """Host logic of the ingestion (series_to_slice_inner's role, src/utils/mod.rs:101-206): one chunk of the working dtype
    is borrowed, everything else gathered + cast; validity at any bit offset, across chunks of any length."""
    rng = np.random.default_rng(11)
    n = 1000
    x = rng.normal(size=n + 70)
    mask = rng.random(n + 70) < 0.2
    full = pa.array(x, mask=mask)
    for off in list(range(0, 18)) + [63, 64, 65]:
        for ln in (0, 1, 7, 8, 9, 63, 64, 65, n):
            a = full.slice(off, ln)
            vals, valid, nulls, borrowed = _debug_import(so, a)
            want_valid = ~mask[off:off + ln]
            assert borrowed and np.array_equal(valid, want_valid) and nulls == int((~want_valid).sum())
            assert np.array_equal(vals[valid], x[off:off + ln][want_valid])
    # several chunks (odd lengths, with and without nulls, sliced) are gathered; bits land at their global positions
    for _ in range(20):
        cuts = np.sort(rng.integers(0, n, size=rng.integers(1, 6)))
        cuts = np.concatenate([[0], cuts, [n]])
        chunks = []
        for a, b in zip(cuts, cuts[1:]):
            ch = pa.array(x[a:b], mask=mask[a:b]) if rng.random() < 0.7 else pa.array(np.where(mask[a:b], 0.0, x[a:b]))
            chunks.append((ch, a, b, ch.null_count > 0 or not mask[a:b].any()))
        arr = pa.chunked_array([c[0] for c in chunks], type=pa.float64())
        vals, valid, nulls, borrowed = _debug_import(so, arr)
        want_valid = np.ones(n, dtype=bool)
        want_vals = x[:n].copy()
        for ch, a, b, _ in chunks:
            if ch.null_count:
                want_valid[a:b] = ~mask[a:b]
            else:
                want_vals[a:b] = ch.to_numpy(zero_copy_only=False)  # (masked values were written as 0.0 or kept as they are)
        assert borrowed == (len(chunks) == 1) and np.array_equal(valid, want_valid) and nulls == int((~want_valid).sum())
        assert np.array_equal(vals[valid], want_vals[want_valid])
    # casts: every integer width and the other float width; f32 target; int64 keys stay exact and are borrowed
    for dt in (np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32):
        v = rng.integers(0, 100, size=50).astype(dt)
        vals, valid, nulls, borrowed = _debug_import(so, pa.array(v).slice(3, 40))
        assert not borrowed and nulls == 0 and valid.all() and np.array_equal(vals, v[3:43].astype(np.float64))
    v32 = rng.normal(size=64).astype(np.float32)
    vals, _, _, borrowed = _debug_import(so, pa.array(v32).slice(5), kind=1)
    assert borrowed and np.array_equal(vals, v32[5:])
    vals, _, _, borrowed = _debug_import(so, pa.array(x[:64]), kind=1)
    assert not borrowed and np.array_equal(vals, x[:64].astype(np.float32))
    keys = np.array([(1 << 62) + 1, -(1 << 62) - 3, 9007199254740993, 0], dtype=np.int64)
    vals, _, _, borrowed = _debug_import(so, pa.array(keys), kind=2)
    assert borrowed and np.array_equal(vals, keys)
    vals, _, _, borrowed = _debug_import(so, pa.array(keys.astype(np.int32) % 1000), kind=2)
    assert not borrowed and np.array_equal(vals, (keys.astype(np.int32) % 1000).astype(np.int64))
    with pytest.raises(ph.PluginFailure, match="non-numeric"):
        _debug_import(so, pa.array(["a", "b"]))


def test_result_export_takes_over_device_written_rows(so):
    """Rolling / recursive results: the n x p' block the device wrote becomes the List values buffer as it is -- leading
    invalid rows are skipped by the buffer start, invalid rows in the middle closed up in place; validity bits packed 8 at a time."""
    rng = np.random.default_rng(4)
    for n, width, lead, holes in [(0, 3, 0, 0), (1, 2, 1, 0), (1, 2, 0, 0), (37, 3, 0, 0), (37, 3, 5, 0), (64, 1, 7, 9), (1001, 9, 255, 40),
                                  (513, 4, 513, 0), (130, 2, 0, 130)]:
        rows = rng.normal(size=(n, width))
        flags = np.ones(n, dtype=np.uint8)
        flags[:lead] = 0
        if holes and n > lead:
            flags[rng.choice(np.arange(lead, n), size=min(holes, n - lead), replace=False)] = 0
        flags[flags > 0] = rng.choice([1, 2, 255], size=int((flags > 0).sum()))  # any non-zero byte means valid
        ret = ph.SeriesExport()
        so.pds_plugin_debug_export_rows.restype = None
        so.pds_plugin_debug_export_rows.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.POINTER(ph.SeriesExport)]
        so.pds_plugin_debug_export_rows(rows.ctypes.data, n, width, flags.ctypes.data, C.byref(ret))
        assert ret.release
        field = pa.Field._import_from_c(C.addressof(ret.field.contents))
        out = pa.Array._import_from_c(C.addressof(ret.arrays[0].contents), field.type)
        C.CFUNCTYPE(None, C.POINTER(ph.SeriesExport))(ret.release)(C.byref(ret))
        out.validate(full=True)
        got = out.to_pylist()
        assert len(got) == n
        for i in range(n):
            if flags[i]:
                assert got[i]["coeffs"] == rows[i].tolist() and got[i]["pred"] == rows[i, 0]
            else:
                assert got[i]["coeffs"] is None and got[i]["pred"] is None


def test_kwargs_pickle_parser(so):
    # the exact dicts python/polars_ds/exprs/expr_linear.py builds (:237-248, :546-552)
    lr_kwargs = {"bias": True, "null_policy": "skip", "l1_reg": 0.0, "l2_reg": 0.25, "solver": "qr", "tol": 1e-5,
                 "max_iter": 200, "weighted": False, "positive": False, "singular_x_tol": 1e-12}
    sww = {"null_policy": "0.5", "n": 256, "bias": False, "lambda": 0.1, "min_size": 70000}
    for d in (lr_kwargs, sww, {"n": -3, "big": 2**40, "s": "x" * 300, "f": -1.5e300}):
        raw = pickle.dumps(d, protocol=5)
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        out = C.create_string_buffer(4096)
        assert so.pds_plugin_debug_parse_kwargs(buf, len(raw), out, 4096) == 0
        got = dict(item.split("=", 1) for item in out.value.decode().strip(";").split(";"))
        for k, v in d.items():
            if isinstance(v, bool):
                assert got[k] == ("true" if v else "false")
            elif isinstance(v, int):
                assert int(got[k]) == v
            elif isinstance(v, float):
                assert float(got[k]) == v
            else:
                assert got[k] == f"'{v}'"


def test_kwargs_parser_rejects_garbage_without_crashing(so):
    """Mutated / truncated / random byte strings: the parser answers 0 or an error code, never reads out of bounds
    (tests/mock_device/sanitize.sh runs the same sources under ASan + UBSan)."""
    import random

    rnd = random.Random(7)
    base = [pickle.dumps(d, protocol=pr) for pr in (2, 3, 4, 5) for d in (
        {"bias": True, "null_policy": "skip", "l1_reg": 0.0, "l2_reg": 0.25, "solver": "qr", "tol": 1e-5, "max_iter": 200},
        {"null_policy": "0.5", "n": 256, "bias": False, "lambda": 0.1, "min_size": 70000},
        {"n": -3, "huge": 2**63 - 1, "neg": -2**63, "s": "x" * 300, "f": -1.5e300, "none": None}, {})]
    out = C.create_string_buffer(1 << 14)
    outcomes = set()
    for _ in range(20_000):
        b = bytearray(rnd.choice(base))
        r = rnd.random()
        if r < 0.4:
            for _ in range(rnd.randint(1, 4)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif r < 0.6:
            b = b[: rnd.randrange(len(b) + 1)]
        elif r < 0.8:
            pos = rnd.randrange(len(b) + 1)
            b[pos:pos] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 8)))
        else:
            b = bytearray(rnd.randrange(256) for _ in range(rnd.randint(0, 64)))
        buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) or b"\0")
        outcomes.add(so.pds_plugin_debug_parse_kwargs(buf, len(b), out, len(out)))
    assert outcomes <= {0, -1, -2} and {0, -1} <= outcomes


def test_output_fields(so):
    f = ph.output_field(so, "pl_lr")
    assert f.name == "coeffs" and f.type == pa.large_list(pa.field("item", pa.float64()))
    assert ph.output_field(so, "pl_lr_f32").type == pa.large_list(pa.field("item", pa.float32()))
    f = ph.output_field(so, "pl_lr_pred")
    assert [c.name for c in f.type] == ["pred", "resid"]
    f = ph.output_field(so, "pl_lin_reg_report")
    assert f.name == "lin_reg_report" and [c.name for c in f.type] == ["features", "beta", "std_err", "t", "p>|t|", "0.025", "0.975", "r2", "adj_r2"]
    f = ph.output_field(so, "pl_rolling_lr")
    assert [c.name for c in f.type] == ["coeffs", "pred"]
    # pl_lr_by: the key field carries the key COLUMN's name (what the exported array is called), as Int64
    ins = [pa.field("gid", pa.int32()), pa.field("y", pa.float64()), pa.field("x1", pa.float64())]
    f = ph.output_field(so, "pl_lr_by", ins)
    assert [c.name for c in f.type] == ["gid", "coeffs"] and f.type[0].type == pa.int64()
    assert [c.name for c in ph.output_field(so, "pl_lr_by_f32", ins).type] == ["gid", "coeffs"]
    assert [c.name for c in ph.output_field(so, "pl_lr_by").type] == ["key", "coeffs"]


# ---------------------------------------------------------------------------------------------------- GPU
LR = {"bias": False, "null_policy": "skip", "l1_reg": 0.0, "l2_reg": 0.0, "solver": "qr", "tol": 1e-5, "max_iter": 200,
      "weighted": False, "positive": False, "singular_x_tol": 1e-12}


def _cols(X, y, names=None):
    names = names or [f"x{j + 1}" for j in range(X.shape[1])]
    return [("y", pa.array(y))] + [(nm, pa.array(np.ascontiguousarray(X[:, j]))) for j, nm in enumerate(names)]


@pytest.mark.gpu
def test_pl_lr_series_in_series_out(so, orc):
    rng = np.random.default_rng(208)
    X = rng.random((100_000, 4))
    y = X @ [0.5, 0.25, -0.15, 0.2] + 1e-4 * rng.random(100_000)  # BASELINE configs[0]
    field, out = ph.call_plugin(so, "pl_lr", _cols(X, y), LR)
    assert field.name == "coeffs" and len(out) == 1
    b = np.array(out[0].as_py())
    ref = orc.pl_lr(X, y)
    assert np.linalg.norm(b - ref) / np.linalg.norm(ref) < 1e-10
    # multi-chunk + int column (cast like series_to_slice_inner) + bias + ridge
    xi = rng.integers(0, 10, size=100_000)
    ins = [("y", pa.chunked_array([pa.array(y[:30_000]), pa.array(y[30_000:])])), ("xi", pa.array(xi)), ("x2", pa.array(X[:, 1]))]
    _, out = ph.call_plugin(so, "pl_lr", ins, dict(LR, bias=True, l2_reg=0.1))
    ref = orc.pl_lr(np.c_[xi.astype(float), X[:, 1]], y, add_bias=True, l2_reg=0.1)
    assert np.linalg.norm(np.array(out[0].as_py()) - ref) / np.linalg.norm(ref) < 1e-10
    # collinear -> the gate returns a 1-row null list
    _, out = ph.call_plugin(so, "pl_lr", [("y", pa.array(y)), ("a", pa.array(X[:, 0])), ("b", pa.array(2 * X[:, 0]))], LR)
    assert len(out) == 1 and out[0].as_py() is None
    # f32 twin
    _, out = ph.call_plugin(so, "pl_lr_f32", _cols(X.astype(np.float32), y.astype(np.float32)), dict(LR, singular_x_tol=1e-6))
    assert out.type == pa.large_list(pa.field("item", pa.float32()))
    assert np.allclose(out[0].as_py(), [0.5, 0.25, -0.15, 0.2], atol=1e-3)
    # errors keep the reference's strings
    with pytest.raises(ph.PluginFailure, match="#Data < #features"):
        ph.call_plugin(so, "pl_lr", _cols(X[:2], y[:2]), LR)
    with pytest.raises(ph.PluginFailure, match="Invalid NullPolicy"):
        ph.call_plugin(so, "pl_lr", _cols(X, y), dict(LR, null_policy="bogus"))


@pytest.mark.gpu
def test_pl_lr_pred_literal_skip_null_frame(so):
    # tests/test_linear_exprs.py:411-432
    ins = [("y", pa.array([8.5, 9.5, 10.5, 11.5, 12.5])), ("x", pa.array([None, 2.0, 3.0, 4.0, 5.0]))]
    field, out = ph.call_plugin(so, "pl_lr_pred", ins, dict(LR, bias=True))
    rows = out.to_pylist()
    assert rows[0] == {"pred": None, "resid": None}
    np.testing.assert_allclose([r["pred"] for r in rows[1:]], [9.5, 10.5, 11.5, 12.5], atol=1e-10)
    np.testing.assert_allclose([r["resid"] for r in rows[1:]], 0.0, atol=1e-10)
    with pytest.raises(ph.PluginFailure, match="Nulls found in data"):
        ph.call_plugin(so, "pl_lr_pred", ins, dict(LR, bias=True, null_policy="raise"))


@pytest.mark.gpu
def test_pl_lin_reg_report_struct(so, orc):
    rng = np.random.default_rng(2)
    n = 20_000
    X = rng.normal(size=(n, 3))
    y = X @ [0.5, 0.0, -1.0] + 2.0 + rng.normal(size=n)
    ins = [("y_var", pa.array([float(np.var(y, ddof=1))])), ("y", pa.array(y))] + [(nm, pa.array(X[:, j])) for j, nm in enumerate(["a", "b", "c"])]
    field, out = ph.call_plugin(so, "pl_lin_reg_report", ins, dict(LR, bias=True, std_err="hc1", null_policy="raise"))
    t = pa.Table.from_struct_array(out) if hasattr(pa.Table, "from_struct_array") else pa.Table.from_batches([pa.RecordBatch.from_struct_array(out)])
    assert t.column_names == ["features", "beta", "hc1_se", "t", "p>|t|", "0.025", "0.975", "r2", "adj_r2"]
    assert t["features"].to_pylist() == ["a", "b", "c", "__bias__"]
    ro = orc.lin_reg_report(np.c_[X, np.ones(n)], y, std_err="hc1")
    np.testing.assert_allclose(t["beta"].to_numpy(), ro["beta"], rtol=1e-10)
    np.testing.assert_allclose(t["hc1_se"].to_numpy(), ro["std_err"], rtol=1e-10)
    np.testing.assert_allclose(t["p>|t|"].to_numpy(), ro["p"], rtol=1e-8, atol=1e-300)
    assert len(set(t["r2"].to_pylist())) == 1 and abs(t["r2"][0].as_py() - ro["r2"]) < 1e-12
    # nulls + null_policy through the plugin ABI: rows with a null anywhere are skipped (series_to_mat_for_lr)
    m = rng.random(n) < 0.05
    ins2 = [ins[0], ins[1], ("a", pa.array(np.where(m, -7.0, X[:, 0]), mask=m)), ins[3], ins[4]]
    _, out2 = ph.call_plugin(so, "pl_lin_reg_report", ins2, dict(LR, bias=True, std_err="se", null_policy="skip"))
    t2 = pa.Table.from_batches([pa.RecordBatch.from_struct_array(out2)])
    ro2 = orc.lin_reg_report(np.c_[X[~m], np.ones((~m).sum())], y[~m], y_var=float(np.var(y, ddof=1)))
    np.testing.assert_allclose(t2["beta"].to_numpy(), ro2["beta"], rtol=1e-10)
    np.testing.assert_allclose(t2["std_err"].to_numpy(), ro2["std_err"], rtol=1e-10)
    with pytest.raises(ph.PluginFailure, match="Nulls found in data"):
        ph.call_plugin(so, "pl_lin_reg_report", ins2, dict(LR, bias=True, std_err="se", null_policy="raise"))


@pytest.mark.gpu
def test_pl_rolling_and_recursive_struct(so, orc, golden):
    rows = golden["rolling_w5_head"]
    ins = [("y", pa.array([r["y"] for r in rows])), ("x1", pa.array([r["x1"] for r in rows])), ("x2", pa.array([r["x2"] for r in rows]))]
    kw = {"null_policy": "zero", "n": 5, "bias": False, "lambda": 0.0, "min_size": 2}  # the notebook's call
    _, out = ph.call_plugin(so, "pl_rolling_lr", ins, kw)
    res = out.to_pylist()
    assert all(r == {"coeffs": None, "pred": None} for r in res[:4])
    np.testing.assert_allclose(res[4]["coeffs"], rows[4]["coeffs"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res[4]["pred"], rows[4]["pred"], rtol=2e-5, atol=2e-6)
    rng = np.random.default_rng(5)
    X = rng.random((3000, 2))
    y = X @ [1.0, -1.0] + 0.01 * rng.random(3000)
    _, out = ph.call_plugin(so, "pl_recursive_lr", _cols(X, y), {"null_policy": "raise", "n": 4, "bias": True, "lambda": 0.0, "min_size": 0})
    res = out.to_pylist()
    ref = orc.recursive_lr(np.c_[X, np.ones(3000)], y, 4)
    assert res[2]["coeffs"] is None and res[3]["coeffs"] is not None
    np.testing.assert_allclose(res[-1]["coeffs"], ref[-1], rtol=1e-8)
    # rolling with nulls under "skip": rows holding a null leave the window, short windows are null (:858-908)
    xn = X[:, 0].copy()
    mask = rng.random(3000) < 0.1
    ins = [("y", pa.array(y)), ("x1", pa.array(xn, mask=mask)), ("x2", pa.array(X[:, 1]))]
    _, out = ph.call_plugin(so, "pl_rolling_lr", ins, {"null_policy": "skip", "n": 8, "bias": False, "lambda": 0.0, "min_size": 6})
    Xn = np.c_[np.where(mask, np.nan, xn), X[:, 1]]
    ref, valid = orc.rolling_skipping_lr(Xn, y, 8, 6)
    got_valid = np.array([r["coeffs"] is not None for r in out.to_pylist()])
    assert np.array_equal(got_valid[7:], valid) and not got_valid[:7].any()


@pytest.mark.gpu
def test_pl_recursive_lr_null_policies(so, orc):
    # pl_recursive_lr :1131-1181: skip / fill drop rows, the expanding fit runs on the rest and is spread back
    rng = np.random.default_rng(11)
    n, n0 = 2000, 6
    X = rng.random((n, 2))
    y = X @ [1.0, -1.0] + 0.3 + 0.01 * rng.random(n)
    mx = rng.random(n) < 0.08
    my = rng.random(n) < 0.03
    mx[:3] = True  # nulls inside the first start_with rows push the first result back
    ins = [("y", pa.array(y, mask=my)), ("x1", pa.array(X[:, 0], mask=mx)), ("x2", pa.array(X[:, 1]))]
    for policy, keep, Xf in (("skip", ~(mx | my), X), ("0.5", ~my, np.c_[np.where(mx, 0.5, X[:, 0]), X[:, 1]])):
        _, out = ph.call_plugin(so, "pl_recursive_lr", ins, {"null_policy": policy, "n": n0, "bias": True, "lambda": 0.0, "min_size": 0})
        res = out.to_pylist()
        idx = np.flatnonzero(keep)
        ref = orc.recursive_lr(np.c_[Xf[idx], np.ones(len(idx))], y[idx], n0)  # row j of ref: fit on the first n0 + j kept rows
        first = idx[n0 - 1]
        for i in range(n):
            if (not keep[i]) or i < first:
                assert res[i] == {"coeffs": None, "pred": None}
        for j in (0, 1, 50, len(idx) - n0):
            i = idx[n0 - 1 + j]
            tol = 1e-6 if j > 10 else 1e-3  # the earliest fits are nearly singular
            np.testing.assert_allclose(res[i]["coeffs"], ref[j], rtol=tol, atol=tol)
            np.testing.assert_allclose(res[i]["pred"], np.r_[Xf[i], 1.0] @ np.array(res[i]["coeffs"]), rtol=1e-10, atol=1e-12)
    with pytest.raises(ph.PluginFailure, match="Nulls found in data"):
        ph.call_plugin(so, "pl_recursive_lr", ins, {"null_policy": "raise", "n": n0, "bias": True, "lambda": 0.0, "min_size": 0})


@pytest.mark.gpu
def test_reference_quirks_switch(so, orc):
    """PDS_REFERENCE_QUIRKS=1 reproduces the two accidents of the reference's result assembly that the default build
    does not: pl_lr_pred + ignore + nulls -> ONE null row (linear_regression.rs:194-197, :790-806);
    pl_recursive_lr + skip / fill + nulls -> pred from compacted row j (:1158-1166)."""
    import os

    rng = np.random.default_rng(5)
    n, n0 = 400, 5
    X = rng.random((n, 2))
    y = X @ [2.0, -1.0] + 0.5 + 0.01 * rng.random(n)
    mx = rng.random(n) < 0.1
    ins = [("y", pa.array(y)), ("x1", pa.array(X[:, 0], mask=mx)), ("x2", pa.array(X[:, 1]))]
    kw_rec = {"null_policy": "skip", "n": n0, "bias": True, "lambda": 0.0, "min_size": 0}
    _, plain = ph.call_plugin(so, "pl_recursive_lr", ins, kw_rec)
    _, pred_plain = ph.call_plugin(so, "pl_lr_pred", ins, dict(LR, bias=True, null_policy="ignore"))
    assert len(pred_plain) == n
    os.environ["PDS_REFERENCE_QUIRKS"] = "1"
    so.pds_plugin_reload_settings()  # (the plugin layer reads its environment once: plugin_settings.hpp)
    try:
        _, pred_q = ph.call_plugin(so, "pl_lr_pred", ins, dict(LR, bias=True, null_policy="ignore"))
        _, rec_q = ph.call_plugin(so, "pl_recursive_lr", ins, kw_rec)
        # no nulls: nothing changes
        clean = [("y", pa.array(y)), ("x1", pa.array(X[:, 0])), ("x2", pa.array(X[:, 1]))]
        _, pred_clean = ph.call_plugin(so, "pl_lr_pred", clean, dict(LR, bias=True, null_policy="ignore"))
    finally:
        del os.environ["PDS_REFERENCE_QUIRKS"]
        so.pds_plugin_reload_settings()
    assert pred_q.to_pylist() == [{"pred": None, "resid": None}]
    assert len(pred_clean) == n and pred_clean.null_count == 0
    plain, rec_q = plain.to_pylist(), rec_q.to_pylist()
    idx = np.flatnonzero(~mx)
    for j, i in enumerate(idx):
        if j < n0 - 1:
            assert rec_q[i] == {"coeffs": None, "pred": None}
            continue
        assert rec_q[i]["coeffs"] == plain[i]["coeffs"]  # the coefficients are the same rows either way
        r = idx[j - (n0 - 1)]  # compacted row j - (n - 1) of the reference's x
        np.testing.assert_allclose(rec_q[i]["pred"], np.r_[X[r], 1.0] @ np.array(plain[i]["coeffs"]), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(plain[i]["pred"], np.r_[X[i], 1.0] @ np.array(plain[i]["coeffs"]), rtol=1e-10, atol=1e-12)
    for i in np.flatnonzero(mx):
        assert rec_q[i] == {"coeffs": None, "pred": None}


@pytest.mark.gpu
def test_borrowed_chunks_slices_and_bitmaps(so, orc):
    """A single Float64 chunk is read in place (no marshalling copy): slices (offset != 0), validity bitmaps at bit offsets
    that are not byte aligned, several chunks of odd lengths -- and a fill policy never writes into the caller's buffers."""
    rng = np.random.default_rng(77)
    n = 5000
    X = rng.normal(size=(n + 11, 3))
    y = X @ [1.0, -2.0, 0.5] + 0.1 * rng.normal(size=n + 11)
    mask = rng.random(n + 11) < 0.03
    for off in (3, 8, 11):  # bit offsets 3 and 11 take the bit loop, 8 the byte-copy path
        ins = [("y", pa.array(y).slice(off, n)), ("x1", pa.array(X[:, 0], mask=mask).slice(off, n)),
               ("x2", pa.array(X[:, 1]).slice(off, n)), ("x3", pa.array(X[:, 2]).slice(off, n))]
        _, out = ph.call_plugin(so, "pl_lr", ins, dict(LR, null_policy="skip"))
        m = ~mask[off:off + n]
        ref = orc.pl_lr(X[off:off + n][m], y[off:off + n][m])
        np.testing.assert_allclose(out[0].as_py(), ref, rtol=1e-10)
        _, out = ph.call_plugin(so, "pl_lr_pred", ins, dict(LR, null_policy="skip"))
        pred = out.field("pred").to_pylist()
        assert [v is None for v in pred] == list(~m)
        np.testing.assert_allclose(np.array([v for v in pred if v is not None]), X[off:off + n][m] @ ref, rtol=1e-9, atol=1e-11)
    # chunks of 13 / 1001 / rest rows with nulls in every chunk
    cuts = [0, 13, 1014, n]
    xa = pa.array(X[:n, 0], mask=mask[:n])
    ins = [("y", pa.chunked_array([pa.array(y[a:b]) for a, b in zip(cuts, cuts[1:])])),
           ("x1", pa.chunked_array([xa.slice(a, b - a) for a, b in zip(cuts, cuts[1:])])),
           ("x2", pa.array(X[:n, 1])), ("x3", pa.array(X[:n, 2]))]
    _, out = ph.call_plugin(so, "pl_lr", ins, dict(LR, null_policy="skip"))
    m = ~mask[:n]
    np.testing.assert_allclose(out[0].as_py(), orc.pl_lr(X[:n][m], y[:n][m]), rtol=1e-10)
    # fill policies rewrite a private copy: the engine's buffers are untouched
    x1 = pa.array(X[:n, 0], mask=mask[:n])
    raw_before = np.frombuffer(x1.buffers()[1], dtype=np.float64).copy()
    ins = [("y", pa.array(y[:n])), ("x1", x1), ("x2", pa.array(X[:n, 1]))]
    _, out = ph.call_plugin(so, "pl_rolling_lr", ins, {"null_policy": "zero", "n": 50, "bias": False, "lambda": 0.0, "min_size": 2})
    assert len(out) == n
    _, out = ph.call_plugin(so, "pl_recursive_lr", ins, {"null_policy": "1.5", "n": 50, "bias": False, "lambda": 0.0, "min_size": 0})
    assert len(out) == n
    assert np.array_equal(np.frombuffer(x1.buffers()[1], dtype=np.float64), raw_before)


@pytest.mark.gpu
def test_pl_lr_by_int64_keys_are_exact(so, orc):
    # keys beyond 2^53 (not representable as f64) and negative keys keep their identity, in any row order
    rng = np.random.default_rng(5)
    base = np.array([(1 << 60) + 1, (1 << 60) + 2, (1 << 60) + 3, -(1 << 61), -5, 0, 9007199254740993], dtype=np.int64)
    per = 40
    key = np.repeat(base, per)
    X = rng.normal(size=(len(key), 2))
    y = X @ [2.0, -1.0] + 0.05 * rng.normal(size=len(key)) + np.repeat(np.arange(len(base)), per)
    perm = rng.permutation(len(key))
    for order in (np.arange(len(key)), perm):
        _, out = ph.call_plugin(so, "pl_lr_by", [("key", pa.array(key[order]))] + _cols(X[order], y[order]), dict(LR, bias=True))
        res = out.to_pylist()
        want = sorted(base.tolist()) if order is perm else None
        got = [r["key"] for r in res]
        assert sorted(got) == sorted(base.tolist()) and (want is None or got == want)
        for r in res:
            m = key == r["key"]
            np.testing.assert_allclose(r["coeffs"], orc.pl_lr(X[m], y[m], add_bias=True), rtol=1e-9, atol=1e-11)
    # Int32 keys are widened
    k32 = np.repeat(np.array([5, -7, 100], dtype=np.int32), per)
    _, out = ph.call_plugin(so, "pl_lr_by", [("key", pa.array(k32))] + _cols(X[:3 * per], y[:3 * per]), LR)
    assert [r["key"] for r in out.to_pylist()] == [-7, 5, 100] or [r["key"] for r in out.to_pylist()] == [5, -7, 100]


@pytest.mark.gpu
def test_pl_lr_by_matches_per_group_calls(so, orc):
    # tests/test_linear_exprs.py:918-953: the batched path == one pl_lr call per group
    rng = np.random.default_rng(0)
    G, per = 200, 25
    key = np.repeat(np.arange(G) * 3 + 7, per)
    X = rng.normal(size=(G * per, 2))
    y = X @ [1.5, -0.5] + rng.normal(size=G * per)
    field, out = ph.call_plugin(so, "pl_lr_by", [("key", pa.array(key))] + _cols(X, y), LR)
    res = out.to_pylist()
    assert len(res) == G and res[0]["key"] == 7 and res[-1]["key"] == (G - 1) * 3 + 7
    for g in (0, 17, G - 1):
        s = slice(g * per, (g + 1) * per)
        _, single = ph.call_plugin(so, "pl_lr", _cols(X[s], y[s]), LR)
        np.testing.assert_allclose(res[g]["coeffs"], single[0].as_py(), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
def test_pl_lr_by_keys_in_any_row_order(so, orc):
    # group_by(key) does not need a sorted frame: shuffled rows give the same per-key coefficients, keys ascending
    rng = np.random.default_rng(3)
    G, per = 150, 30
    key = np.repeat(rng.permutation(G) * 2 - 50, per)
    X = rng.normal(size=(G * per, 3))
    y = X @ [0.5, -1.5, 2.0] + 0.2 * rng.normal(size=G * per) + key * 0.01
    perm = rng.permutation(G * per)
    _, out = ph.call_plugin(so, "pl_lr_by", [("key", pa.array(key[perm]))] + _cols(X[perm], y[perm]), dict(LR, bias=True))
    res = out.to_pylist()
    assert [r["key"] for r in res] == sorted(set(key.tolist()))
    for r in res[::17]:
        m = key == r["key"]
        np.testing.assert_allclose(r["coeffs"], orc.pl_lr(X[m], y[m], add_bias=True), rtol=1e-9, atol=1e-11)
    # weights: [key, w, y, x...] with kwargs.weighted -- per key the weighted fit of pl_lr
    w = rng.random(G * per) + 0.2
    ins_w = [("key", pa.array(key[perm])), ("w", pa.array(w[perm]))] + _cols(X[perm], y[perm])
    _, out = ph.call_plugin(so, "pl_lr_by", ins_w, dict(LR, bias=True, weighted=True))
    res_w = out.to_pylist()
    assert [r["key"] for r in res_w] == sorted(set(key.tolist()))
    for r in res_w[::19]:
        m = key == r["key"]
        np.testing.assert_allclose(r["coeffs"], orc.pl_lr(X[m], y[m], add_bias=True, weights=w[m]), rtol=1e-9, atol=1e-11)
    # with nulls (null_policy = skip) the rows are ordered on the host and take the bitmap-aware entry point
    mask = rng.random(G * per) < 0.05
    ins = [("key", pa.array(key[perm])), ("y", pa.array(y[perm])), ("x1", pa.array(X[perm][:, 0], mask=mask[perm]))] + [
        (f"x{j + 1}", pa.array(X[perm][:, j])) for j in (1, 2)]
    _, out = ph.call_plugin(so, "pl_lr_by", ins, dict(LR, bias=True, null_policy="skip"))
    res = out.to_pylist()
    assert [r["key"] for r in res] == sorted(set(key.tolist()))
    for r in res[::23]:
        m = (key == r["key"]) & ~mask
        np.testing.assert_allclose(r["coeffs"], orc.pl_lr(X[m], y[m], add_bias=True), rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
def test_pl_lr_by_pred_is_the_reference_group_by_test(so, orc):
    """/root/reference/tests/test_linear_exprs.py:435-474 (test_lin_reg_in_group_by): group_by("A").agg(lin_reg(..., return_pred=True))
    equals the per-group `pl_lr_pred` results -- here ONE `pl_lr_by_pred` call over the frame, rows in frame order."""
    A = np.array([1] * 4 + [2] * 4, dtype=np.int64)
    Y = np.ones(8)
    X1 = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.float64)
    X2 = np.array([2, 3, 4, 1, 6, 7, 8, 5], dtype=np.float64)
    kw = dict(LR, bias=False)
    ins = [("A", pa.array(A)), ("Y", pa.array(Y)), ("X1", pa.array(X1)), ("X2", pa.array(X2))]
    field, out = ph.call_plugin(so, "pl_lr_by_pred", ins, kw)
    assert len(out) == 8 and [f.name for f in out.type] == ["pred", "resid"]
    got = out.to_pylist()
    for a in (1, 2):
        m = A == a
        _, single = ph.call_plugin(so, "pl_lr_pred", [("Y", pa.array(Y[m])), ("X1", pa.array(X1[m])), ("X2", pa.array(X2[m]))], kw)
        want = single.to_pylist()
        for r, w in zip([g for g, k in zip(got, m) if k], want):
            assert abs(r["pred"] - w["pred"]) < 1e-10 and abs(r["resid"] - w["resid"]) < 1e-10
    # the oracle, per group
    X = np.c_[X1, X2]
    for a in (1, 2):
        m = A == a
        b = orc.pl_lr(X[m], Y[m])
        np.testing.assert_allclose([g["pred"] for g, k in zip(got, m) if k], X[m] @ b, rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
def test_pl_lr_by_pred_large_host_frame_takes_the_sliced_route(so, orc):
    """A host frame of >= 2^22 rows with ordered keys: `pl_lr_by_pred` cuts it into slices over the process's contexts
    (pds_lr_by_key_pred_multi_*: PDS_BY_KEY_CONTEXTS per device); predictions equal the single-context call's
    (PDS_BY_KEY_MULTI_MIN_ROWS=0) row by row and the oracle's on sampled groups."""
    rng = np.random.default_rng(77)
    G = 45_000
    sizes = rng.integers(60, 130, size=G)
    key = np.repeat(np.arange(G, dtype=np.int64) * 3 + 5, sizes)
    n = len(key)
    assert n >= 1 << 22
    X = rng.normal(size=(n, 2))
    y = X @ [0.7, -1.1] + 1e-4 * key + 0.3 * rng.normal(size=n)
    ins = [("key", pa.array(key))] + _cols(X, y)
    _, out = ph.call_plugin(so, "pl_lr_by_pred", ins, dict(LR, bias=True))
    pred = out.field("pred").to_numpy(zero_copy_only=False)
    import os

    os.environ["PDS_BY_KEY_MULTI_MIN_ROWS"] = "0"
    so.pds_plugin_reload_settings()
    try:
        _, out1 = ph.call_plugin(so, "pl_lr_by_pred", ins, dict(LR, bias=True))
    finally:
        del os.environ["PDS_BY_KEY_MULTI_MIN_ROWS"]
        so.pds_plugin_reload_settings()
    pred1 = out1.field("pred").to_numpy(zero_copy_only=False)
    assert out.field("pred").null_count == 0 and len(pred) == n
    assert np.max(np.abs(pred - pred1) / (np.abs(pred1) + 1.0)) < 1e-10
    off = np.concatenate([[0], np.cumsum(sizes)])
    for g in rng.choice(G, size=25, replace=False):
        sl = slice(off[g], off[g + 1])
        b = orc.pl_lr(X[sl], y[sl], add_bias=True)
        np.testing.assert_allclose(pred[sl], X[sl] @ b[:2] + b[2], rtol=1e-9, atol=1e-10)


@pytest.mark.gpu
def test_pl_lr_by_pred_shuffled_rows_weights_bias_and_null_groups(so, orc):
    """Rows in any order: every row gets ITS group's prediction where the row is (the `.over(key)` broadcast of
    examples/basics.ipynb cells 16 / 18); a collinear group is null for all of its rows (linear_regression.rs:745-750); weights."""
    rng = np.random.default_rng(31)
    G, per = 120, 35
    key = np.repeat(rng.permutation(G) * 5 - 100, per).astype(np.int64)
    X = rng.normal(size=(G * per, 3))
    bad = key == key[7 * per]
    X[bad, 1] = 2.0 * X[bad, 0]  # one collinear group -> the rank gate fires
    y = X @ [0.5, -1.5, 2.0] + 0.2 * rng.normal(size=G * per) + key * 0.01
    perm = rng.permutation(G * per)
    kp, Xp, yp = key[perm], X[perm], y[perm]
    _, out = ph.call_plugin(so, "pl_lr_by_pred", [("key", pa.array(kp))] + _cols(Xp, yp), dict(LR, bias=True))
    assert len(out) == G * per
    pred = np.array([np.nan if v is None else v for v in out.field("pred").to_pylist()])
    resid = np.array([np.nan if v is None else v for v in out.field("resid").to_pylist()])
    assert out.field("pred").null_count == per and out.field("resid").null_count == per
    assert np.isnan(pred[kp == key[7 * per]]).all()
    for k in np.unique(key)[::11]:
        m = kp == k
        if k == key[7 * per]:
            continue
        b = orc.pl_lr(Xp[m], yp[m], add_bias=True)
        want = Xp[m] @ b[:3] + b[3]
        np.testing.assert_allclose(pred[m], want, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(resid[m], yp[m] - want, rtol=0, atol=1e-9)
    # weights ([key, w, y, x...], kwargs.weighted): per key the weighted fit, predictions from the unweighted rows
    w = rng.random(G * per) + 0.2
    ins_w = [("key", pa.array(kp)), ("w", pa.array(w))] + _cols(Xp, yp)
    _, out = ph.call_plugin(so, "pl_lr_by_pred", ins_w, dict(LR, bias=True, weighted=True))
    pred_w = np.array([np.nan if v is None else v for v in out.field("pred").to_pylist()])
    for k in np.unique(key)[::13]:
        m = kp == k
        if k == key[7 * per]:
            continue
        b = orc.pl_lr(Xp[m], yp[m], add_bias=True, weights=w[m])
        np.testing.assert_allclose(pred_w[m], Xp[m] @ b[:3] + b[3], rtol=1e-9, atol=1e-11)
    # ordered keys take the no-movement route and must agree with the shuffled call row by row
    order = np.argsort(kp, kind="stable")
    _, out2 = ph.call_plugin(so, "pl_lr_by_pred", [("key", pa.array(kp[order]))] + _cols(Xp[order], yp[order]), dict(LR, bias=True))
    pred2 = np.array([np.nan if v is None else v for v in out2.field("pred").to_pylist()])
    ok = ~np.isnan(pred[order])
    np.testing.assert_allclose(pred2[ok], pred[order][ok], rtol=1e-10, atol=1e-12)
    # rows with nulls: what every group's pl_lr_pred does (linear_regression.rs:151-267, 790-812) -- "skip" fits on the rows
    # without a null and the dropped rows come back null; a fill value fills the features and drops the rows whose target is null
    mx = rng.random(G * per) < 0.1
    my = rng.random(G * per) < 0.05
    mx[kp == key[7 * per]] = False  # (a filled feature would break the collinear group's collinearity)
    ins_n = [("key", pa.array(kp)), ("y", pa.array(yp, mask=my)), ("x1", pa.array(Xp[:, 0], mask=mx))] + [
        (f"x{j + 1}", pa.array(Xp[:, j])) for j in (1, 2)]
    for policy, dropped, Xf in (("skip", mx | my, Xp), ("0.5", my, np.c_[np.where(mx, 0.5, Xp[:, 0]), Xp[:, 1:]])):
        _, outn = ph.call_plugin(so, "pl_lr_by_pred", ins_n, dict(LR, bias=True, null_policy=policy))
        assert len(outn) == G * per
        pn = np.array([np.nan if v is None else v for v in outn.field("pred").to_pylist()])
        nulls_n = np.array([v is None for v in outn.field("pred").to_pylist()])
        bad_group = kp == key[7 * per]
        assert np.array_equal(nulls_n, dropped | bad_group), policy
        for k in np.unique(key)[::17]:
            m = (kp == k) & ~dropped
            if k == key[7 * per]:
                continue
            b = orc.pl_lr(Xf[m], yp[m], add_bias=True)
            np.testing.assert_allclose(pn[m], Xf[m] @ b[:3] + b[3], rtol=1e-9, atol=1e-10)
    # "ignore": a group that holds a null fits on NaN -- all of ITS rows come back NaN (valid), the other groups as usual
    mi = np.zeros(G * per, bool)
    hit = np.unique(key)[[3, 40, 77]]
    for k in hit:
        mi[np.flatnonzero(kp == k)[:2]] = True
    ins_i = [("key", pa.array(kp)), ("y", pa.array(yp)), ("x1", pa.array(Xp[:, 0], mask=mi))] + [
        (f"x{j + 1}", pa.array(Xp[:, j])) for j in (1, 2)]
    _, outi = ph.call_plugin(so, "pl_lr_by_pred", ins_i, dict(LR, bias=True, null_policy="ignore"))
    pl_ = outi.field("pred").to_pylist()
    rl_ = outi.field("resid").to_pylist()
    in_hit = np.isin(kp, hit)
    bad_group = kp == key[7 * per]
    assert all((v is None) == bool(b) for v, b in zip(pl_, bad_group))          # only the collinear group is null
    assert all(v is not None and np.isnan(v) for v, h in zip(pl_, in_hit) if h)  # NaN rows, not null rows
    assert all(v is not None and np.isnan(v) for v, h in zip(rl_, in_hit) if h)
    pi = np.array([np.nan if v is None else v for v in pl_])
    rest = ~in_hit & ~bad_group
    np.testing.assert_allclose(pi[rest], pred[rest], rtol=1e-10, atol=1e-12)     # untouched groups: the null-free call's answers


@pytest.mark.gpu
def test_pl_lr_by_null_keys_form_one_group(so, orc):
    """Polars' group_by makes the null keys one group: pl_lr_by reports it with a null key, pl_lr_by_pred predicts its rows."""
    rng = np.random.default_rng(77)
    n = 600
    key = rng.integers(0, 5, size=n).astype(np.int64)
    isnull = rng.random(n) < 0.2
    X = rng.normal(size=(n, 2))
    y = X @ [1.0, -2.0] + 0.1 * rng.normal(size=n) + np.where(isnull, 3.0, key * 0.5)
    karr = pa.array(key, mask=isnull)
    _, out = ph.call_plugin(so, "pl_lr_by", [("key", karr)] + _cols(X, y), dict(LR, bias=True))
    res = out.to_pylist()
    assert [r["key"] for r in res] == [0, 1, 2, 3, 4, None]
    np.testing.assert_allclose(res[-1]["coeffs"], orc.pl_lr(X[isnull], y[isnull], add_bias=True), rtol=1e-9, atol=1e-11)
    for r in res[:-1]:
        m = (key == r["key"]) & ~isnull
        np.testing.assert_allclose(r["coeffs"], orc.pl_lr(X[m], y[m], add_bias=True), rtol=1e-9, atol=1e-11)
    _, outp = ph.call_plugin(so, "pl_lr_by_pred", [("key", karr)] + _cols(X, y), dict(LR, bias=True))
    pred = np.array(outp.field("pred").to_pylist(), dtype=np.float64)
    b = orc.pl_lr(X[isnull], y[isnull], add_bias=True)
    np.testing.assert_allclose(pred[isnull], X[isnull] @ b[:2] + b[2], rtol=1e-9, atol=1e-11)
    # all keys null: one group, null key
    _, out = ph.call_plugin(so, "pl_lr_by", [("key", pa.array(key, mask=np.ones(n, dtype=bool)))] + _cols(X, y), dict(LR, bias=True))
    res = out.to_pylist()
    assert len(res) == 1 and res[0]["key"] is None
    np.testing.assert_allclose(res[0]["coeffs"], orc.pl_lr(X, y, add_bias=True), rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
def test_concurrent_pl_lr_calls_are_coalesced(so, orc):
    """group_by().agg(pds.lin_reg(...)) unchanged: Polars' rayon threads call pl_lr once per group.  Calls that arrive while
    a batch is on the device leave together as one grouped launch; every caller must get its own group's coefficients."""
    from concurrent.futures import ThreadPoolExecutor

    rng = np.random.default_rng(12)
    frames = []
    for g in range(96):
        n = int(rng.integers(20, 200))
        X = rng.normal(size=(n, 3))
        y = X @ rng.normal(size=3) + 0.5 + 0.1 * rng.normal(size=n)
        frames.append((X, y))
    frames[7] = (np.c_[frames[7][0][:, 0], 2.0 * frames[7][0][:, 0], frames[7][0][:, 2]], frames[7][1])  # collinear -> null list

    def one(i):
        X, y = frames[i]
        _, out = ph.call_plugin(so, "pl_lr", _cols(X, y), dict(LR, bias=True))
        return out[0].as_py()

    with ThreadPoolExecutor(max_workers=16) as ex:
        res = list(ex.map(one, range(len(frames)))) + list(ex.map(one, range(len(frames))))
    for i, r in enumerate(res):
        X, y = frames[i % len(frames)]
        bo = orc.pl_lr(X, y, add_bias=True)
        if bo is None:
            assert r is None
        else:
            np.testing.assert_allclose(r, bo, rtol=1e-9, atol=1e-11)
    # the same through native threads (no GIL between the calls): results stable across repeats, requests accounted for
    kw = pickle.dumps(dict(LR, bias=False), protocol=5)
    buf = (C.c_uint8 * len(kw)).from_buffer_copy(kw)
    sec, dev = C.c_double(), C.c_double()
    so.pds_plugin_debug_coalesce_stats(None, None, None, 1)
    fails = so.pds_plugin_debug_concurrent_lr(24, 40, 100, 8, buf, len(kw), C.byref(sec), C.byref(dev))
    b, r, m = C.c_longlong(), C.c_longlong(), C.c_longlong()
    so.pds_plugin_debug_coalesce_stats(C.byref(b), C.byref(r), C.byref(m), 0)
    assert fails == 0 and dev.value < 1e-12
    assert r.value == 24 * 40 and 1 <= b.value <= r.value and m.value <= 24
    # errors stay per call: too few rows raises for that caller only
    with pytest.raises(ph.PluginFailure, match="#Data < #features"):
        ph.call_plugin(so, "pl_lr", _cols(frames[0][0][:2], frames[0][1][:2]), dict(LR, bias=True))


@pytest.mark.gpu
def test_concurrent_pred_weighted_and_null_bearing_calls_are_coalesced(so, orc):
    """The per-group calls of `group_by().agg(lin_reg(..., return_pred=True))`, of weighted fits and of fits on frames with
    nulls (skip / fill) batch too: mixed kinds arrive together, every caller gets the answer of ITS frame and ITS kind, and
    the queue accounts for every request (none falls back to the per-call path)."""
    from concurrent.futures import ThreadPoolExecutor

    rng = np.random.default_rng(21)
    frames = []
    for g in range(60):
        n = int(rng.integers(25, 160))
        X = rng.normal(size=(n, 3))
        y = X @ rng.normal(size=3) + 0.3 + 0.1 * rng.normal(size=n)
        frames.append((X, y, rng.random(n) + 0.2, rng.random(n) < 0.1))
    X7 = frames[7][0]
    frames[7] = (np.c_[X7[:, 0], 2.0 * X7[:, 0], X7[:, 2]],) + frames[7][1:]  # collinear -> the gate fires (unweighted kinds)

    def one(job):
        i, kind = job
        X, y, w, mask = frames[i]
        if kind == "pred":
            _, out = ph.call_plugin(so, "pl_lr_pred", _cols(X, y), dict(LR, bias=True))
            return out
        if kind == "wpred":
            _, out = ph.call_plugin(so, "pl_lr_pred", [("w", pa.array(w))] + _cols(X, y), dict(LR, bias=True, weighted=True))
            return out
        if kind == "weighted":
            _, out = ph.call_plugin(so, "pl_lr", [("w", pa.array(w))] + _cols(X, y), dict(LR, bias=True, weighted=True))
            return out[0].as_py()
        ins = [("y", pa.array(y)), ("x1", pa.array(X[:, 0], mask=mask)), ("x2", pa.array(X[:, 1])), ("x3", pa.array(X[:, 2]))]
        _, out = ph.call_plugin(so, "pl_lr", ins, dict(LR, bias=True, null_policy="skip" if kind == "skip" else "zero"))
        return out[0].as_py()

    jobs = [(i, k) for k in ("pred", "wpred", "weighted", "skip", "zero") for i in range(len(frames))]
    order = rng.permutation(len(jobs))
    so.pds_plugin_debug_coalesce_stats(None, None, None, 1)
    with ThreadPoolExecutor(max_workers=16) as ex:
        res = list(ex.map(one, [jobs[j] for j in order]))
    b, r, m = C.c_longlong(), C.c_longlong(), C.c_longlong()
    so.pds_plugin_debug_coalesce_stats(C.byref(b), C.byref(r), C.byref(m), 0)
    assert r.value == len(jobs) and 1 <= b.value <= r.value
    for j, got in zip(order, res):
        i, kind = jobs[j]
        X, y, w, mask = frames[i]
        Xb = np.c_[X, np.ones(len(y))]
        if kind in ("pred", "wpred"):
            bo = orc.pl_lr(X, y, add_bias=True, weights=w if kind == "wpred" else None)
            if bo is None:
                assert got.field("pred").null_count == len(y)
                continue
            np.testing.assert_allclose(got.field("pred").to_numpy(zero_copy_only=False), Xb @ bo, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(got.field("resid").to_numpy(zero_copy_only=False), y - Xb @ bo, rtol=0, atol=1e-9)
        elif kind == "weighted":
            if i == 7:  # (collinear frame, no gate on the weighted path: the coefficients are not unique, the fit is)
                np.testing.assert_allclose(Xb @ np.asarray(got), Xb @ orc.pl_lr(X, y, add_bias=True, weights=w), rtol=1e-8, atol=1e-9)
            else:
                np.testing.assert_allclose(got, orc.pl_lr(X, y, add_bias=True, weights=w), rtol=1e-9, atol=1e-10)
        else:
            if kind == "skip":
                bo = orc.pl_lr(X[~mask], y[~mask], add_bias=True)
            else:
                Xz = X.copy()
                Xz[mask, 0] = 0.0
                bo = orc.pl_lr(Xz, y, add_bias=True)
            if bo is None:
                assert got is None
            else:
                np.testing.assert_allclose(got, bo, rtol=1e-9, atol=1e-10)
    # errors stay per call: a skip frame left with fewer rows than coefficients raises for that caller, as the single-frame path does
    Xs, ys = frames[0][0][:5], frames[0][1][:5]
    ins = [("y", pa.array(ys)), ("x1", pa.array(Xs[:, 0], mask=np.array([True, True, True, False, False]))), ("x2", pa.array(Xs[:, 1])),
           ("x3", pa.array(Xs[:, 2]))]
    with pytest.raises(ph.PluginFailure, match="#Data < #features"):
        ph.call_plugin(so, "pl_lr", ins, dict(LR, bias=True, null_policy="skip"))


@pytest.mark.gpu
def test_pl_lr_multi_and_rcond(so, orc):
    # tests/test_linear_exprs.py:1069-1113: struct fields named after the (aliased) targets; :477-512 rcond
    rng = np.random.default_rng(4)
    n = 8000
    X = rng.normal(size=(n, 3))
    Y = np.c_[X @ [1.0, 2.0, -1.0] + 0.5, X @ [0.0, -1.0, 3.0]] + 0.1 * rng.normal(size=(n, 2))
    ins = [("target_0", pa.array(Y[:, 0])), ("target_1", pa.array(Y[:, 1]))] + [(f"x{j}", pa.array(X[:, j])) for j in range(3)]
    kw = {"bias": True, "null_policy": "raise", "solver": "qr", "last_target_idx": 2, "l2_reg": 0.0, "singular_x_tol": 1e-12}
    field, out = ph.call_plugin(so, "pl_lr_multi", ins, kw)
    row = out.to_pylist()[0]
    assert list(row) == ["target_0", "target_1"]
    for i in range(2):
        np.testing.assert_allclose(row[f"target_{i}"], orc.pl_lr(X, Y[:, i], add_bias=True), rtol=1e-10, atol=1e-12)
    _, out = ph.call_plugin(so, "pl_lr_multi_pred", ins, kw)
    t = out.to_pylist()
    assert list(t[0]) == ["target_0_pred", "target_0_resid", "target_1_pred", "target_1_resid"] and len(t) == n
    b = orc.pl_lr(X, Y[:, 1], add_bias=True)
    np.testing.assert_allclose([r["target_1_pred"] for r in t[:50]], (np.c_[X, np.ones(n)] @ b)[:50], rtol=1e-10)
    _, out = ph.call_plugin(so, "pl_lr_w_rcond", [("y", pa.array(Y[:, 0]))] + ins[2:], dict(LR, tol=0.3))
    r = out.to_pylist()[0]
    ref, _, _, sv = np.linalg.lstsq(X, Y[:, 0], rcond=0.3)
    np.testing.assert_allclose(r["coeffs"], ref, atol=1e-10)
    np.testing.assert_allclose(r["singular_values"], sv, rtol=1e-10)
    # the f32 twin (linear_regression_f32.rs:515-566): List<f32> fields, rcond = max(tol as f32, f32::EPSILON * max(n, p'))
    f32_ins = [("y", pa.array(Y[:, 0].astype(np.float32)))] + [(f"x{j}", pa.array(X[:, j].astype(np.float32))) for j in range(3)]
    field, out = ph.call_plugin(so, "pl_lr_w_rcond_f32", f32_ins, dict(LR, tol=0.3))
    r32 = out.to_pylist()[0]
    assert out.type.field("coeffs").type == pa.large_list(pa.float32()) and list(r32) == ["coeffs", "singular_values"]
    bo, svo = orc.solve_lr_rcond(X.astype(np.float32), Y[:, 0].astype(np.float32), rcond=np.float32(0.3))
    np.testing.assert_allclose(r32["coeffs"], ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r32["coeffs"], bo, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r32["singular_values"], sv, rtol=1e-5)
    # null policies reach pl_lr_w_rcond too (series_to_mat_for_lr is called at :660): skip == the fit on the complete rows
    x0 = X[:, 0].copy()
    mask = np.zeros(n, dtype=bool)
    mask[::17] = True
    nul_ins = [("y", pa.array(Y[:, 0])), ("x0", pa.array(x0, mask=mask))] + ins[3:]
    with pytest.raises(ph.PluginFailure, match="Nulls found in data"):
        ph.call_plugin(so, "pl_lr_w_rcond", nul_ins, dict(LR, tol=0.3, null_policy="raise"))
    _, out = ph.call_plugin(so, "pl_lr_w_rcond", nul_ins, dict(LR, tol=0.3, null_policy="skip"))
    ref_s, _, _, sv_s = np.linalg.lstsq(X[~mask], Y[~mask, 0], rcond=0.3)
    np.testing.assert_allclose(out.to_pylist()[0]["coeffs"], ref_s, atol=1e-10)
    _, out = ph.call_plugin(so, "pl_lr_w_rcond", nul_ins, dict(LR, tol=0.3, null_policy="zero"))
    Xz = X.copy()
    Xz[mask, 0] = 0.0
    np.testing.assert_allclose(out.to_pylist()[0]["coeffs"], np.linalg.lstsq(Xz, Y[:, 0], rcond=0.3)[0], atol=1e-10)
