"""
pyarrow + ctypes standing in for the Polars engine when calling the `_polars_plugin_*` symbols of
libpds_lstsq_hip.so (test infrastructure; polars itself is not installable in this image).

call_plugin("pl_lr", [("y", arr), ("x1", arr), ...], {"bias": False, ...}) does what Polars does: exports every
input Series over the Arrow C Data Interface into a SeriesExport, pickles the kwargs dict with protocol 5, calls
the symbol, and imports the returned Series.
"""
from __future__ import annotations

import ctypes as C
import pickle

import pyarrow as pa


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
    ("children", C.POINTER(C.POINTER(ArrowSchema))), ("dictionary", C.POINTER(ArrowSchema)), ("release", C.c_void_p),
    ("private_data", C.c_void_p),
]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
    ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(ArrowArray))), ("dictionary", C.POINTER(ArrowArray)),
    ("release", C.c_void_p), ("private_data", C.c_void_p),
]


class SeriesExport(C.Structure):
    _fields_ = [("field", C.POINTER(ArrowSchema)), ("arrays", C.POINTER(C.POINTER(ArrowArray))), ("len", C.c_size_t),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


class PluginFailure(RuntimeError):
    pass


def _export_series(name: str, arr) -> tuple[SeriesExport, list]:
    chunks = arr.chunks if isinstance(arr, pa.ChunkedArray) else [arr]
    schema = ArrowSchema()
    pa.field(name, chunks[0].type)._export_to_c(C.addressof(schema))
    c_arrays = []
    for ch in chunks:
        a = ArrowArray()
        ch._export_to_c(C.addressof(a))
        c_arrays.append(a)
    ptrs = (C.POINTER(ArrowArray) * len(c_arrays))(*[C.pointer(a) for a in c_arrays])
    se = SeriesExport(C.pointer(schema), ptrs, len(c_arrays), None, None)
    return se, [schema, c_arrays, ptrs, chunks]


def _release_inputs(keep) -> None:
    rel_s = C.CFUNCTYPE(None, C.POINTER(ArrowSchema))
    rel_a = C.CFUNCTYPE(None, C.POINTER(ArrowArray))
    for schema, c_arrays, _ptrs, _chunks in keep:
        if schema.release:
            rel_s(schema.release)(C.byref(schema))
        for a in c_arrays:
            if a.release:
                rel_a(a.release)(C.byref(a))


def call_plugin(lib: C.CDLL, symbol: str, inputs, kwargs: dict | None):
    """Returns the result as a pyarrow Array; raises PluginFailure with the plugin's error message."""
    exports, keep = [], []
    for name, arr in inputs:
        se, k = _export_series(name, arr)
        exports.append(se)
        keep.append(k)
    ins = (SeriesExport * len(exports))(*exports)
    kw = pickle.dumps(kwargs, protocol=5) if kwargs else b""
    kwbuf = (C.c_uint8 * max(len(kw), 1)).from_buffer_copy(kw or b"\0")
    ret = SeriesExport()
    fn = getattr(lib, "_polars_plugin_" + symbol)
    fn.restype = None
    fn.argtypes = [C.POINTER(SeriesExport), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(SeriesExport), C.c_void_p]
    try:
        fn(ins, len(exports), kwbuf, len(kw), C.byref(ret), None)
    finally:
        _release_inputs(keep)
    if not ret.release:
        lib._polars_plugin_get_last_error_message.restype = C.c_char_p
        raise PluginFailure(lib._polars_plugin_get_last_error_message().decode())
    assert ret.len == 1
    field = pa.Field._import_from_c(C.addressof(ret.field.contents))
    out = pa.Array._import_from_c(C.addressof(ret.arrays[0].contents), field.type)
    C.CFUNCTYPE(None, C.POINTER(SeriesExport))(ret.release)(C.byref(ret))
    return field, out


def output_field(lib: C.CDLL, symbol: str, input_fields=None) -> pa.Field:
    """`input_fields`: the pa.Field list Polars passes as the expression's input schema (None: no inputs, as before)."""
    ret = ArrowSchema()
    fn = getattr(lib, "_polars_plugin_field_" + symbol)
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(ArrowSchema)]
    if not input_fields:
        fn(None, 0, C.byref(ret))
    else:
        arr = (ArrowSchema * len(input_fields))()
        for i, f in enumerate(input_fields):
            f._export_to_c(C.addressof(arr[i]))
        fn(C.cast(arr, C.c_void_p), len(input_fields), C.byref(ret))
        for i in range(len(input_fields)):  # (the callee borrows them)
            if arr[i].release:
                C.CFUNCTYPE(None, C.POINTER(ArrowSchema))(arr[i].release)(C.byref(arr[i]))
    return pa.Field._import_from_c(C.addressof(ret))
