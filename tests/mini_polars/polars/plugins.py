"""`polars.plugins.register_plugin_function` of mini_polars (see the package docstring): keyword names as in Polars >= 1.0."""
from __future__ import annotations

from . import _Plugin, col


def register_plugin_function(*, plugin_path, function_name, args, kwargs=None, is_elementwise=False, changes_length=False,
                             returns_scalar=False, cast_to_supertype=False, input_wildcard_expansion=False,
                             pass_name_to_apply=False, use_abs_path=False):
    if not isinstance(args, (list, tuple)):
        args = [args]
    args = [col(a) if isinstance(a, str) else a for a in args]
    return _Plugin(plugin_path, function_name, args, kwargs or {}, returns_scalar, changes_length)
