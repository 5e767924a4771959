"""Minimal gin-config compatible reader for the NEWT hot path.

The reference takes every constructor argument of its modules from gin-config
(`gin/models/newt.gin`).  gin-config is not installed in
this image and only a small part of its surface is used on the forward path, so
this module re-implements that part from the documented behaviour:

  * ``parse_config_file`` / ``parse_config``: macros (``name = value``),
    macro references (``%name``), bindings (``Class.param = value``), scoped
    bindings (``scope/Class.param = value``), configurable references
    (``@Class`` / ``@Class()``), ``include 'file.gin'`` and ``import x`` lines.
  * ``@configurable`` (bare and called), ``external_configurable``,
    ``config_scope`` (context manager), ``constant``, ``clear_config``,
    ``query_parameter``.

Semantics reproduced: a configurable's missing keyword arguments are filled from
bindings; bindings made under an active scope win over global bindings
(reference use: ``with gin.config_scope("noise_synth")``,
/root/reference/neural_waveshaping_synthesis/models/neural_waveshaping.py:58).
Explicitly passed arguments always win.
"""
from __future__ import annotations

import ast
import contextlib
import functools
import inspect
import os
import re
import threading

__all__ = [
    "configurable",
    "external_configurable",
    "config_scope",
    "parse_config_file",
    "parse_config",
    "constant",
    "clear_config",
    "query_parameter",
    "bind_parameter",
    "REQUIRED",
]

REQUIRED = object()

_REGISTRY: dict[str, object] = {}      # selector name -> wrapped callable
_BINDINGS: dict[tuple[str, str], dict[str, object]] = {}  # (scope, name) -> {param: value}
_MACROS: dict[str, object] = {}
_CONSTANTS: dict[str, object] = {}
_TLS = threading.local()


class _MacroRef:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def resolve(self):
        if self.name in _MACROS:
            return _resolve(_MACROS[self.name])
        if self.name in _CONSTANTS:
            return _CONSTANTS[self.name]
        raise ValueError(f"gin: undefined macro %{self.name}")


class _ConfigurableRef:
    __slots__ = ("name", "call")

    def __init__(self, name, call):
        self.name = name
        self.call = call

    def resolve(self):
        scope = ""
        name = self.name
        if "/" in name:
            scope, name = name.rsplit("/", 1)
        fn = _lookup(name)
        if self.call:
            with config_scope(scope) if scope else contextlib.nullcontext():
                return fn()
        return fn


def _resolve(v):
    if isinstance(v, (_MacroRef, _ConfigurableRef)):
        return v.resolve()
    if isinstance(v, list):
        return [_resolve(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_resolve(x) for x in v)
    if isinstance(v, dict):
        return {k: _resolve(x) for k, x in v.items()}
    return v


def _short(name: str) -> str:
    return name.rsplit(".", 1)[-1]


def _lookup(name: str):
    if name in _REGISTRY:
        return _REGISTRY[name]
    s = _short(name)
    if s in _REGISTRY:
        return _REGISTRY[s]
    raise ValueError(f"gin: no configurable named '{name}'")


def _scope_stack():
    if not hasattr(_TLS, "scopes"):
        _TLS.scopes = []
    return _TLS.scopes


@contextlib.contextmanager
def config_scope(name):
    st = _scope_stack()
    if name:
        st.append(str(name))
    try:
        yield
    finally:
        if name:
            st.pop()


def _bindings_for(name: str) -> dict:
    """Global bindings overlaid by bindings of each active scope (innermost last)."""
    out = dict(_BINDINGS.get(("", name), {}))
    for sc in _scope_stack():
        out.update(_BINDINGS.get((sc, name), {}))
    return out


def _wrap(fn, name, module=None):
    is_class = inspect.isclass(fn)
    target = fn.__init__ if is_class else fn
    try:
        sig = inspect.signature(target)
        params = list(sig.parameters.values())
        if is_class:
            params = params[1:]
        names = [p.name for p in params if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
        has_var_kw = any(p.kind == p.VAR_KEYWORD for p in params)
    except (TypeError, ValueError):
        names, has_var_kw = [], True

    def fill(args, kwargs):
        bound = _bindings_for(name)
        if not bound:
            return kwargs
        given = set(kwargs)
        given.update(names[: len(args)])
        kw = dict(kwargs)
        for k, v in bound.items():
            if k in given:
                continue
            if not has_var_kw and k not in names:
                raise TypeError(f"gin: '{name}' has no parameter '{k}'")
            kw[k] = _resolve(v)
        return kw

    if is_class:
        orig_init = fn.__init__

        @functools.wraps(orig_init)
        def __init__(self, *args, **kwargs):
            # Also reached through ``super().__init__()`` of a subclass (the
            # reference's FastNEWT relies on that to get the NEWT.* bindings).
            orig_init(self, *args, **fill(args, kwargs))

        fn.__init__ = __init__
        wrapped = fn
    else:
        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            return fn(*args, **fill(args, kwargs))

    _REGISTRY[name] = wrapped
    if module:
        _REGISTRY[f"{module}.{name}"] = wrapped
    return wrapped


def configurable(name_or_fn=None, module=None, **_ignored):
    """``@configurable`` or ``@configurable("name", module=...)``."""
    if callable(name_or_fn):
        return _wrap(name_or_fn, name_or_fn.__name__, module)

    def deco(fn):
        return _wrap(fn, name_or_fn or fn.__name__, module)

    return deco


def external_configurable(fn, name=None, module=None, **_ignored):
    """Register a third-party callable.  Unlike ``configurable`` the original
    object is left untouched; a configured subclass / wrapper is registered."""
    nm = name or fn.__name__
    if inspect.isclass(fn):
        sub = type(fn.__name__, (fn,), {"__module__": fn.__module__, "__doc__": fn.__doc__})
        return _wrap(sub, nm, module)
    return _wrap(functools.wraps(fn)(lambda *a, **k: fn(*a, **k)), nm, module)


def constant(name, value):
    _CONSTANTS[name] = value


def bind_parameter(binding_key: str, value):
    scope, sel, param = _split_key(binding_key)
    _BINDINGS.setdefault((scope, sel), {})[param] = value


def query_parameter(binding_key: str):
    if binding_key.startswith("%"):
        return _MacroRef(binding_key[1:]).resolve()
    scope, sel, param = _split_key(binding_key)
    d = _BINDINGS.get((scope, sel), {})
    if param not in d:
        raise ValueError(f"gin: no binding for '{binding_key}'")
    return _resolve(d[param])


def clear_config():
    _BINDINGS.clear()
    _MACROS.clear()


def _split_key(key: str):
    scope = ""
    if "/" in key:
        scope, key = key.rsplit("/", 1)
    sel, param = key.rsplit(".", 1)
    return scope, _short(sel), param


_TOKEN_REF = re.compile(r"(?<![\w.])([%@])([A-Za-z_][\w./]*)(\(\))?")


def _parse_value(text: str):
    """Python literal with %macro / @configurable references."""
    refs = []

    def sub(m):
        kind, name, call = m.group(1), m.group(2), m.group(3)
        refs.append(_MacroRef(name) if kind == "%" else _ConfigurableRef(name, bool(call)))
        return f"__gin_ref__({len(refs) - 1})"

    # leave string literals alone: split on quotes conservatively
    parts = re.split(r"('(?:[^'\\]|\\.)*'|\"(?:[^\"\\]|\\.)*\")", text)
    for i in range(0, len(parts), 2):
        parts[i] = _TOKEN_REF.sub(sub, parts[i])
    node = ast.parse("".join(parts).strip(), mode="eval").body

    def ev(n):
        if isinstance(n, ast.Constant):
            return n.value
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "__gin_ref__":
            return refs[n.args[0].value]
        if isinstance(n, ast.List):
            return [ev(e) for e in n.elts]
        if isinstance(n, ast.Tuple):
            return tuple(ev(e) for e in n.elts)
        if isinstance(n, ast.Dict):
            return {ev(k): ev(v) for k, v in zip(n.keys, n.values)}
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.USub, ast.UAdd)):
            v = ev(n.operand)
            return -v if isinstance(n.op, ast.USub) else v
        if isinstance(n, ast.Name) and n.id in ("True", "False", "None"):
            return {"True": True, "False": False, "None": None}[n.id]
        raise ValueError(f"gin: unsupported value syntax: {text!r}")

    return ev(node)


def _logical_lines(text: str):
    buf, depth = "", 0
    for raw in text.splitlines():
        line = re.sub(r"(?<!['\"])#.*$", "", raw).rstrip()
        if not line.strip() and depth == 0:
            continue
        buf = (buf + " " + line.strip()) if buf else line.strip()
        depth = sum(buf.count(c) for c in "([{") - sum(buf.count(c) for c in ")]}")
        if depth <= 0 and not buf.endswith("\\"):
            yield buf
            buf, depth = "", 0
        elif buf.endswith("\\"):
            buf = buf[:-1]
    if buf:
        yield buf


def parse_config(text, skip_unknown=False, _base_dir="."):
    if isinstance(text, (list, tuple)):
        text = "\n".join(text)
    for line in _logical_lines(text):
        if line.startswith("import ") or line.startswith("from "):
            continue
        m = re.match(r"include\s+['\"](.+)['\"]\s*$", line)
        if m:
            path = m.group(1)
            cands = [path, os.path.join(_base_dir, path)]
            for c in cands:
                if os.path.exists(c):
                    parse_config_file(c, skip_unknown)
                    break
            else:
                raise IOError(f"gin: include file not found: {path}")
            continue
        if "=" not in line:
            raise ValueError(f"gin: cannot parse line: {line!r}")
        key, val = line.split("=", 1)
        key, val = key.strip(), val.strip()
        value = _parse_value(val)
        if "." not in key.rsplit("/", 1)[-1]:
            _MACROS[key] = value
            continue
        scope, sel, param = _split_key(key)
        if sel not in _REGISTRY and not skip_unknown:
            # gin resolves lazily for modules imported later; keep the binding
            pass
        _BINDINGS.setdefault((scope, sel), {})[param] = value


def parse_config_file(path, skip_unknown=False):
    with open(path) as f:
        text = f.read()
    parse_config(text, skip_unknown, _base_dir=os.path.dirname(os.path.abspath(path)) or ".")
