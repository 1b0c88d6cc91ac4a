import enum
import importlib
import inspect


def optional_import(module, name="", **_kw):
    try:
        mod = importlib.import_module(module)
        return (getattr(mod, name) if name else mod), True
    except Exception:  # pragma: no cover
        return None, False


def ensure_tuple_rep(x, n):
    if isinstance(x, (list, tuple)):
        if len(x) != n:
            raise ValueError(f"Sequence must have length {n}, got {len(x)}.")
        return tuple(x)
    return (x,) * n


def look_up_option(key, supported, default="no_default"):
    if isinstance(supported, type) and issubclass(supported, enum.Enum):
        if isinstance(key, supported):
            return key
        for item in supported:
            if item.value == key:
                return item
        raise ValueError(f"Unsupported option {key!r}")
    if isinstance(supported, dict):
        if key in supported:
            return supported[key]
        raise ValueError(f"Unsupported option {key!r}, available: {list(supported)}")
    if key in supported:
        return key
    raise ValueError(f"Unsupported option {key!r}, available: {list(supported)}")


def has_option(obj, keywords):
    if not callable(obj):
        return False
    sig = inspect.signature(obj)
    if isinstance(keywords, str):
        keywords = (keywords,)
    return all(k in sig.parameters for k in keywords)


class SkipMode(enum.Enum):
    CAT = "cat"
    ADD = "add"
    MUL = "mul"


def alias(*_names):
    return lambda obj: obj


def export(_modname):
    return lambda obj: obj


def deprecated_arg(*_a, **_k):
    return lambda obj: obj
