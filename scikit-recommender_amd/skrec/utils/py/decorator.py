"""``timer`` / ``typeassert`` decorators (reference: utils/py/decorator.py), collections.abc-safe."""
import time
from collections.abc import Iterable
from functools import wraps
from inspect import signature

__all__ = ["timer", "typeassert"]


def typeassert(*type_args, **type_kwargs):
    def decorate(func):
        sig = signature(func)
        expected = sig.bind_partial(*type_args, **type_kwargs).arguments

        @wraps(func)
        def wrapper(*args, **kwargs):
            for name, value in sig.bind(*args, **kwargs).arguments.items():
                if name not in expected:
                    continue
                types = expected[name]
                types = tuple(types) if isinstance(types, Iterable) else (types,)
                if value is None and None in types:
                    continue
                real = tuple(t for t in types if t is not None)
                if not isinstance(value, real):
                    raise TypeError(f"Argument {name} must be {real}")
            return func(*args, **kwargs)
        return wrapper
    return decorate


def timer(func):
    @wraps(func)
    def wrapper(*args, **kwargs):
        t0 = time.time()
        result = func(*args, **kwargs)
        print("%s function cost: %fs" % (func.__name__, time.time() - t0))
        return result
    return wrapper
