"""Import harness for the reference (scikit-recommender) -- used ONLY by tests/golden/make_golden.py,
ONLY in the build container where /root/reference is mounted.  Nothing here runs on the GPU box
and nothing of the reference is copied: the reference's Python is imported from where it lies, and
its four Cython extensions are compiled where they lie (cython + g++ called directly, outputs only
under oracle/_ref/cyext/, git-ignored).

Four import-time adjustments are needed on this image:
  1. ``collections.Iterable`` alias        (reference: utils/py/decorator.py:10, io/data_iterator.py:11
                                            use the pre-3.10 name)
  2. a no-colour ``colorama`` placeholder  (reference: skrec/__init__.py:12-13, utils/py/evaluator.py:10;
                                            cosmetic only: ANSI colour strings)
  3. a ``hyperopt`` placeholder            (reference: utils/hyperopt.py:11; only used when
                                            run_config.hyperopt is True, which the goldens never set)
  4. ``scipy.sparse.dok_matrix._update``   (reference: recommender/LayerGCN.py:182 calls this private
                                            bulk-insert, removed in scipy >= 1.13; re-added here as a
                                            plain dict update of the same (row, col) -> 1 entries)
None of them touches the numerical path that the golden vectors pin.
"""
import collections
import collections.abc
import importlib.abc
import importlib.machinery
import os
import subprocess
import sys
import sysconfig
import types

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CYEXT = os.path.join(REPO, "oracle", "_ref", "cyext")
_PYX = ["pyx_init", "pyx_utils", "pyx_random", "pyx_eval_matrix"]
_PKG = "skrec.utils.py.cython"


def build_cyext():
    """cythonize + compile the reference's .pyx files in place (no setup.py, no copy)."""
    import numpy
    os.makedirs(CYEXT, exist_ok=True)
    src_dir = os.path.join(REF, "skrec", "utils", "py", "cython")
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    for m in _PYX:
        so = os.path.join(CYEXT, m + ext)
        if os.path.exists(so):
            continue
        cpp = os.path.join(CYEXT, m + ".cpp")
        subprocess.run(["cython", "-3", "--cplus", "-o", cpp, os.path.join(src_dir, m + ".pyx")],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++11", "-pthread", "-w",
                        "-I" + src_dir, "-I" + numpy.get_include(),
                        "-I" + sysconfig.get_paths()["include"], "-o", so, cpp], check=True)
        os.remove(cpp)


class _CyFinder(importlib.abc.MetaPathFinder):
    """Resolve skrec.utils.py.cython.pyx_* to the extension files built under oracle/_ref/cyext."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PKG + "."):
            return None
        leaf = fullname.rsplit(".", 1)[1]
        if leaf not in _PYX:
            return None
        so = os.path.join(CYEXT, leaf + sysconfig.get_config_var("EXT_SUFFIX"))
        loader = importlib.machinery.ExtensionFileLoader(fullname, so)
        return importlib.machinery.ModuleSpec(fullname, loader, origin=so)


def install():
    assert os.path.isdir(REF), "the reference is only available in the build container"
    build_cyext()
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable
    if "colorama" not in sys.modules:
        try:
            import colorama  # noqa: F401
        except ImportError:
            m = types.ModuleType("colorama")

            class _Blank:
                def __getattr__(self, name):
                    return ""
            m.Fore, m.Back, m.Style = _Blank(), _Blank(), _Blank()
            m.init = lambda *a, **k: None
            sys.modules["colorama"] = m
    if "hyperopt" not in sys.modules:
        try:
            import hyperopt  # noqa: F401
        except ImportError:
            m = types.ModuleType("hyperopt")
            for n in ("fmin", "tpe", "hp", "Trials", "space_eval"):
                setattr(m, n, None)
            sys.modules["hyperopt"] = m
    import scipy.sparse as _sp
    if not hasattr(_sp.dok_matrix, "_update"):
        _sp.dok_matrix._update = lambda self, data: self._dict.update(data)
    sys.meta_path.insert(0, _CyFinder())
    # the reference must win over this repo's own drop-in package of the same name
    sys.path = [p for p in sys.path if "scikit-recommender_amd" not in p]
    sys.path.insert(0, REF)
    import skrec  # noqa: F401
    assert skrec.__file__.startswith(REF)
    return skrec
