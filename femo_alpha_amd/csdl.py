"""``csdl`` namespace used by the operator surface: the real ``csdl_alpha`` when it can be imported.  Otherwise the
repository's stand-in for the slice of CSDL the operators touch (base classes, variables, an inline recorder with a
reverse-mode driver) -- test and example scaffolding, kept OUTSIDE the product package: ``examples/csdl_standin.py``
(with its SLSQP driver ``examples/optimize.py``).  A deployment has ``csdl_alpha``; without either, importing the
operator surface fails and says so."""
try:                                    # pragma: no cover - csdl_alpha is absent in this image
    import csdl_alpha as _csdl
    from csdl_alpha import *            # noqa: F401,F403
    experimental = _csdl.experimental
    check_parameter = _csdl.check_parameter
    HAVE_CSDL_ALPHA = True
except Exception:                       # ModuleNotFoundError here
    import importlib.util as _ilu
    import os as _os
    import sys as _sys
    _path = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "examples", "csdl_standin.py")
    if not _os.path.exists(_path):
        raise ImportError("the FEAModel / StateOperation / OutputOperation surface needs csdl_alpha (not installed); the "
                          "repository's stand-in examples/csdl_standin.py is not beside the package either")
    _mod = _sys.modules.get("femo_csdl_standin")
    if _mod is None:
        _spec = _ilu.spec_from_file_location("femo_csdl_standin", _path)
        _mod = _ilu.module_from_spec(_spec)
        _sys.modules["femo_csdl_standin"] = _mod
        _spec.loader.exec_module(_mod)
    globals().update({_k: getattr(_mod, _k) for _k in _mod.__all__})
    HAVE_CSDL_ALPHA = False
