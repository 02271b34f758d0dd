"""``csdl`` namespace used by the operator surface: the real ``csdl_alpha`` when it can be
imported, otherwise the in-tree stand-in (femo_alpha_amd/csdl_shim.py)."""
try:                                    # pragma: no cover - csdl_alpha is absent in this image
    import csdl_alpha as _csdl
    from csdl_alpha import *            # noqa: F401,F403
    experimental = _csdl.experimental
    check_parameter = _csdl.check_parameter
    HAVE_CSDL_ALPHA = True
except Exception:                       # ModuleNotFoundError here
    from .csdl_shim import *            # noqa: F401,F403
    from .csdl_shim import experimental, check_parameter
    HAVE_CSDL_ALPHA = False
