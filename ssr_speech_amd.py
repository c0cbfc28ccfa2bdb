"""Import alias: the package directory is ``ssr-speech_amd/`` (a hyphen is not importable),
so this one-file module turns itself into a package whose ``__path__`` is that directory.

    import ssr_speech_amd                       # runs ssr-speech_amd/__init__.py
    from ssr_speech_amd.models.ssr import SSR_Speech
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ssr-speech_amd")
__path__ = [_PKG_DIR]
__file__ = _os.path.join(_PKG_DIR, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
