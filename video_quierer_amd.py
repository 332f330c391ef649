"""Import shim: the package directory is ``video-quierer_amd/`` (the name the
build contract fixes; a hyphen is not importable), so this module turns itself
into a package whose search path is that directory.  ``import video_quierer_amd``
and ``from video_quierer_amd.core.feature_extractor import FeatureExtractor``
both resolve into ``video-quierer_amd/``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "video-quierer_amd")
__path__ = [_PKG_DIR]
if __spec__ is not None:  # make importlib treat this module as a package
    __spec__.submodule_search_locations = __path__

with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
