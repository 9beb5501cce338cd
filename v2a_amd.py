"""Import shim: registers the package directory `video-to-audio-and-piano-rp_amd/` (not a valid
Python identifier) as the importable package `v2a_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-to-audio-and-piano-rp_amd")
_spec = importlib.util.spec_from_file_location("v2a_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["v2a_amd"] = _mod
_spec.loader.exec_module(_mod)
