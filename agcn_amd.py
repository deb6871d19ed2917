"""Import alias: the package directory is named ``2s-agcn_amd`` (not a valid
Python identifier), so this module loads it under the name ``agcn_amd``.

``import agcn_amd`` (with the repo root on ``sys.path``) replaces this stub in
``sys.modules`` by the real package, after which ``agcn_amd.lib``,
``agcn_amd.ops`` ... resolve as ordinary submodules.
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "2s-agcn_amd")
_spec = importlib.util.spec_from_file_location(
    "agcn_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["agcn_amd"] = _mod
_spec.loader.exec_module(_mod)
