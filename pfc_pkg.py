"""Importer for the in-tree package directory ``pressurefieldcontact.jl_amd/``.

The directory name contains a dot (it mirrors the reference's repository name), so a plain
``import`` statement cannot name it.  ``load()`` registers it under the importable alias ``pfc_amd``
and returns the module; every entry point (tests, bench.py, __graft_entry__.py) goes through here.
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "pressurefieldcontact.jl_amd")
ALIAS = "pfc_amd"


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
