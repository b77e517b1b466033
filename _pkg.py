"""Loads the product package from its (non-identifier) directory `gst-plugin-rs_amd/`
under the importable module name `gst_plugin_rs_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "gst-plugin-rs_amd")
MODNAME = "gst_plugin_rs_amd"


def load():
    if MODNAME in sys.modules:
        return sys.modules[MODNAME]
    spec = importlib.util.spec_from_file_location(
        MODNAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[MODNAME] = mod
    spec.loader.exec_module(mod)
    return mod


vfx = load()
