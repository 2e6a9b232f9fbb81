"""Import helper: the package directory is named `uoparallel-seismic-project_amd`
(hyphen, not importable by name), so load it from its path and register it as
`uoparallel_seismic_project_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "uoparallel-seismic-project_amd")
PKG_NAME = "uoparallel_seismic_project_amd"


def load():
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(
        PKG_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
