"""Import helper: the package directory is literally `tortoise.cpp_amd/` (the dot makes it
un-importable by name), so it is registered in sys.modules as `tortoise_cpp_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "tortoise.cpp_amd")


def load():
    if "tortoise_cpp_amd" in sys.modules:
        return sys.modules["tortoise_cpp_amd"]
    spec = importlib.util.spec_from_file_location(
        "tortoise_cpp_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["tortoise_cpp_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
