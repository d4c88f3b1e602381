"""Import alias: ``import building_gan_amd`` loads the package kept in ``building-gan_amd/``
(a directory name with a hyphen cannot be written in an import statement)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "building-gan_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
