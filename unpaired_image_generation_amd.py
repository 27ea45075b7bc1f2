"""Import shim: the product package lives in the directory `unpaired-image-generation_amd/` (a name Python cannot
import directly).  `import unpaired_image_generation_amd` executes this file, which loads that directory as the
package of the same (underscored) name, so `unpaired_image_generation_amd.networks` etc. resolve inside it."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "unpaired-image-generation_amd")
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
