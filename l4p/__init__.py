"""``l4p`` — import alias of the MI355X engine package ``l4p_amd``.

The reference's callers import the model and data surface under the package name ``l4p``
(/root/reference/demo/demo.py:12-17: ``from l4p.models.utils import prepare_model``,
``from l4p.data.video_dataset import VideoDataset``; configs/model.yaml names ``l4p.l4p.L4PLitModule``,
``l4p.models.l4p_videomae.L4P_VideoMAE``, ``l4p.models.task_heads.…``).  With this directory ahead of the reference on
``sys.path`` those lines resolve to the engine unchanged: ``l4p.<x>`` IS the module ``l4p_amd.<x>`` (the same module
object, no copy, no reference code).  Sub-modules the engine does not provide (visualisation, the training datasets:
l4p.utils.vis, l4p.utils.viser, l4p.data.davis, …) raise ImportError naming this fact.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import l4p_amd as _engine

_PREFIX, _REAL = __name__ + ".", _engine.__name__ + "."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _REAL + fullname[len(_PREFIX):]
        try:
            found = importlib.util.find_spec(real)
        except ModuleNotFoundError:
            found = None
        if found is None:
            raise ModuleNotFoundError(
                f"No module named {fullname!r}: the MI355X engine (package l4p_amd, aliased as l4p) does not provide it — it "
                "covers the inference hot path (l4p.l4p, l4p.models.*, l4p.data.video_dataset, l4p.utils.geometry_utils)",
                name=fullname)
        return importlib.util.spec_from_loader(fullname, self, is_package=found.submodule_search_locations is not None)

    def create_module(self, spec):
        mod = importlib.import_module(_REAL + spec.name[len(_PREFIX):])
        self._real_spec = mod.__spec__
        return mod

    def exec_module(self, module):
        module.__spec__ = self._real_spec  # keep the engine module's own identity (the import system stamped the alias spec)


sys.meta_path.insert(0, _AliasFinder())
__path__ = []  # a package: sub-module imports go through the finder above
__version__ = getattr(_engine, "__version__", "0")
