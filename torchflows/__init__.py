"""``torchflows`` -- import alias of ``torchflows_amd`` (module aliasing only; no reference text).

User code written against the reference imports ``torchflows.flows``,
``torchflows.bijections.finite.autoregressive.architectures`` ... (e.g. the reference's own
``test/test_cuda.py``).  With this directory on ``sys.path`` (the repository root) those imports resolve to THE
SAME module objects as ``torchflows_amd.flows`` ... -- one class hierarchy, so ``isinstance`` checks and state
dicts are interchangeable between the two spellings.  A name this build does not implement (the residual /
continuous families, SURVEY.md section 8 "out of scope") raises ``ModuleNotFoundError`` as it would for any
missing module.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

import torchflows_amd as _impl

_ALIAS, _REAL = __name__, _impl.__name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """``torchflows.x.y`` -> the module object of ``torchflows_amd.x.y`` (imported on first use)."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + "."):
            return None
        real = _REAL + fullname[len(_ALIAS):]
        try:
            module = importlib.import_module(real)
        except ModuleNotFoundError as exc:
            if exc.name is not None and (exc.name == real or real.startswith(exc.name + ".")):
                return None                     # no such module in this build: the normal error follows
            raise
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(module, "__path__"))
        spec._alias_target = module
        return spec

    def create_module(self, spec):
        return spec._alias_target

    def exec_module(self, module):
        return None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

# the package itself: everything torchflows_amd exports, plus its sub-packages as attributes
from torchflows_amd import *  # noqa: F401,F403,E402
from torchflows_amd import Flow, BaseFlow, __version__  # noqa: F401,E402
__path__ = []          # sub-modules come from the finder above, never from this directory
