"""`import hugectr` -- the module name of the reference's pybind11 extension
(PYBIND11_MODULE(hugectr, m), R/HugeCTR/src/pybind/module_main.cpp:36-48), served by the
MI355X-native implementation in `hugectr_amd.hugectr`.  Existing scripts (`import hugectr`,
`from hugectr.tools import DataGenerator, DataGeneratorParams`) run unchanged with the repository
root on PYTHONPATH."""
import sys as _sys

from hugectr_amd import hugectr as _impl
from hugectr_amd.hugectr import *  # noqa: F401,F403

for _n in dir(_impl):
    if not _n.startswith("_"):
        globals()[_n] = getattr(_impl, _n)

# `from hugectr.tools import DataGeneratorParams, DataGenerator` (R/README.md:62) needs a module
tools = type(_sys)("hugectr.tools")
for _n in dir(_impl.tools):
    if not _n.startswith("_"):
        setattr(tools, _n, getattr(_impl.tools, _n))
_sys.modules[__name__ + ".tools"] = tools
__version__ = "25.03-mi355x"
