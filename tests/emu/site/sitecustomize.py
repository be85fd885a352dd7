"""TEST INFRASTRUCTURE ONLY: child processes of an HCTR_EMU=1 test run (torch.multiprocessing
workers, scripts started with subprocess) get the same stand-ins as the parent (tests/emu/
fakecuda.py); tests/conftest.py puts this directory on PYTHONPATH for that run only."""
import os
import sys

if os.environ.get("HCTR_EMU") == "1":
    _emu = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _root = os.path.dirname(os.path.dirname(_emu))
    for _p in (_root, _emu):
        if _p not in sys.path:
            sys.path.insert(0, _p)
    import fakecuda
    fakecuda.install(os.environ.get("HCTR_EMU_VARIANT"))
