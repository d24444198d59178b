"""swem_amd: MI355X (gfx950) implementation of the SWEM inference hot path.

Public surface = the reference's ``methods/SWEM`` module API:
``SWEM`` (swem.py), ``SWEMCore`` / ``MemoryBank`` (modules.py), plus the evaluator loop and helpers.
"""
from .modules import MemoryBank, SWEMCore  # noqa: F401
from .swem import SWEM  # noqa: F401
