"""Topology builders for the BASELINE.json configurations (chain, toy CTF, random-N)."""
from . import chainpattern, toy_ctf  # noqa: F401
