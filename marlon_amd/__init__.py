"""marlon_amd — MI355X-native batched CyberBattleSim step engine (host side).

One hot path of zsh239040/MARLon, rebuilt for gfx950: `CyberBattleEnv.step` as driven by
`marlon.simulate` and marlon's env wrappers.  The compute lives in marlon_amd/csrc (HIP, C ABI
in include/mcbs.h); this package is the host-side mirror of the reference's Python interface.
Importing the package does not load the native library; marlon_amd.engine does, and fails
loudly if it is missing (there is no CPU fallback in the product).
"""
__version__ = "0.1.0"
