"""Same import path as the reference's VLAAttacker/white_patch/TMA.py; implementation: roboticattack_amd.attack.tma."""
from roboticattack_amd.attack.tma import OpenVLAAttacker  # noqa: F401
