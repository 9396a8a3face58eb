"""Same import path as the reference's VLAAttacker/white_patch/UADA.py; implementation: roboticattack_amd.attack.uada."""
from roboticattack_amd.attack.uada import IGNORE_INDEX, OpenVLAAttacker  # noqa: F401
