"""Same import path as the reference's VLAAttacker/white_patch/UPA.py; implementation: roboticattack_amd.attack.upa."""
from roboticattack_amd.attack.upa import OpenVLAAttacker  # noqa: F401
