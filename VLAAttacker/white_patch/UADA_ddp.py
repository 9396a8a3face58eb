"""Same import path as the reference's VLAAttacker/white_patch/UADA_ddp.py; implementation: roboticattack_amd.attack.uada_ddp."""
from roboticattack_amd.attack.uada_ddp import OpenVLAAttacker  # noqa: F401
