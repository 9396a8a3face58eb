"""Same import path (and spelling) as the reference's operator module; implementation: roboticattack_amd.transform."""
from roboticattack_amd.transform import RandomPatchTransform  # noqa: F401
