"""Drop-in import paths of the reference (`from white_patch.UADA import OpenVLAAttacker`, ...)."""
