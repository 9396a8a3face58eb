"""Attack loops: UADA (single GPU), UADA_ddp (one process per GPU, RCCL), UPA, TMA."""
