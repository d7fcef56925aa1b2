from ..d2.registry import META_ARCH_REGISTRY


def build_model(cfg):
    """D2 build_model: META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg).to(cfg.MODEL.DEVICE)."""
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
