from .atari_model import AtariModel42, AtariModel84  # noqa: F401
