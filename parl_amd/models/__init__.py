from .atari_model import AtariModel42, AtariModel84, GemmConv2d  # noqa: F401
