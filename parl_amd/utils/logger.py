"""Minimal stand-in for parl.utils.logger (file + stdout logger, parl/utils/logger.py)."""
import logging
import sys

logger = logging.getLogger('parl_amd')
if not logger.handlers:
    _h = logging.StreamHandler(sys.stdout)
    _h.setFormatter(logging.Formatter('[%(asctime)s %(levelname)s] %(message)s', '%m-%d %H:%M:%S'))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)
    logger.propagate = False
